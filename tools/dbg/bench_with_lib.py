"""bench.py with another build of the library: HDP_DBG_LIB=<path> python tools/dbg/bench_with_lib.py [bench args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hdp_amd import _lib
if os.environ.get("HDP_DBG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["HDP_DBG_LIB"])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
