# A/B sweep of the metrics overlap knobs in the bench regime (ON THE GPU BOX): prints ms per step, thresholds, metrics per band
mkdir -p gpurun_out/r02
run() { tag=$1; shift; env "$@" timeout -k 10 300 python tools/dbg/bench_with_lib.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r02/sw_$tag.json 2> gpurun_out/r02/sw_$tag.err; python3 -c "
import json
d=json.loads(open('gpurun_out/r02/sw_$tag.json').read().strip().split('\n')[-1])
print('$tag', round(d['ms_per_step'],1), round(d['kernels']['thresholds_kernel']['ms_per_launch'],1), round(d['kernels']['metrics_kernel']['ms_per_launch'],1))
"; }
L=hdp_amd/libhdp_a5.so
run a5lds20 HDP_DBG_LIB=$L HDP_METRICS_YEARS_LDS=20480
run a5lds16 HDP_DBG_LIB=$L HDP_METRICS_YEARS_LDS=16384
run a5lds24 HDP_DBG_LIB=$L HDP_METRICS_YEARS_LDS=24576
run a5lds20b48 HDP_DBG_LIB=$L HDP_METRICS_YEARS_LDS=20480 HDP_METRICS_BATCH=49152
run a5lds20b64 HDP_DBG_LIB=$L HDP_METRICS_YEARS_LDS=20480 HDP_METRICS_BATCH=65536
run a5lds20b40 HDP_DBG_LIB=$L HDP_METRICS_YEARS_LDS=20480 HDP_METRICS_BATCH=40960
run a5lds28b48 HDP_DBG_LIB=$L HDP_METRICS_YEARS_LDS=28672 HDP_METRICS_BATCH=49152
