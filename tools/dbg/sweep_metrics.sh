# A/B sweep of the metrics overlap knobs in the bench regime (ON THE GPU BOX): prints ms per step, thresholds, metrics per band
mkdir -p gpurun_out/r02
run() { tag=$1; shift; env "$@" timeout -k 10 300 python tools/dbg/bench_with_lib.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r02/sw_$tag.json 2> gpurun_out/r02/sw_$tag.err; python3 -c "
import json
d=json.loads(open('gpurun_out/r02/sw_$tag.json').read().strip().split('\n')[-1])
print('$tag', round(d['ms_per_step'],1), round(d['kernels']['thresholds_kernel']['ms_per_launch'],1), round(d['kernels']['metrics_kernel']['ms_per_launch'],1))
"; }
run base X=1
run w15 HDP_DBG_LIB=hdp_amd/libhdp_w15.so
run w16 HDP_DBG_LIB=hdp_amd/libhdp_w16.so
run w15lds20 HDP_DBG_LIB=hdp_amd/libhdp_w15.so HDP_METRICS_YEARS_LDS=20480
run w16lds20 HDP_DBG_LIB=hdp_amd/libhdp_w16.so HDP_METRICS_YEARS_LDS=20480
