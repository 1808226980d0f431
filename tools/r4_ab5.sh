mkdir -p gpurun_out/r4 && export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_nlospose_gpu.py tests/test_stages_gpu.py tests/test_data_parallel_gpu.py tests/test_entry_points.py -q -m gpu > gpurun_out/r4/gpu_tests_bns4.log 2>&1; tail -5 gpurun_out/r4/gpu_tests_bns4.log | cut -c1-200
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r4/bench_$name.json 2> gpurun_out/r4/bench_$name.err; python -c "
import json;d=json.load(open('gpurun_out/r4/bench_$name.json'));k=d['hip_kernel_ms_per_step'];print('$name', d['ms_per_step'], d.get('unoverlapped_profiled_ms_per_step'), 'dgrad', k.get('conv_igemm_dgrad'), 'deconv', k.get('conv_igemm_deconv'), 'k1', k.get('conv_igemm_k1'), 'bn_red', k.get('bn_bwd_reduce'), 'bn_apply', k.get('bn_bwd_apply'))"; }
run fuse1 HP_BN_FUSE=1
run fuse0 HP_BN_FUSE=0
run fuse1b HP_BN_FUSE=1
run fuse0b HP_BN_FUSE=0
