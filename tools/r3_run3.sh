set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_nlospose_gpu.py tests/test_lct_gpu.py tests/test_stages_gpu.py tests/test_data_parallel_gpu.py -q -m gpu -x -s > gpurun_out/r3/gpu_tests_3.log 2>&1 ;
tail -5 gpurun_out/r3/gpu_tests_3.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r3/bench_defer_on.json 2> gpurun_out/r3/bench_defer_on.err &&
HP_BN_DEFER=0 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r3/bench_defer_off.json 2> gpurun_out/r3/bench_defer_off.err ;
timeout -k 10 300 python tools/time_conv_layers.py 512 128 4 wgrad fp32 > gpurun_out/r3/layers_wgrad_fp32.log 2>&1
python - <<'PY'
import json
for f in ("on","off"):
    l=json.loads(open(f"gpurun_out/r3/bench_defer_{f}.json").read().strip().splitlines()[-1])
    print(f, l["ms_per_step"], l["roofline"]["frac"], {k:v for k,v in l["hip_kernel_ms_per_step"].items() if v>5})
PY
