set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py -q -m gpu -x -k "stem" > gpurun_out/r3/gpu_tests_45.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_45.log
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/r3/bench_45.json 2> gpurun_out/r3/bench_45.err
python3 - <<'PY'
import json
l=json.loads(open('gpurun_out/r3/bench_45.json').read().strip().splitlines()[-1])
print(l['ms_per_step'], l['value'], l['roofline']['frac'])
print({k:v for k,v in l['hip_kernel_ms_per_step'].items() if k.startswith('conv') or k.startswith('dconv')})
PY
