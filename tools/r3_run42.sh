set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r3/gpu_tests_42.log 2>&1 ; tail -5 gpurun_out/r3/gpu_tests_42.log
timeout -k 10 600 python bench.py > gpurun_out/r3/bench_default_42.json 2> gpurun_out/r3/bench_default_42.err
python3 - <<'PY'
import json
l=json.loads(open('gpurun_out/r3/bench_default_42.json').read().strip().splitlines()[-1])
print(l['ms_per_step'], l['value'], l['roofline']['frac'])
print({k:(v['ms_per_step']) for k,v in l['extra'].items()}, l['extra']['configs3_highres'].get('bf16_thin_channel'))
print({k:v for k,v in l['hip_kernel_ms_per_step'].items() if k.startswith('dconv')})
PY
