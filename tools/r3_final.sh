set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r3/gpu_tests_final.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_final.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3/smoke_final.log 2>&1 ; tail -2 gpurun_out/r3/smoke_final.log
timeout -k 10 600 python bench.py > gpurun_out/r3/bench_default_final.json 2> gpurun_out/r3/bench_default_final.err
python3 - <<'PY'
import json
l=json.loads(open('gpurun_out/r3/bench_default_final.json').read().strip().splitlines()[-1])
print(l['ms_per_step'], l['value'], l['roofline']['frac'], l['roofline'].get('traffic'))
print({k:(v['ms_per_step']) for k,v in l['extra'].items()}, l['extra']['configs3_highres'].get('bf16_thin_channel',{}).get('ms_per_step'), l['cpu_baseline']['value'])
PY
