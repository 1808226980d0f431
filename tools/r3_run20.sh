set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r3/gpu_tests_20.log 2>&1 ; tail -4 gpurun_out/r3/gpu_tests_20.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3/smoke_20.log 2>&1; tail -3 gpurun_out/r3/smoke_20.log
timeout -k 10 600 python bench.py > gpurun_out/r3/bench_default_20.json 2> gpurun_out/r3/bench_default_20.err ; tail -3 gpurun_out/r3/bench_default_20.err
python3 -c "
import json
l=json.loads(open('gpurun_out/r3/bench_default_20.json').read().strip().splitlines()[-1])
print(l['ms_per_step'], l['value'], l['roofline'])
print({k:(v['ms_per_step']) for k,v in l['extra'].items()}, l['cpu_baseline']['value'])"
