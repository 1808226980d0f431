set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 300 python bench.py --workload sformer --conv-precision bf16 --attention fp16 --steps 10 --warmup 3 > gpurun_out/r3/bench_sformer_53.json 2> gpurun_out/r3/bench_sformer_53.err && \
timeout -k 10 300 python bench.py --workload sformer --steps 10 --warmup 3 > gpurun_out/r3/bench_sformer_53f.json 2> gpurun_out/r3/bench_sformer_53f.err && \
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py -q -m gpu -x > gpurun_out/r3/gpu_tests_53.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_53.log
python3 - <<'PY'
import json
for f in ("bench_sformer_53","bench_sformer_53f"):
    l=json.loads(open(f'gpurun_out/r3/{f}.json').read().strip().splitlines()[-1])
    print(f, l['ms_per_step'], l['value'], l.get('hip_kernel_ms_per_step'))
PY
