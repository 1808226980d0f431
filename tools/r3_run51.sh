set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_xformers.py tests/test_sformer.py tests/test_conv_gpu.py -q -m gpu -x > gpurun_out/r3/gpu_tests_51.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_51.log
timeout -k 10 300 python bench.py --workload sformer --conv-precision bf16 --attention fp16 --steps 10 --warmup 3 > gpurun_out/r3/bench_sformer_51.json 2> gpurun_out/r3/bench_sformer_51.err
timeout -k 10 300 python bench.py --workload sformer --steps 10 --warmup 3 > gpurun_out/r3/bench_sformer_51f.json 2> gpurun_out/r3/bench_sformer_51f.err
python3 - <<'PY'
import json
for f in ("bench_sformer_51","bench_sformer_51f"):
    l=json.loads(open(f'gpurun_out/r3/{f}.json').read().strip().splitlines()[-1])
    print(f, l['ms_per_step'], l['value'], l.get('hip_kernel_ms_per_step'))
PY
