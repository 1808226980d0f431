set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_sformer.py tests/test_xformers.py -q -m gpu -x > gpurun_out/r3/gpu_tests_49.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_49.log
timeout -k 10 300 python bench.py --workload sformer --conv-precision bf16 --attention fp16 --steps 10 --warmup 3 > gpurun_out/r3/bench_sformer_49.json 2> gpurun_out/r3/bench_sformer_49.err
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --conv-precision bf16 > gpurun_out/r3/bench_49_bf16.json 2> gpurun_out/r3/bench_49_bf16.err
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --conv-precision bf16x3 > gpurun_out/r3/bench_49_bf16x3.json 2> gpurun_out/r3/bench_49_bf16x3.err
python3 - <<'PY'
import json
for f in ("bench_sformer_49","bench_49_bf16","bench_49_bf16x3"):
    l=json.loads(open(f'gpurun_out/r3/{f}.json').read().strip().splitlines()[-1])
    print(f, l['ms_per_step'], l['value'], {k:v for k,v in l.get('hip_kernel_ms_per_step',{}).items() if 'linear' in k or 'conv_igemm' in k})
PY
