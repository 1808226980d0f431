set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_nlospose_gpu.py tests/test_entry_points.py tests/test_highres_gpu.py -q -m gpu -s -k "bf16 or highres" > gpurun_out/r3/gpu_tests_39.log 2>&1 ; grep -a "bf16s\]\|passed\|failed\|bf16s:\|fp32:\|Error" gpurun_out/r3/gpu_tests_39.log | tail -10
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --conv-precision bf16s > gpurun_out/r3/bench_39_bf16s.json 2> gpurun_out/r3/bench_39_bf16s.err
timeout -k 10 300 python bench.py --workload highres --steps 5 --warmup 2 --conv-precision bf16s > gpurun_out/r3/bench_highres_39b.json 2> gpurun_out/r3/bench_highres_39b.err
python3 - <<'PY'
import json
for f in ("bench_39_bf16s", "bench_highres_39b"):
    try:
        l=json.loads(open(f"gpurun_out/r3/{f}.json").read().strip().splitlines()[-1])
        print(f, l["ms_per_step"], {k:v for k,v in l["hip_kernel_ms_per_step"].items() if k.startswith("dconv")})
    except Exception as e: print(f, "ERR", e)
PY
