set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/r3/bench_25a.json 2> gpurun_out/r3/bench_25a.err
timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline --wgrad-stream > gpurun_out/r3/bench_25b.json 2> gpurun_out/r3/bench_25b.err
timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/r3/bench_25c.json 2> gpurun_out/r3/bench_25c.err
python3 - <<'PY'
import json
for t in "abc":
    try:
        l=json.loads(open(f"gpurun_out/r3/bench_25{t}.json").read().strip().splitlines()[-1])
        print(t, l["ms_per_step"], l["roofline"], {k:v for k,v in l["hip_kernel_ms_per_step"].items() if k.startswith("conv")})
    except Exception as e: print(t, "ERR", e)
PY
