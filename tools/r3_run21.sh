set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_dconv_gpu.py tests/test_stages_gpu.py tests/test_highres_gpu.py tests/test_entry_points.py -q -m gpu -x > gpurun_out/r3/gpu_tests_21.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_21.log
timeout -k 10 300 python bench.py --workload highres --steps 5 --warmup 2 > gpurun_out/r3/bench_highres_21.json 2> gpurun_out/r3/bench_highres_21.err
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r3/bench_21.json 2> gpurun_out/r3/bench_21.err
python3 - <<'PY'
import json
for f in ("bench_highres_21","bench_21"):
    l=json.loads(open(f"gpurun_out/r3/{f}.json").read().strip().splitlines()[-1])
    print(f, l["ms_per_step"], {k:v for k,v in l["hip_kernel_ms_per_step"].items() if k.startswith(("dconv","gn_"))})
PY
