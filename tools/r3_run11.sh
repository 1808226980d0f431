set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_entry_points.py tests/test_conv_gpu.py -q -m gpu -x -s -k "bf16_storage_mode_trains or side_stream" > gpurun_out/r3/gpu_tests_11.log 2>&1 ; tail -6 gpurun_out/r3/gpu_tests_11.log
timeout -k 10 300 python bench.py --workload highres --steps 5 --warmup 2 > gpurun_out/r3/bench_highres_11.json 2> gpurun_out/r3/bench_highres_11.err
timeout -k 10 300 python bench.py --conv-precision bf16s --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r3/bench_bf16s_11.json 2> gpurun_out/r3/bench_bf16s_11.err
python3 - <<'PY'
import json
for f in ("highres","bf16s"):
    l=json.loads(open(f"gpurun_out/r3/bench_{f}_11.json").read().strip().splitlines()[-1])
    print(f, l["ms_per_step"], {k:v for k,v in l["hip_kernel_ms_per_step"].items() if v>0.5})
PY
