set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r3/gpu_tests_9.log 2>&1 ; tail -4 gpurun_out/r3/gpu_tests_9.log
rm -rf gpurun_out/r3/prof9; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r3/prof9 -o t512 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r3/bench_prof9.json 2> gpurun_out/r3/bench_prof9.err
find gpurun_out/r3/prof9 -name "*kernel_stats.csv" | head -2
python3 - <<'PY'
import json,glob,csv
l=json.loads(open("gpurun_out/r3/bench_prof9.json").read().strip().splitlines()[-1])
print(l["ms_per_step"], l["roofline"]["frac"], l["mfma_tflops_by_kernel"])
f=glob.glob("gpurun_out/r3/prof9/**/*kernel_stats.csv",recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    steps=5  # 1 warmup + 3 timed + 1 profile step
    for r in rows[:28]:
        print(f"{float(r['TotalDurationNs'])/steps/1e6:8.2f} ms/step {int(r['Calls'])//steps:4d}  {r['Name'][:100]}")
PY
