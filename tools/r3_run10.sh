set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -q -m gpu -x -k "fp32 or not bf16" > gpurun_out/r3/gpu_tests_10.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_10.log
for s in 0 1; do
HP_WGRAD_UNI=$s timeout -k 10 300 python tools/time_conv_layers.py 512 128 4 wgrad fp32 l1.0.conv1,l1.0.conv2,l1.0.conv3,l1.1.conv1,l2.0,l2.1.conv2,l3.1,l4.1,deconv,head > gpurun_out/r3/layers_uni_$s.log 2>&1 || exit 1
done
paste <(grep -v "amdgpu\|^T=" gpurun_out/r3/layers_uni_0.log | cut -c1-14,72-110) <(grep -v "amdgpu\|^T=" gpurun_out/r3/layers_uni_1.log | cut -c72-110)
rm -rf gpurun_out/r3/prof10; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r3/prof10 -o t512 --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r3/bench_prof10.json 2> gpurun_out/r3/bench_prof10.err
python3 - <<'PY'
import json,glob,csv
l=json.loads(open("gpurun_out/r3/bench_prof10.json").read().strip().splitlines()[-1])
print(l["ms_per_step"], l["roofline"]["frac"], l["mfma_tflops_by_kernel"])
f=glob.glob("gpurun_out/r3/prof10/**/*kernel_stats.csv",recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    steps=5
    for r in rows[:32]:
        print(f"{float(r['TotalDurationNs'])/steps/1e6:8.2f} ms/step {int(r['Calls'])//steps:4d}  {r['Name'][:110]}")
PY
