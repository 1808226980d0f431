set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -q -m gpu -x -k "fp32 or not bf16" > gpurun_out/r3/gpu_tests_7.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_7.log
for s in 0 1; do
HP_WGRAD_BL=$s timeout -k 10 300 python tools/time_conv_layers.py 512 128 4 wgrad fp32 l1.0.conv1,l1.0.conv2,l1.0.conv3,l1.1.conv1,l2.0,l2.1.conv2,l3.1,l4.1,deconv,head > gpurun_out/r3/layers_wbl_$s.log 2>&1 || exit 1
done
paste <(grep -v "amdgpu\|^T=" gpurun_out/r3/layers_wbl_0.log | cut -c1-14,72-110) <(grep -v "amdgpu\|^T=" gpurun_out/r3/layers_wbl_1.log | cut -c72-110)
timeout -k 10 900 python -m pytest tests/test_nlospose_gpu.py tests/test_stages_gpu.py -q -m gpu -x > gpurun_out/r3/gpu_tests_7b.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_7b.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r3/bench_bl.json 2> gpurun_out/r3/bench_bl.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r3/bench_bl.json").read().strip().splitlines()[-1])
print(l["ms_per_step"], l["roofline"]["frac"], l["mfma_tflops_by_kernel"])
PY
