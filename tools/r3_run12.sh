set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -q -m gpu -x -k "bf16" > gpurun_out/r3/gpu_tests_12.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_12.log
for s in 0 32 64; do
HP_WGRAD_BLH=$s timeout -k 10 300 python tools/time_conv_layers.py 512 128 4 wgrad bf16s l1.0.conv2,l1.0.conv3,l1.1.conv1,l2.0,l2.1.conv2,l3.1,l4.1.conv2,deconv,head > gpurun_out/r3/layers_blh_$s.log 2>&1 || exit 1
done
paste <(grep -v "amdgpu\|^T=" gpurun_out/r3/layers_blh_0.log | cut -c1-14,72-110) <(grep -v "amdgpu\|^T=" gpurun_out/r3/layers_blh_32.log | cut -c72-110) <(grep -v "amdgpu\|^T=" gpurun_out/r3/layers_blh_64.log | cut -c72-110)
timeout -k 10 300 python bench.py --conv-precision bf16s --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r3/bench_bf16s_12.json 2> gpurun_out/r3/bench_bf16s_12.err
python3 -c "
import json
l=json.loads(open('gpurun_out/r3/bench_bf16s_12.json').read().strip().splitlines()[-1])
print(l['ms_per_step'], l['mfma_tflops_by_kernel'])"
