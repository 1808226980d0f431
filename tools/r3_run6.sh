set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -q -m gpu -x -k "fp32 or not bf16" > gpurun_out/r3/gpu_tests_6.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_6.log
for s in 0 1; do
HP_IGEMM_BL=$s timeout -k 10 300 python tools/time_conv_layers.py 512 128 4 fwd,dgrad fp32 l1.0.conv2,l1.0.conv3,l1.1.conv1,l2.0,l2.1.conv2,l3.1,l4.1.conv2,deconv,head > gpurun_out/r3/layers_bl_$s.log 2>&1 || exit 1
done
paste <(grep -v "amdgpu\|^T=" gpurun_out/r3/layers_bl_0.log | cut -c1-14,72-140) <(grep -v "amdgpu\|^T=" gpurun_out/r3/layers_bl_1.log | cut -c72-140)
