set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
for s in 0 2 4 8; do
HP_MFMA_STAGGER=$s timeout -k 10 300 python tools/time_conv_layers.py 512 128 4 fwd,dgrad,wgrad fp32 l1.0.conv2,l2.1.conv2,l3.1.conv2,deconv1,deconv2 > gpurun_out/r3/layers_stagger_$s.log 2>&1 || exit 1
done
timeout -k 10 300 python -m pytest tests/test_conv_gpu.py -q -m gpu -x -k "deferred" > gpurun_out/r3/gpu_tests_4.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_4.log
for s in 0 2 4 8; do grep -E "deconv2|l1.0.conv2" gpurun_out/r3/layers_stagger_$s.log | cut -c1-15,70-200; done
