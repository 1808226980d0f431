set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py -q -m gpu -x -k "stem" > gpurun_out/r3/gpu_tests_stem.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_stem.log
timeout -k 10 300 python tools/time_conv_layers.py 512 128 4 fwd,dgrad,wgrad fp32 stem > gpurun_out/r3/layers_stem.log 2>&1; grep -v amdgpu gpurun_out/r3/layers_stem.log | tail -2
