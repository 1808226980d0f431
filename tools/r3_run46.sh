mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 500 python tools/time_conv_layers.py 512 128 4 fwd,dgrad,wgrad fp32 > gpurun_out/r3/layers46.log 2>&1
grep -v amdgpu.ids gpurun_out/r3/layers46.log | cut -c1-190
