mkdir -p gpurun_out/r3
timeout -k 10 300 python tools/dbg/unet_bf16_layers.py > gpurun_out/r3/dbg33.log 2>&1; grep -v amdgpu.ids gpurun_out/r3/dbg33.log | tail -30
