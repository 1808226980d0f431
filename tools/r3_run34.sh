mkdir -p gpurun_out/r3
timeout -k 10 300 python tools/dbg/unet_bf16_chain.py > gpurun_out/r3/dbg34.log 2>&1; grep -v amdgpu.ids gpurun_out/r3/dbg34.log | tail -30
