mkdir -p gpurun_out/r3
timeout -k 10 300 python tools/dbg/wgrad_bf16.py > gpurun_out/r3/dbg38.log 2>&1; grep -v amdgpu.ids gpurun_out/r3/dbg38.log | tail -60
