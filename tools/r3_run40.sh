mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 500 python tools/dbg/bf16s_curves.py wgrad_bf16 > gpurun_out/r3/curves_a.log 2>&1
HP_DCONV_WGRAD_BF16=0 timeout -k 10 500 python tools/dbg/bf16s_curves.py wgrad_exact > gpurun_out/r3/curves_b.log 2>&1
grep -h "bf16s\|fp32 last" gpurun_out/r3/curves_a.log gpurun_out/r3/curves_b.log
