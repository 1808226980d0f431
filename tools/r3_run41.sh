mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
for s in 3; do
HP_DCONV_SETS=$s timeout -k 10 200 python tools/time_dconv_layers.py > gpurun_out/r3/dconv_layers_fp32_s$s.log 2>&1
HP_DCONV_SETS=$s HP_TIME_DCONV_PRECISION=bf16 timeout -k 10 200 python tools/time_dconv_layers.py > gpurun_out/r3/dconv_layers_bf16_s$s.log 2>&1
done
cd gpurun_out/r3
paste -d'\n' dconv_layers_fp32_s3.log dconv_layers_bf16_s3.log | grep -v "amdgpu.ids\|calibration" | cut -c1-130 | grep "sum\|4->4\|8->4 \|1->4\|8->8"
