set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_dconv_gpu.py -q -m gpu -x > gpurun_out/r3/gpu_tests_28.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_28.log
for s in 2 3; do
HP_DCONV_SETS=$s timeout -k 10 200 python tools/time_dconv_layers.py > gpurun_out/r3/dconv_layers_fp32_s$s.log 2>&1
HP_DCONV_SETS=$s HP_TIME_DCONV_PRECISION=bf16 timeout -k 10 200 python tools/time_dconv_layers.py > gpurun_out/r3/dconv_layers_bf16_s$s.log 2>&1
done
cd gpurun_out/r3
paste -d'\n' dconv_layers_fp32_s2.log dconv_layers_fp32_s3.log dconv_layers_bf16_s2.log dconv_layers_bf16_s3.log | grep -v "amdgpu.ids\|calibration" | cut -c1-130
