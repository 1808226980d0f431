set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_nlospose_gpu.py tests/test_sformer.py tests/test_xformers.py -q -m gpu -x > gpurun_out/r3/gpu_tests_17.log 2>&1 ; tail -3 gpurun_out/r3/gpu_tests_17.log
timeout -k 10 300 python tools/time_conv_layers.py 512 128 4 fwd,dgrad fp32 l1,l2.0,l2.1,l3.1,l4.1,head > gpurun_out/r3/layers_epi.log 2>&1
grep -v "amdgpu\|^T=" gpurun_out/r3/layers_epi.log | cut -c1-14,72-140
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r3/bench_17.json 2> gpurun_out/r3/bench_17.err
python3 -c "
import json
l=json.loads(open('gpurun_out/r3/bench_17.json').read().strip().splitlines()[-1])
print(l['ms_per_step'], l['roofline']['frac'], l['mfma_tflops_by_kernel'])"
