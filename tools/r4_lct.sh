mkdir -p gpurun_out/r4 && export PYTHONUNBUFFERED=1
timeout -k 10 800 python -m pytest tests/test_lct_gpu.py tests/test_highres_gpu.py tests/test_stages_gpu.py "tests/test_nlospose_gpu.py::test_eval_forward_vs_reference_golden" -q -m gpu > gpurun_out/r4/gpu_tests_lct2.log 2>&1; tail -6 gpurun_out/r4/gpu_tests_lct2.log | cut -c1-300
for f in 1 0; do HP_LCT_FOLD=$f timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > gpurun_out/r4/bench_fold$f.json 2> gpurun_out/r4/bench_fold$f.err; python - <<PY
import json
d=json.load(open('gpurun_out/r4/bench_fold$f.json'))
k=d['hip_kernel_ms_per_step']
print('fold=$f', d['ms_per_step'], {n:v for n,v in k.items() if n.startswith('lct')})
PY
done
