# Round-4 closing run on the GPU box: the whole -m gpu suite, smoke(), the default bench line (as the driver runs it).
mkdir -p gpurun_out/r4 && export PYTHONUNBUFFERED=1
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r4/gpu_tests_final.log 2>&1; tail -14 gpurun_out/r4/gpu_tests_final.log | cut -c1-220
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/smoke_final.log 2>&1; tail -2 gpurun_out/r4/smoke_final.log
timeout -k 10 900 python bench.py > gpurun_out/r4/bench_final.json 2> gpurun_out/r4/bench_final.err; python tools/show_bench.py gpurun_out/r4/bench_final.json | head -4
