# Round 4: the stage-range tests, a marker trace of the headline step, and the fp32 per-layer bound table.
mkdir -p gpurun_out/r4 && export PYTHONUNBUFFERED=1
timeout -k 10 300 python -m pytest tests/test_entry_points.py tests/test_abi_host.py -q -m gpu -k "ranges or reduces" > gpurun_out/r4/gpu_tests_ranges.log 2>&1; tail -5 gpurun_out/r4/gpu_tests_ranges.log | cut -c1-220
export HP_ROCTX=1
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --marker-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4/markers -o t512 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/gpurun_out/r4/bench_markers.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4/bench_markers.err ) &&
unset HP_ROCTX &&
find gpurun_out/r4/markers -name "*marker*stats*" | head && find gpurun_out/r4/markers -name "*_kernel_trace.csv" -delete; find gpurun_out/r4/markers -name "*marker*stats*.csv" -exec cat {} \; | head -30 &&
timeout -k 10 500 python tools/bound_table.py 512 128 4 fp32 > gpurun_out/r4/bound_table_fp32.txt 2> gpurun_out/r4/bound_table_fp32.err; tail -3 gpurun_out/r4/bound_table_fp32.txt
