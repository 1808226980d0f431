cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
rm -rf gpurun_out/r4/trace
rocprofv3 --kernel-trace -d gpurun_out/r4/trace -o t --output-format csv -- python3 bench.py --steps 3 --warmup 2 --profile-steps 1 --no-cpu-baseline --no-extra > gpurun_out/r4/trace_bench.json 2> gpurun_out/r4/trace_bench.err || exit 2
f=$(find gpurun_out/r4/trace -name 't_kernel_trace.csv' | head -1)
# steps in the trace: 2 warm-up + 3 timed + 1 + 1 profiled = 7; step index 3 = a timed (overlapped) step, 6 = the one-stream profiled step
python3 tools/gap_analysis.py "$f" 7 3 > gpurun_out/r4/gaps_overlapped.txt 2>&1
python3 tools/gap_analysis.py "$f" 7 6 > gpurun_out/r4/gaps_onestream.txt 2>&1
rm -rf gpurun_out/r4/trace
cat gpurun_out/r4/gaps_overlapped.txt; echo; cat gpurun_out/r4/gaps_onestream.txt
