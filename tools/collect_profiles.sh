#!/bin/bash
# Collects the committed evidence under profiles/ on a GPU box (run through gpurun; ~6 minutes):
#   rocprofv3 kernel stats of the default bench command (round 4: 7 steps per run = 1 warm-up + 3 timed + 1 + 2 profiled), its JSON line, the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate
#   runs, no trace domains besides --kernel-trace) aggregated per kernel family, and the same for --workload highres.
# The program itself follows `--` (no env / shell / launcher in between).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=${1:-round4}
O=gpurun_out/prof_$R
rm -rf "$O"; mkdir -p "$O" profiles
A="--steps 3 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d $O/stats -o t512 --output-format csv -- python3 bench.py $A > $O/t512_bench.json 2> $O/t512_bench.err || exit 2
# (steps executed by the command: 1 warm-up + 2 timed + 1 after the stream switch + 1 profiled = 5: the divisor below)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-extra > /dev/null 2> $O/fetch.err || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-extra > /dev/null 2> $O/write.err || exit 4
f=$(find $O/stats -name 't512_kernel_stats.csv' | head -1); cp "$f" profiles/${R}_t512_rocprofv3_kernel_stats.csv
tail -1 $O/t512_bench.json > profiles/${R}_t512_bench_under_rocprof.json
# round 4: the default command overlaps the weight gradients with the rest of backward (two kernels in flight: contended durations
# in the trace above); the SAME step with everything on one stream is the trace whose per-kernel averages `roofline` must agree with
rocprofv3 --kernel-trace --stats -d $O/ostats -o t512o --output-format csv -- python3 bench.py $A --no-wgrad-stream > $O/t512_onestream_bench.json 2> $O/t512_onestream_bench.err || exit 12
f=$(find $O/ostats -name 't512o_kernel_stats.csv' | head -1); cp "$f" profiles/${R}_t512_onestream_rocprofv3_kernel_stats.csv
tail -1 $O/t512_onestream_bench.json > profiles/${R}_t512_onestream_bench_under_rocprof.json
python3 tools/aggregate_pmc.py $(find $O/fetch -name 'f_counter_collection.csv' | head -1) $(find $O/write -name 'w_counter_collection.csv' | head -1) 5 \
    profiles/${R}_t512_pmc_hbm_traffic.csv profiles/t512_pmc_hbm_traffic.json > $O/agg_t512.txt || exit 5
# BASELINE configs[2]'s per-GPU share: the same step with bf16 arithmetic and bf16 activation storage
rocprofv3 --kernel-trace --stats -d $O/bstats -o b16 --output-format csv -- python3 bench.py $A --conv-precision bf16s > $O/bf16s_bench.json 2> $O/bf16s_bench.err || exit 10
f=$(find $O/bstats -name 'b16_kernel_stats.csv' | head -1); cp "$f" profiles/${R}_t512_bf16s_rocprofv3_kernel_stats.csv
tail -1 $O/bf16s_bench.json > profiles/${R}_t512_bf16s_bench_under_rocprof.json
H="--workload highres --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats -d $O/hstats -o hr --output-format csv -- python3 bench.py $H > $O/highres_bench.json 2> $O/highres_bench.err || exit 6
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/hfetch -o f --output-format csv -- python3 bench.py --workload highres --steps 2 --warmup 1 > /dev/null 2> $O/hfetch.err || exit 7
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/hwrite -o w --output-format csv -- python3 bench.py --workload highres --steps 2 --warmup 1 > /dev/null 2> $O/hwrite.err || exit 8
f=$(find $O/hstats -name 'hr_kernel_stats.csv' | head -1); cp "$f" profiles/${R}_highres_rocprofv3_kernel_stats.csv
tail -1 $O/highres_bench.json > profiles/${R}_highres_bench_under_rocprof.json
python3 tools/aggregate_pmc.py $(find $O/hfetch -name 'f_counter_collection.csv' | head -1) $(find $O/hwrite -name 'w_counter_collection.csv' | head -1) 3 \
    profiles/${R}_highres_pmc_hbm_traffic.csv > $O/agg_highres.txt || exit 9
# the raw traces are scratch; keep the summaries
rm -rf $O/stats $O/ostats $O/fetch $O/write $O/hstats $O/hfetch $O/hwrite $O/bstats
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_* profiles/t512_pmc_hbm_traffic.json gpurun_out/profiles_$R/
echo "profiles collected"; ls -la profiles | tail -12
