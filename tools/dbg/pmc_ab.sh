#!/bin/bash
# dev tool: HBM-side traffic per kernel family of the headline step with the environment as exported by the caller
#   HP_IGEMM_SLAB=1 bash tools/dbg/pmc_ab.sh tag
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-ab}; O=gpurun_out/pmc_$T; rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/fetch.json 2> $O/fetch.err || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $O/write.err || exit 4
python3 tools/aggregate_pmc.py $(find $O/fetch -name 'f_counter_collection.csv' | head -1) $(find $O/write -name 'w_counter_collection.csv' | head -1) 4 $O/traffic.csv $O/traffic.json > $O/agg.txt || exit 5
rm -rf $O/fetch $O/write
cat $O/traffic.json; tail -1 $O/fetch.json | cut -c1-120
