set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
HP_DIST_BACKEND=gloo HP_SHARE_GPU=1 timeout -k 10 900 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r3/bench_n2_default.json 2> gpurun_out/r3/bench_n2_default.err
echo rc=$?
tail -5 gpurun_out/r3/bench_n2_default.err
python3 -c "
import json
l=json.loads(open('gpurun_out/r3/bench_n2_default.json').read().strip().splitlines()[-1])
print(l['n_gpus'], l['ms_per_step'], l['value'], l['config']['world_size'], l['config']['backend'], l['extra'])"
