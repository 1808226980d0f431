mkdir -p gpurun_out/r4 && export PYTHONUNBUFFERED=1
python -c "import torch; print(torch.cuda.Stream.priority_range())" > gpurun_out/r4/prio.txt 2>&1
timeout -k 10 800 python -m pytest tests/test_xformers.py -q -m gpu -x > gpurun_out/r4/gpu_tests_4.log 2>&1; tail -3 gpurun_out/r4/gpu_tests_4.log
for cfg in "off:0:0" "on:1:0" "on:1:1" "on:1:-1"; do
  IFS=: read name ws pr <<< "$cfg"
  HP_WGRAD_STREAM=$ws HP_WGRAD_PRIO=$pr timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extra > gpurun_out/r4/bench_ws_${ws}_${pr}.json 2> gpurun_out/r4/bench_ws_${ws}_${pr}.err
  python -c "
import json;d=json.load(open('gpurun_out/r4/bench_ws_${ws}_${pr}.json'));print('ws=$ws prio=$pr', d['ms_per_step'])"
done
