set -x
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
timeout -k 10 300 python tools/time_conv_layers.py 512 128 4 fwd,dgrad fp32 l1.0.conv2,l1.0.conv3,l1.1.conv1,l2.0,l2.1.conv2,l3.1.conv2,l4.1.conv2,deconv,head > gpurun_out/r3/layers_v4_off.log 2>&1 &&
HP_IGEMM_V4=1 timeout -k 10 300 python tools/time_conv_layers.py 512 128 4 fwd,dgrad fp32 l1.0.conv2,l1.0.conv3,l1.1.conv1,l2.0,l2.1.conv2,l3.1.conv2,l4.1.conv2,deconv,head > gpurun_out/r3/layers_v4_on.log 2>&1 &&
HP_IGEMM_V4=1 timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "fp32 or not bf16" > gpurun_out/r3/conv_test_v4.log 2>&1 ;
timeout -k 10 300 python -m pytest tests/test_sformer.py tests/test_ingest.py -q -m gpu -s > gpurun_out/r3/sformer_test.log 2>&1 ;
HP_DIST_BACKEND=gloo HP_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --workload native --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3/bench_selflaunch.json 2> gpurun_out/r3/bench_selflaunch.err ;
HP_FORCE_REDUCER=1 timeout -k 10 300 python bench.py --workload native --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r3/bench_force.json 2> gpurun_out/r3/bench_force.err ;
cat gpurun_out/r3/bench_force.json | cut -c1-300; cat gpurun_out/r3/bench_selflaunch.json | cut -c1-300
