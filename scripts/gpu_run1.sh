#!/bin/bash
# round 4, GPU pass 1: new kernels' parity first, then the whole GPU tier, A/B of the ConvT kernels, bench, WGAN timeline
mkdir -p gpurun_out/r4a
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_losses.py tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r4a/pytest_new.log 2>&1; echo "pytest_new rc=$?"
tail -5 gpurun_out/r4a/pytest_new.log
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_losses.py --deselect tests/test_gpu_ops.py --durations=8 > gpurun_out/r4a/pytest_rest.log 2>&1; echo "pytest_rest rc=$?"
tail -15 gpurun_out/r4a/pytest_rest.log
python scripts/edge_ab.py > gpurun_out/r4a/edge_new.json 2> gpurun_out/r4a/edge_new.err; cat gpurun_out/r4a/edge_new.json
SG_NO_EDGE=16 python scripts/edge_ab.py > gpurun_out/r4a/edge_old.json 2> gpurun_out/r4a/edge_old.err; cat gpurun_out/r4a/edge_old.json
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r4a/bench.json
timeout 600 bash scripts/timeline_run.sh > gpurun_out/r4a/timeline.log 2>&1
cp gpurun_out/timeline/wgan.txt gpurun_out/r4a/wgan_timeline.txt; head -3 gpurun_out/r4a/wgan_timeline.txt; tail -2 gpurun_out/r4a/wgan_timeline.txt
