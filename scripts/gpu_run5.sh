#!/bin/bash
mkdir -p gpurun_out/r4e
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r4e/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4e/pytest.log
python scripts/edge_ab.py > gpurun_out/r4e/edge_new.json 2> gpurun_out/r4e/edge_new.err; cat gpurun_out/r4e/edge_new.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r4e/bench.json 2> gpurun_out/r4e/bench.err; echo "bench rc=$?"; cut -c1-260 gpurun_out/r4e/bench.json
