#!/bin/bash
mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "transpose or dgrad or convT" > gpurun_out/r4c/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4c/pytest.log
python scripts/edge_ab.py > gpurun_out/r4c/edge_new.json 2> gpurun_out/r4c/edge_new.err; cat gpurun_out/r4c/edge_new.json
