#!/bin/bash
# round 5, GPU pass C: the LDS-staged one-channel forward (parity + cold A/B against the gather form), headline parity test
repo=$(pwd); out=$repo/gpurun_out/r5c; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
python scripts/edge_cold.py fwd > $out/edge_cold_lds.json 2> $out/edge_cold_lds.err; cat $out/edge_cold_lds.json
SG_FWD_C1_LDS=0 python scripts/edge_cold.py fwd > $out/edge_cold_gather.json 2> $out/edge_cold_gather.err; cat $out/edge_cold_gather.json
timeout 600 python -m pytest tests/test_gpu_modules.py tests/test_gpu_losses.py -x -q -m gpu > $out/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -3 $out/pytest2.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cat $out/bench.json | cut -c1-300
