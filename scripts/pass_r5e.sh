#!/bin/bash
repo=$(pwd); out=$repo/gpurun_out/r5e; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
python scripts/edge_cold.py fwd > $out/fwd.json 2>/dev/null; cat $out/fwd.json
python scripts/edge_cold.py convT > $out/convT_split.json 2>/dev/null; cat $out/convT_split.json
SG_CONVT_SPLIT=1 python scripts/edge_cold.py convT > $out/convT_nosplit.json 2>/dev/null; cat $out/convT_nosplit.json
SG_CONVT_MIN_BATCH=8 python scripts/edge_cold.py convT > $out/convT_min8.json 2>/dev/null; cat $out/convT_min8.json
SG_CONVT_MIN_BATCH=8 SG_CONVT_SPLIT=1 python scripts/edge_cold.py convT > $out/convT_min8_nosplit.json 2>/dev/null; cat $out/convT_min8_nosplit.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cat $out/bench.json | cut -c1-200
