#!/bin/bash
repo=$(pwd); out=$repo/gpurun_out/r5d; mkdir -p $out
python scripts/edge_cold.py fwd > $out/edge_cold_lds.json 2>/dev/null; cat $out/edge_cold_lds.json
for v in fwdc1_abl1 fwdc1_abl2 fwdc1_abl4 fwdc1_abl3; do
  SHAPEGAN_HIP_LIB=$repo/scripts/_abl/$v.so python scripts/edge_cold.py fwd > $out/edge_cold_$v.json 2> $out/edge_cold_$v.err; cat $out/edge_cold_$v.json
done
bash scripts/kernel_pmc.sh conv_fwd_c1 python scripts/edge_target.py fwd > $out/fwd_pmc.txt 2>&1; cat $out/fwd_pmc.txt
timeout 600 python -m pytest tests/test_gpu_losses.py -x -q -m gpu -k "bad_batch or bad_index" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
