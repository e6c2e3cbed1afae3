#!/bin/bash
# round 5, GPU pass B: where does conv_fwd_c1's time go (ablations, cold), what the write path takes (microbenchmark), new tests
repo=$(pwd); out=$repo/gpurun_out/r5b; mkdir -p $out
scripts/micro/write_pattern > $out/write_pattern.jsonl 2> $out/write_pattern.err; cat $out/write_pattern.jsonl
python scripts/edge_cold.py all > $out/edge_cold_base.json 2> $out/edge_cold_base.err; cat $out/edge_cold_base.json
for v in fwdc1_abl1 fwdc1_abl2 fwdc1_abl4 fwdc1_abl3; do
  SHAPEGAN_HIP_LIB=$repo/scripts/_abl/$v.so python scripts/edge_cold.py fwd > $out/edge_cold_$v.json 2> $out/edge_cold_$v.err; cat $out/edge_cold_$v.json
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_ops.py -x -q -m gpu -k "headline or loaders_deliver or data_is_never or keeps_its_weight" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout 600 python - > $out/dropin_loop.json 2> $out/dropin_loop.err <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench, torch
step, info, _ = bench.make_wgan(0)
for _ in range(5): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
v = 20 / (time.perf_counter() - t0)
d = bench.dropin_loop_numbers()
d["trainer_step_steps_per_s"] = round(v, 3)
d["trainer_step_launches_per_step"] = bench._count_launches(step)
print(json.dumps(d))
PY
cat $out/dropin_loop.json; tail -3 $out/dropin_loop.err
