#!/bin/bash
# round 4, GPU pass: ConvT kernel parity + A/B, bench, timeline
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_modules.py -x -q -m gpu > gpurun_out/r4b/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4b/pytest.log
python scripts/edge_ab.py > gpurun_out/r4b/edge_new.json 2> gpurun_out/r4b/edge_new.err; cat gpurun_out/r4b/edge_new.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r4b/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4b/bench.json'))
for k in d['kernels']: print(k['kernel'], k['us'], k['frac'])
print(json.dumps(d['sdfnet'])[:600]); print(json.dumps(d['other_configs'])[:900])
PY
