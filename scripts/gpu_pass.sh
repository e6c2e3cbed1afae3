#!/bin/bash
# generic GPU pass of a working session: parity of the kernels touched, edge-kernel A/B numbers, the bench line (no CPU leg)
out=gpurun_out/${1:-pass}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
python scripts/edge_ab.py > $out/edge.json 2> $out/edge.err; cat $out/edge.json
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - $out <<'PY'
import json,sys
d=json.load(open(sys.argv[1]+'/bench.json'))
print(d['value'], d['ms_per_step'])
for k in d['kernels']: print(k['kernel'], k['us'], k['frac'])
s=d['sdfnet']
print('fwd', s['fwd_mpoints_per_s'], s['fwd_frac_of_f32_mfma_peak_executed'])
for k in ('train_ref_20k_L128','train_ref_20k_L128_eager','train_cfg_200k_L256'): print(k, s[k]['ms_per_step'], s[k]['frac_of_f32_mfma_peak_executed'])
for k,v in d['other_configs'].items(): print(k, v['value'], v['ms_per_step'])
PY
