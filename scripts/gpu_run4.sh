#!/bin/bash
mkdir -p gpurun_out/r4d
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4d/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4d/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4d/bench.json 2> gpurun_out/r4d/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4d/bench.json'))
print(d['value'], d['ms_per_step'])
for k in d['kernels']: print(k['kernel'], k['us'], k['frac'])
s=d['sdfnet']
print('fwd', s['fwd_mpoints_per_s'], s['fwd_frac_of_f32_mfma_peak_executed'])
for k in ('train_ref_20k_L128','train_ref_20k_L128_eager','train_cfg_200k_L256'): print(k, s[k]['ms_per_step'], s[k]['frac_of_f32_mfma_peak_executed'])
for k,v in d['other_configs'].items(): print(k, v['value'], v['ms_per_step'])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o sdf -- python $GRAFT_REPO_ROOT/scripts/sdf_train_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r4d/sdf_prof.log 2>&1
f=$(find /tmp/sp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r4d/sdf_train_kernel_stats.csv && head -12 $f | cut -c1-160
