out=gpurun_out/full2; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $out/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - $out <<'PY'
import json,sys
d=json.load(open(sys.argv[1]+'/bench.json'))
print(d['value'], d['ms_per_step'])
for k in d['kernels']: print(k['kernel'], k['us'], k['frac'])
s=d['sdfnet']
print('fwd', s['fwd_mpoints_per_s'], s['fwd_frac_of_f32_mfma_peak_executed'])
for k in ('train_ref_20k_L128','train_ref_20k_L128_eager','train_cfg_200k_L256'): print(k, s[k]['ms_per_step'], s[k]['frac_of_f32_mfma_peak_executed'])
for k,v in d['other_configs'].items(): print(k, v['value'], v['ms_per_step'])
print(d['dropin_loop'])
PY
