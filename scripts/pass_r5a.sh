#!/bin/bash
# round 5, GPU pass A: paired stores of conv_dgrad_halo (A/B + counters), the whole GPU tier, the bench line with cold HBM rows
repo=$(pwd); out=$repo/gpurun_out/r5a; mkdir -p $out
for v in base nopair; do
  lib=$repo/scripts/_abl/$v.so; [ $v = base ] && lib=$repo/shapegan_amd/libshapegan_hip.so
  SHAPEGAN_HIP_LIB=$lib python scripts/dgrad_target.py time > $out/dgrad_time_$v.json 2> $out/dgrad_time_$v.err; echo "== $v"; cat $out/dgrad_time_$v.json
done
( cd /tmp && export TMPDIR=/tmp
for v in base nopair; do
  lib=$repo/scripts/_abl/$v.so; [ $v = base ] && lib=$repo/shapegan_amd/libshapegan_hip.so
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    n=${c%% *}
    SHAPEGAN_HIP_LIB=$lib rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/dg/$v$n -o x -- python $repo/scripts/dgrad_target.py > $out/pmc_$v$n.log 2>&1
    f=$(find /tmp/dg/$v$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/dgrad_${v}_${n}.csv
  done
done )
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/r5a/dgrad_*_*.csv')):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if 'conv_dgrad_halo' not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'].split('(')[0][-28:], r['Grid_Size'], r['Counter_Name'])
        agg.setdefault(key, []).append(float(r['Counter_Value']))
    for k, v in agg.items():
        v.sort(); print(f.split('/')[-1], k, v[len(v)//2])
PY
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -3 $out/bench.err
python - $out <<'PY'
import json,sys
d=json.load(open(sys.argv[1]+'/bench.json'))
print(d['value'], d['ms_per_step'])
for k in d['kernels']: print(k['kernel'], k['us'], k['frac'], k.get('us_warm'), k.get('us_in_step'))
s=d['sdfnet']
for k in ('train_ref_20k_L128','train_ref_20k_L128_eager','train_cfg_200k_L256'): print(k, s[k]['ms_per_step'], s[k]['frac_of_f32_mfma_peak_executed'])
for k,v in d['other_configs'].items(): print(k, v['value'], v['ms_per_step'])
PY
