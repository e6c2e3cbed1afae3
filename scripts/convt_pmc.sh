#!/bin/bash
# rocprofv3 counter passes for the ConvT(64->1) kernel (each counter group in its own run, kernel-trace only)
repo=$(pwd); out=$repo/gpurun_out/convt_pmc; mkdir -p $out
echo "# rocprofv3 --kernel-trace --pmc <group> -- python scripts/convt_pmc.py: median counter value and dispatch time per grid size"
echo "# (kernel name + grid size identify the form and the batch: stream = one h parity, stream2 = both, all = all 64 taps per workgroup)"
cd /tmp && export TMPDIR=/tmp
run_pmc() { name=$1; ctrs=$2
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/cp/$name -o $name -- python $repo/scripts/convt_pmc.py > $out/$name.log 2>&1
  f=$(find /tmp/cp/$name -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/${name}_counters.csv; }
run_pmc sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
run_pmc fetch "FETCH_SIZE"
run_pmc tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run_pmc vm "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
run_pmc lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"
cd $repo
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/convt_pmc/*_counters.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.OrderedDict()
    for r in rows:
        if 'convT' not in r['Kernel_Name']: continue
        key=(r['Kernel_Name'].split('<')[0][-22:], r['Grid_Size'], r['Counter_Name'])
        agg.setdefault(key, []).append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
    for (kn,g,c),v in agg.items():
        v.sort(); m=v[len(v)//2]
        print(f.split('/')[-1][:8], kn, 'grid',g, c, '%.4g'%m[0], 'ns',m[1])
PY
