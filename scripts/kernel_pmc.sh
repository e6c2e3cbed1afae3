#!/bin/bash
# SQ / LDS / TCC counter groups (one rocprofv3 run each, kernel-trace only) for the kernels whose name contains $1, over the command
# that follows:   bash scripts/kernel_pmc.sh sdfnet_bwd python scripts/prof_targets.py sdfstep
pat=$1; shift
repo=$(pwd); out=$repo/gpurun_out/kernel_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run_pmc() { name=$1; ctrs=$2; shift; shift
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/kp/$name -o $name -- "$@" > $out/$name.log 2>&1
  f=$(find /tmp/kp/$name -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/${name}_counters.csv; }
cmd=("$@"); for i in "${!cmd[@]}"; do [ -e "$repo/${cmd[$i]}" ] && cmd[$i]="$repo/${cmd[$i]}"; done
run_pmc sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "${cmd[@]}"
run_pmc lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS" "${cmd[@]}"
run_pmc inst "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"  "${cmd[@]}"
run_pmc tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "${cmd[@]}"
run_pmc mfma "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" "${cmd[@]}"
cd $repo
python - "$pat" <<'PY'
import csv, collections, glob, sys
pat = sys.argv[1]
for f in sorted(glob.glob('gpurun_out/kernel_pmc/*_counters.csv')):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if pat not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'].split('(')[0][-48:], r['Counter_Name'])
        agg.setdefault(key, []).append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    for (k, c), v in agg.items():
        v.sort(); m = v[len(v) // 2]
        print('%-10s %-48s %-28s %.4g  ns %d' % (f.split('/')[-1][:-13], k, c, m[0], m[1]))
PY
