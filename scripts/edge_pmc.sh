#!/bin/bash
# counters of the one-channel conv kernels (GPU box): one rocprofv3 --pmc pass per counter group over scripts/edge_prof.py
repo=$(pwd); out=$repo/gpurun_out/r03/edge_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; ctrs=$2
  timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/epmc/$name -o $name -- python $repo/scripts/edge_prof.py > $out/$name.log 2>&1
  f=$(find /tmp/epmc/$name -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/${name}_counters.csv || tail -5 $out/$name.log; }
pass p1 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"
pass p2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"
pass p3 "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"
pass p4 "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TA_BUSY_avr TCC_EA0_WRREQ_STALL_sum"
python $repo/scripts/pmc_table.py $out/edge_pmc_table.csv $out/p*_counters.csv 2>/dev/null
grep -E "c1_kernel|convT_c1" $out/edge_pmc_table.csv
