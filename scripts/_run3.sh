repo=$PWD; cd /tmp && export TMPDIR=/tmp
for v in base v1 none; do
  lib=$repo/scripts/_abl/$v.so; [ $v = base ] && lib=$repo/shapegan_amd/libshapegan_hip.so
  for g in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL" "SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU2"; do
  rm -rf /tmp/ic/$v
  SHAPEGAN_HIP_LIB=$lib rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/ic/$v -o $v -- python $repo/scripts/prof_targets.py sdfstep > /tmp/ic_$v.log 2>&1
  f=$(find /tmp/ic/$v -name "*counter_collection.csv" | head -1)
  python - $f $v <<'PY'
import csv,sys,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'sdfnet_bwd' in r['Kernel_Name'] or 'sdfnet_fwd' in r['Kernel_Name']:
        agg[(r['Kernel_Name'][13:26],r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    v.sort(); print(sys.argv[2], k[0],k[1],'%.4g'%v[len(v)//2])
PY
  done
done
