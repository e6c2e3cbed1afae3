#!/bin/bash
repo=$(pwd); out=$repo/gpurun_out/ab; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$repo/scripts/_abl/$v.so; [ $v = base ] && lib=$repo/shapegan_amd/libshapegan_hip.so
  SHAPEGAN_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl/$v -o $v -- python $repo/scripts/sdf_step_prof.py 200000 256 > $out/$v.log 2>&1
  f=$(find /tmp/tl/$v -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python $repo/scripts/step_timeline.py $f adam_kernel 2 > $out/$v.txt 2>&1
  echo "== $v"; grep "step:\|sdfnet_fwd\|sdfnet_bwd\|gemm_nt_bigk" $out/$v.txt
done
