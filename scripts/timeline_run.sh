#!/bin/bash
# kernel timelines of one steady-state step (GPU box): sorted 200k/L256 auto-decoder step, 20k/L128 step, WGAN step
repo=$(pwd); out=$repo/gpurun_out/timeline; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
tl() { name=$1; marker=$2; per=$3; shift; shift; shift
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl/$name -o $name -- "$@" > $out/$name.log 2>&1
  f=$(find /tmp/tl/$name -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python $repo/scripts/step_timeline.py $f $marker $per > $out/$name.txt 2>&1; }
tl sdf200k adam_kernel 2 python $repo/scripts/sdf_step_prof.py 200000 256
tl sdf20k adam_kernel 2 python $repo/scripts/sdf_step_prof.py 20000 128
tl sdf20k_graphed adam_dev_multi 1 python $repo/scripts/sdf_step_prof_graphed.py 20000 128
tl wgan rmsprop 6 python $repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras
ls -la $out
