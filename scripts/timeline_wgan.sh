#!/bin/bash
repo=$(pwd); out=$repo/gpurun_out/timeline; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl/wgan -o wgan -- python $repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/wgan.log 2>&1
f=$(find /tmp/tl/wgan -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $repo/scripts/step_timeline.py $f rmsprop 6 > $out/wgan.txt 2>&1
head -1 $out/wgan.txt
