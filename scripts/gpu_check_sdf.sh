#!/bin/bash
# SDFNet path after a kernel change: parity tests, step timings, 200k-step timeline
repo=$(pwd); out=$repo/gpurun_out/sdfcheck; mkdir -p $out
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_dp.py -x -q -k "sdf or SDF or autodecoder or auto_decoder or segments" > $out/tests.log 2>&1
tail -5 $out/tests.log
timeout 200 python scripts/sdf_train_bench.py > $out/bench.log 2>&1; cat $out/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl/sdf200k -o sdf200k -- python $repo/scripts/sdf_step_prof.py 200000 256 > $out/tl.log 2>&1
f=$(find /tmp/tl/sdf200k -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $repo/scripts/step_timeline.py $f adam_kernel 2 > $out/sdf200k.txt 2>&1
timeout 100 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl/sdf20k -o sdf20k -- python $repo/scripts/sdf_step_prof.py 20000 128 > $out/tl20.log 2>&1
f=$(find /tmp/tl/sdf20k -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $repo/scripts/step_timeline.py $f adam_kernel 2 > $out/sdf20k.txt 2>&1
