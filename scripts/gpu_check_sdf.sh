#!/bin/bash
# SDFNet path after a kernel change: parity tests, then step timings / timelines of the variants given as arguments
repo=$(pwd); out=$repo/gpurun_out/sdfcheck; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_dp.py -x -q -k "sdf or SDF or autodecoder or auto_decoder or segments or hybrid or gradient_penalty" > $out/tests.log 2>&1
tail -5 $out/tests.log
timeout 200 python scripts/sdf_train_bench.py > $out/bench.log 2>&1; cat $out/bench.log
timeout 120 python scripts/sdf_fwd_bench.py 2>&1 | grep fwd
bash scripts/ab_run.sh base "$@"
