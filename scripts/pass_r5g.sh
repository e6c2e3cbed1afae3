#!/bin/bash
repo=$(pwd); out=$repo/gpurun_out/r5g; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $out/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cat $out/bench.json | cut -c1-200
