#!/bin/bash
repo=$(pwd); out=$repo/gpurun_out/r5f; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "linear or gemm or layernorm or segmax or point" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
timeout 600 python -m pytest tests/test_gpu_modules.py tests/test_gpu_losses.py -x -q -m gpu > $out/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -3 $out/pytest2.log
python scripts/point_gan_bench.py > $out/point_gan_new.txt 2> $out/pg.err; cat $out/point_gan_new.txt
SG_GEMM128=0 python scripts/point_gan_bench.py > $out/point_gan_old.txt 2> $out/pg_old.err; cat $out/point_gan_old.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o pg -- python $repo/scripts/point_gan_bench.py > $out/pg_prof.log 2>&1; f=$(find /tmp/pg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/point_gan_kernel_stats.csv )
head -12 $out/point_gan_kernel_stats.csv | cut -c1-220
