set -x
repo=$(pwd); mkdir -p $repo/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o pg -- python $repo/scripts/point_gan_bench.py > $repo/gpurun_out/r03/point_gan_f.txt 2>&1
f=$(find /tmp/pg -name "*kernel_stats.csv" | head -1); cp $f $repo/gpurun_out/r03/point_gan_f_stats.csv
head -30 $repo/gpurun_out/r03/point_gan_f_stats.csv | cut -c1-200
