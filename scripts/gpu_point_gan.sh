# point-GAN family only: its tests, its step times, its kernel stats (the files profile_round.sh writes for it)
set -x
repo=$(pwd); out=$repo/gpurun_out/prof_r03; mkdir -p $out $repo/gpurun_out/r03
timeout 300 python -X faulthandler -m pytest tests -m gpu -q --tb=short --timeout=200 -k "layernorm or segmax or colsum or point or Point or scatter_max" 2>&1 | grep -v amdgpu.ids | tail -12 > $repo/gpurun_out/r03/pytest_pg.log
tail -3 $repo/gpurun_out/r03/pytest_pg.log
python scripts/point_gan_bench.py > $out/r03_point_gan_bench.txt 2> $out/point_gan.err
cat $out/r03_point_gan_bench.txt | cut -c1-150
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sgprof/point_gan -o point_gan -- python $repo/scripts/point_gan_bench.py > $out/point_gan.log 2>&1
f=$(find /tmp/sgprof/point_gan -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/r03_point_gan_kernel_stats.csv
