#!/bin/bash
# rocprofv3 evidence of a round, run on the GPU box:  bash scripts/profile_round.sh r02
# kernel-trace/stats and every --pmc group in separate passes (counters never share a pass with other trace domains)
tag=${1:-rXX}
repo=$(pwd)
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run_stats() {  # name, command...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sgprof/$name -o $name -- "$@" > $out/$name.log 2>&1
  f=$(find /tmp/sgprof/$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_${name}_kernel_stats.csv
}
run_pmc() {  # name, counters, command...
  name=$1; ctrs=$2; shift; shift
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/sgprof/$name -o $name -- "$@" > $out/$name.log 2>&1
  f=$(find /tmp/sgprof/$name -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/${name}_counters.csv
}
run_stats bench python $repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline
run_stats wgan_step python $repo/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras
run_stats sdf_train python $repo/scripts/sdf_train_bench.py
run_pmc mfma_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" python $repo/scripts/prof_targets.py mfma
run_pmc hbm_fetch "FETCH_SIZE" python $repo/scripts/prof_targets.py hbm
run_pmc hbm_write "WRITE_SIZE" python $repo/scripts/prof_targets.py hbm
python $repo/scripts/pmc_table.py $out/${tag}_mfma_counters.csv $out/mfma_sq_counters.csv
python $repo/scripts/pmc_table.py $out/${tag}_hbm_counters.csv $out/hbm_fetch_counters.csv $out/hbm_write_counters.csv
ls -la $out | head -40
