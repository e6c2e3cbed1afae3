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
run_stats hybrid_progressive python $repo/bench.py --config hybrid_progressive --steps 2 --warmup 1 --no-cpu-baseline --no-extras
run_stats hybrid_wgan python $repo/bench.py --config hybrid_wgan --steps 4 --warmup 1 --no-cpu-baseline --no-extras
run_stats point_gan python $repo/scripts/point_gan_bench.py
run_stats point_gan_critic python $repo/scripts/point_gan_prof.py critic
run_stats point_gan_generator python $repo/scripts/point_gan_prof.py generator
run_pmc mfma_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" python $repo/scripts/prof_targets.py mfma
run_pmc hbm_fetch "FETCH_SIZE" python $repo/scripts/prof_targets.py hbm
run_pmc hbm_write "WRITE_SIZE" python $repo/scripts/prof_targets.py hbm
run_pmc cfg_mfma_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" python $repo/scripts/prof_targets.py configs
run_pmc cfg_hbm_fetch "FETCH_SIZE" python $repo/scripts/prof_targets.py configs
run_pmc cfg_hbm_write "WRITE_SIZE" python $repo/scripts/prof_targets.py configs
python $repo/scripts/pmc_table.py $out/${tag}_mfma_counters.csv $out/mfma_sq_counters.csv
python $repo/scripts/pmc_table.py $out/${tag}_hbm_counters.csv $out/hbm_fetch_counters.csv $out/hbm_write_counters.csv
python $repo/scripts/pmc_table.py $out/${tag}_configs_mfma_counters.csv $out/cfg_mfma_sq_counters.csv
python $repo/scripts/pmc_table.py $out/${tag}_configs_hbm_counters.csv $out/cfg_hbm_fetch_counters.csv $out/cfg_hbm_write_counters.csv
# what this box streams (the edge-layer argument of DESIGN.md 3.2 rests on it) and how the one-channel kernels scale with the batch
cd $repo
python scripts/stream_calibration.py > $out/${tag}_stream_calibration.json 2> $out/stream.err
python scripts/edge_ab.py > $out/${tag}_edge_kernels_by_batch.json 2> $out/edge_ab.err
python scripts/point_gan_bench.py > $out/${tag}_point_gan_bench.txt 2> $out/point_gan.err
# the plane-streaming ConvT(64 -> 1) kernel alone: SQ / TCC / LDS counter groups at 64, 32 (old kernel) and 256 samples
bash scripts/convt_pmc.sh > $out/${tag}_convT_c1_counters.txt 2> $out/convt_pmc.err
# the driver's own command, and the rehearsal of its multi-GPU form on this one GPU (gloo, two ranks on cuda:0)
python bench.py > $out/${tag}_bench_line.json 2> $out/bench_line.err
SG_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-extras > $out/${tag}_bench_line_2ranks_gloo_one_gpu.json 2> $out/bench_2ranks.err
# ordered launch lists of one steady-state step
bash scripts/timeline_run.sh > $out/timeline.log 2>&1
for n in wgan sdf200k sdf20k sdf20k_graphed; do cp gpurun_out/timeline/$n.txt $out/${tag}_${n}_step_timeline.txt; done
ls -la $out | head -60
