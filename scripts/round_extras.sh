#!/bin/bash
# Round evidence that scripts/profile_round.sh does not produce (run on the GPU box by scripts/round_evidence.sh):
#   <tag>_edge_kernels_cold.json          the four one-channel kernels timed cold AND warm (scripts/edge_cold.py)
#   <tag>_convT_forms.json                every kernel form of sg_convT3d_k4s2p1_to1_pre_impl at 256 / 128 / 64 / 48 samples, cold and warm
#   <tag>_sdfnet_bwd_ablation.txt         sdfnet_bwd_kernel inside the 200 000-point step: shipped / no row sums / no dZ stores / neither /
#                                         default-policy image stores (A/B builds of csrc/sdfnet.hip, scripts/ab_build.sh)
#   <tag>_sdfnet_counters.txt             SQ / LDS / instruction-mix / TCP counter groups of the SDFNet kernels over six 200 000-point steps
#   <tag>_dropin_loop_kernel_stats.txt    launches and kernel time per 5+1 unit of the reference's own loop body on the module surface
#   <tag>_bench_line_2ranks_strong_gloo_one_gpu.json   `python bench.py --gpus 2 --scaling strong` (rehearsal on one GPU, gloo)
tag=${1:-rXX}
repo=$(pwd); out=$repo/gpurun_out/prof_$tag; mkdir -p $out
python scripts/edge_cold.py all > $out/${tag}_edge_kernels_cold.json 2> $out/edge_cold.err
python scripts/edge_cold.py convT_forms > $out/${tag}_convT_forms.json 2> $out/convT_forms.err
for v in "nosum -DSG_ABL_NOSUM" "nostore -DSG_ABL_NOSTORE" "none -DSG_ABL_NOSUM -DSG_ABL_NOSTORE" "imgaux0 -DSG_IMG_AUX=0"; do
  set -- $v; n=$1; shift; bash scripts/ab_build.sh $n sdfnet.hip "$@" > /dev/null 2>&1
done
{ echo "# sdfnet kernels inside the 200 000-point / latent-256 auto-decoder step (rocprofv3 --kernel-trace of scripts/sdf_step_prof.py via"
  echo "# scripts/ab_run.sh; A/B builds of csrc/sdfnet.hip): base = shipped; nosum = no row-sum pieces; nostore = no dZ image stores;"
  echo "# none = neither; imgaux0 = image stores with the default cache policy instead of nt"
  bash scripts/ab_run.sh base nosum nostore none imgaux0 base 2>/dev/null | grep -E "==|sdfnet_fwd|sdfnet_bwd|gemm_nt_bigk|step:"
} > $out/${tag}_sdfnet_bwd_ablation.txt
bash scripts/kernel_pmc.sh sdfnet python scripts/prof_targets.py sdfstep > $out/${tag}_sdfnet_counters.txt 2> $out/sdfnet_pmc.err
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/dp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp -o dp -- python $repo/scripts/dropin_prof.py > $out/dropin_prof.log 2>&1
  f=$(find /tmp/dp -name "*kernel_stats.csv" | head -1); python - $f > $out/${tag}_dropin_loop_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
units = 8.0      # scripts/dropin_prof.py: 3 warm-up + 5 timed 5+1 units
print("# the body of train_wgan.py:60-84 over the module-level surface (scripts/dropin_prof.py): per 5+1 unit")
print("launches per unit %.1f   kernel ms per unit %.3f" % (sum(int(r['Calls']) for r in rows) / units, sum(float(r['TotalDurationNs']) for r in rows) / units / 1e6))
for r in sorted(rows, key=lambda r: -int(r['Calls']))[:40]:
    print("%6.1f calls/unit %9.1f us/unit  %s" % (int(r['Calls']) / units, float(r['TotalDurationNs']) / units / 1e3, r['Name'][:100]))
PY
)
grep "ms per unit" $out/dropin_prof.log >> $out/${tag}_dropin_loop_kernel_stats.txt
SG_DIST_BACKEND=gloo python bench.py --gpus 2 --scaling strong --steps 5 --warmup 2 --no-extras 2> $out/bench_2ranks_strong.err | grep "^{" > $out/${tag}_bench_line_2ranks_strong_gloo_one_gpu.json
ls -la $out | grep -E "edge_kernels_cold|convT_forms|sdfnet_bwd_ablation|sdfnet_counters|dropin_loop|strong"
