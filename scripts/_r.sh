cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_modules.py -x -q -k "point or sdf_generator or lnrelu" 2>&1 | tail -15
python scripts/point_gan_bench.py
for w in critic generator; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f4prof_$w -o f4 -- python scripts/point_gan_prof.py $w > gpurun_out/f4prof_$w.log 2>&1
done
python - <<'PY'
import csv, glob
for w in ("critic", "generator"):
    f = glob.glob("gpurun_out/f4prof_%s/**/*kernel_stats.csv" % w, recursive=True)
    if not f:
        print(w, "no stats"); continue
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("==", w, "total %.2f ms / 10 steps" % (tot / 1e6))
    for r in rows[:22]:
        print("%6.2f%% %6d x %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:120]))
PY
