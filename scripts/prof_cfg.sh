#!/bin/bash
# kernel stats of bench.py --config <cfg> (GPU box): top kernels by time
cfg=$1; steps=${2:-2}
repo=$(pwd); out=$repo/gpurun_out/prof_cfg; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc/$cfg -o $cfg -- python $repo/bench.py --config $cfg --steps $steps --warmup 1 --no-cpu-baseline --no-extras > $out/$cfg.log 2>&1
tail -2 $out/$cfg.log | cut -c1-300
f=$(find /tmp/pc/$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${cfg}_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/${cfg}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:28]:
    print("%6.2f%% %6d calls %10.1f us avg  %s"%(float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:110]))
PY
