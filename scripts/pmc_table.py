"""rocprofv3 --pmc counter_collection CSV(s) -> one row per (kernel, counter): launches, median value, median dispatch ns.
usage: pmc_table.py out.csv in1.csv [in2.csv ...]"""
import collections
import csv
import sys

vals = collections.defaultdict(list)
durs = collections.defaultdict(list)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0][:110]
        vals[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        durs[(name, r["Counter_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(sys.argv[1], "w") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "counter", "launches", "median_value", "median_dispatch_ns"])
    for (name, ctr), v in sorted(vals.items()):
        d = sorted(durs[(name, ctr)])
        w.writerow([name, ctr, len(v), "%.6g" % sorted(v)[len(v) // 2], d[len(d) // 2]])
