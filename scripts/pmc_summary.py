"""Summarises a rocprofv3 --pmc counter_collection CSV for kernels whose name contains argv[2]."""
import csv, sys, collections
d = collections.defaultdict(list); dur = []
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("dispatch ns (median)", sorted(dur)[len(dur) // 2])
for k, v in sorted(d.items()):
    print("%-28s %.4g" % (k, sorted(v)[len(v) // 2]))
