"""Reads a rocprofv3 kernel-trace CSV and reports GPU busy time vs wall time between the first and last kernel."""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2     # look at the second half (steady state)
rows = rows[skip:]
busy = sum(e - s for s, e, _ in rows)
wall = rows[-1][1] - rows[0][0]
gaps = [(rows[i + 1][0] - rows[i][1], rows[i][2][:60], rows[i + 1][2][:60]) for i in range(len(rows) - 1)]
print("kernels %d  wall %.3f ms  busy %.3f ms  idle %.1f%%" % (len(rows), wall / 1e6, busy / 1e6, 100 * (1 - busy / wall)))
gaps.sort(reverse=True)
print("mean gap %.1f us" % (sum(g[0] for g in gaps) / len(gaps) / 1e3))
for g in gaps[:12]:
    print("%8.1f us  after %s  before %s" % (g[0] / 1e3, g[1], g[2]))
