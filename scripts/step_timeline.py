"""Ordered kernel list of the LAST step in a rocprofv3 kernel-trace CSV: a step ends at the last launch of `marker`
(default: the optimizer kernel), starts after the previous such group.  Prints name, duration, idle gap before."""
import csv, re, sys
path, marker = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "adam_kernel")
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 2      # marker launches per step
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
ends = [i for i, r in enumerate(rows) if marker in r[2]]
if len(ends) < 2 * per_step:
    sys.exit("marker %s seen %d times" % (marker, len(ends)))
last, prev = ends[-1], ends[-1 - per_step]
step = rows[prev + 1:last + 1]
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"void |sg::|at::native::|rocprim::ROCPRIM_\d+_NS::detail::", "", n)
    return n[:78]
t0 = rows[prev][1]
busy = 0
print("step: %d launches, wall %.1f us" % (len(step), (step[-1][1] - t0) / 1e3))
for s, e, n in step:
    print("%9.1f  +%7.1f gap  %8.1f us  %s" % ((s - rows[prev][1]) / 1e3, (s - t0) / 1e3, (e - s) / 1e3, short(n)))
    busy += e - s
    t0 = e
print("busy %.1f us" % (busy / 1e3))
