"""Ordered kernel list of one steady-state step (the one of median wall time) in a rocprofv3 kernel-trace CSV: a step ends at the
last launch of a group of `per_step` launches of `marker` (default: the optimizer kernel) and starts after the previous group.
Prints name, duration, idle gap before."""
import csv, re, sys
path, marker = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "adam_kernel")
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 2      # marker launches per step
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
ends = [i for i, r in enumerate(rows) if marker in r[2]]
if len(ends) < 2 * per_step:
    sys.exit("marker %s seen %d times" % (marker, len(ends)))
# every complete step = the launches between two consecutive groups of `per_step` marker launches; the one printed is the step
# of MEDIAN wall time among those after the first two (warm-up), so that one slow launch (a clock dip, a page fault) in the last
# step does not become "the" timeline; min / median / max of all of them are printed with it
bounds = ends[per_step - 1::per_step]
steps = [(bounds[k], bounds[k + 1]) for k in range(len(bounds) - 1)]
cand = steps[2:] if len(steps) > 3 else steps
walls = sorted((rows[b][1] - rows[a][1], a, b) for a, b in cand)
_, prev, last = walls[len(walls) // 2]
step = rows[prev + 1:last + 1]
summary = "steps timed %d: wall min %.1f / median %.1f / max %.1f us" % (len(walls), walls[0][0] / 1e3, walls[len(walls) // 2][0] / 1e3,
                                                                            walls[-1][0] / 1e3)
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
print(summary)
