"""Per-shape kernel times from a rocprofv3 --kernel-trace run: launches of one kernel name are split into duration clusters (the
audio tower's M = 161 792 launches and the frozen image tower's M = 25 600 ones share template instantiations; so do the four
long-K shapes of the DEEP schedule), each printed with its count, mean and share of a step.
usage: python3 tools/kstats_shapes.py <dir> <steps in the run (timed + warm-up)> [min_ms_per_step]"""
import csv
import glob
import re
import sys
from collections import defaultdict

d, steps = sys.argv[1], int(sys.argv[2])
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
    dur[m.group(1) if m else name[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for name, v in dur.items():
    v.sort()
    clusters, cur = [], [v[0]]
    for x in v[1:]:
        if x > 1.25 * cur[0] and x - cur[0] > 8.0:       # a new cluster: 25 % and 8 us above the cluster's smallest member
            clusters.append(cur); cur = [x]
        else:
            cur.append(x)
    clusters.append(cur)
    for c in clusters:
        rows.append((sum(c) / steps / 1e3, name, len(c), sum(c) / len(c), c[0], c[-1]))
tot = sum(r[0] for r in rows)
for ms, name, n, mean, lo, hi in sorted(rows, reverse=True):
    if ms >= floor:
        print("%-64s %5d launches  %8.1f us (%7.1f .. %7.1f)  %7.3f ms/step" % (name[:64], n, mean, lo, hi, ms))
print("sum of kernel time per step: %.3f ms" % tot)
