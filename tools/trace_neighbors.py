"""Which launches sit next to the small ATen / runtime kernels of a step?  Reads a rocprofv3 --kernel-trace CSV and prints, for every
launch whose name matches `pattern`, the kernel before and after it (deduplicated, with counts).
usage: python3 tools/trace_neighbors.py <dir> <pattern>"""
import collections
import csv
import glob
import re
import sys

d, pat = sys.argv[1], re.compile(sys.argv[2])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:60]


c = collections.Counter()
for i, r in enumerate(rows):
    if pat.search(r["Kernel_Name"]):
        prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
        c[(prev, short(r["Kernel_Name"]), nxt, r.get("Grid_Size", r.get("Grid_Size_X", "?")))] += 1
for k, v in c.most_common(40):
    print(v, " | ".join(k))
