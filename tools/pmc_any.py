"""Per-kernel averages of every counter in a rocprofv3 --pmc pass (largest-grid launches of each kernel).
usage: python3 tools/pmc_any.py <dir> [name-filter]"""
import csv, glob, os, re, sys
from collections import defaultdict
folder = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
per = defaultdict(lambda: defaultdict(dict)); grid = {}
for cfile in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(cfile)):
        name = r["Kernel_Name"]
        if flt not in name: continue
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
        short = m.group(1) if m else name[:60]
        key = (cfile, r["Dispatch_Id"])
        per[short][key][r["Counter_Name"]] = float(r["Counter_Value"])
        grid[key] = int(r.get("Grid_Size", 0) or 0)
for short, disp in sorted(per.items()):
    gmax = max(grid[i] for i in disp)
    ids = [i for i in disp if grid[i] >= 0.8 * gmax]
    ctrs = sorted({c for i in ids for c in disp[i]})
    print(short, "launches", len(ids))
    for c in ctrs:
        v = [disp[i][c] for i in ids if c in disp[i]]
        print("   %-32s %14.0f" % (c, sum(v) / len(v)))
