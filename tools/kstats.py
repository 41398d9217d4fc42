"""Print the top kernels of a rocprofv3 `--kernel-trace --stats --output-format csv` run, per step.
usage: python tools/kstats.py <dir> <launch-steps (timed + warm-up)> [rows]"""
import csv
import glob
import sys

d, steps = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 32
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    t = float(r["TotalDurationNs"])
    print("%-72s %6d %8.3f ms/step %9.1f us %5.1f%%" % (r["Name"][:72], int(r["Calls"]), t / steps / 1e6,
                                                         float(r["AverageNs"]) / 1e3, 100 * t / tot))
print("sum of kernel time per step: %.3f ms" % (tot / steps / 1e6))
