"""GPU idle time inside the timed steps from a rocprofv3 kernel trace: python tools/trace_gaps.py <kernel_trace.csv>
Prints, for the last full step, wall time, the union of kernel intervals (busy) and the largest gaps with the kernels around them."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steps are delimited by the LARS update kernel
marks = [i for i, e in enumerate(ev) if "lars_update_kernel" in e[2]]
assert len(marks) >= 3, "need at least three steps in the trace"
a, b = marks[-2] + 1, marks[-1] + 1
step = ev[a:b]
t0, t1 = step[0][0], step[-1][1]
busy, cur_s, cur_e = 0, step[0][0], step[0][1]
gaps = []
for s, e, n in step[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"step wall {(t1 - t0) / 1e6:.3f} ms, busy (union of kernels) {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms in {len(gaps)} gaps, "
      f"{len(step)} launches, sum of kernel durations {sum(e - s for s, e, _ in step) / 1e6:.3f} ms")
for g, n in sorted(gaps, reverse=True)[:12]:
    print(f"  gap {g / 1e3:8.1f} us before {n[:90]}")
