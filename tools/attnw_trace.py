"""Per-workgroup timeline of mha_fwd_wide_kernel (debug build, tools/build_stamps.sh): which CU ran each workgroup, when it started,
when its images had landed, when it ended -- how many workgroups a CU really overlaps, and whether load and compute phases do."""
import ctypes, os, sys, collections
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd import ops, _ffi
b, S, H = 512, 316, 12
qkv = (torch.randn(b * S, 3 * H * 64, device="cuda:0") * 0.5).to(torch.bfloat16)
for _ in range(3):
    out, lse = ops.mha_fwd(qkv, b, S, H, False)
torch.cuda.synchronize()
n = b * H
buf = (ctypes.c_ulonglong * (8192 * 4))()
lib = _ffi.lib()
lib.vipant_debug_attnw_trace.argtypes = [ctypes.c_void_p]
assert lib.vipant_debug_attnw_trace(buf) == 0
rows = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(n)]
t0 = min(r[1] for r in rows)
percu = collections.defaultdict(list)
for i, (hw, a, l, e) in enumerate(rows):
    xcc, hwid = hw >> 32, hw & 0xFFFFFFFF
    cu, sh, se = (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
    percu[(xcc & 15, se, sh, cu)].append((a - t0, l - t0, e - t0, i))
print("CUs seen:", len(percu), " workgroups per CU: min %d max %d" % (min(len(v) for v in percu.values()), max(len(v) for v in percu.values())))
print("kernel span: %.1f us" % ((max(r[3] for r in rows) - t0) / 100.0))
dur = sorted((e - a) / 100.0 for _, a, l, e in rows); ld = sorted((l - a) / 100.0 for _, a, l, e in rows)
print("workgroup lifetime us: median %.2f p10 %.2f p90 %.2f;  start -> images landed: median %.2f p10 %.2f p90 %.2f" % (
    dur[len(dur) // 2], dur[len(dur) // 10], dur[9 * len(dur) // 10], ld[len(ld) // 2], ld[len(ld) // 10], ld[9 * len(ld) // 10]))
key = sorted(percu)[3]
print("timeline of CU", key, "(start, landed, end in us; block):")
for a, l, e, i in sorted(percu[key]):
    print("  %7.2f %7.2f %7.2f  #%d" % (a / 100.0, l / 100.0, e / 100.0, i))
# concurrency: average number of resident workgroups per CU over the kernel
tot = sum(e - a for _, a, l, e in rows)
print("average resident workgroups per CU: %.2f" % (tot / (max(r[3] for r in rows) - t0) / len(percu)))
