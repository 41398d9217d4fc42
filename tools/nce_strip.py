"""InfoNCE group alone at the 8-GPU global batch: all rows (one process owns the whole batch) against the per-rank form the 8-GPU step
runs (gradients for the rank's 512-row strip only).  python tools/nce_strip.py [B] [nrows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import _ffi, ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 512
E = 512
x1 = torch.nn.functional.normalize(torch.randn(B, E, device=dev), dim=-1).requires_grad_()
x2 = torch.nn.functional.normalize(torch.randn(B, E, device=dev), dim=-1).requires_grad_()
ls = torch.tensor(2.6593, device=dev, requires_grad=True)


def timed(row0, n, reps=20):
    for _ in range(3):
        ops.InfoNCEFn.apply(x1, x2, ls, None, row0, n, 1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.InfoNCEFn.apply(x1, x2, ls, None, row0, n, 1.0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if len(sys.argv) > 3:           # "strip" | "all": one form only (for rocprofv3 --kernel-trace --stats)
    print(sys.argv[3], f"{timed(B - nrows, nrows) if sys.argv[3] == 'strip' else timed(0, B):.1f} us")
    sys.exit(0)
print(f"B={B} E={E}: all rows {timed(0, B):.1f} us; strips of {nrows}: " +
      ", ".join(f"row0={r0}: {timed(r0, nrows):.1f} us" for r0 in (0, B // 2, B - nrows)))
print("workspace bytes:", _ffi.query("vipant_infonce_workspace_bytes", B, E))
