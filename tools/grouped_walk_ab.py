"""Tile walk / schedule A/B on the two K = 768, N = 3072 launches of the step (c_fc + QuickGELU, QuickGELU'), alternating in one
process; also checks the variants against each other bit for bit.  VIPANT_GEMM_VARIANT: 0 = shipped default (DEEP schedule on the
column-grouped walk), 8388608 (bit 23) = the round-4 choice (grouped k-step for c_fc, plain DEEP for QuickGELU'), 2048 = grouped walk on the
k-step schedule everywhere, 4096 = grouped nowhere."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
M = 512 * 316
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 15


def rb(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)


x768 = rb(M, 768); w_fc = rb(3072, 768, scale=0.03)
b3072 = torch.randn(3072, device=dev)
o3072 = torch.empty(M, 3072, dtype=torch.bfloat16, device=dev)
c_out = torch.empty(M, 3072, dtype=torch.uint8, device=dev)
c_in = torch.randint(0, 256, (M, 3072), dtype=torch.uint8, device=dev)
cases = [("c_fc  ", lambda: ops.gemm_nt(x768, w_fc, o3072, bias=b3072, aux=c_out, epi=ops.EPI_QUICKGELU_D8), lambda: (o3072.clone(), c_out.clone())),
         ("dgelu8", lambda: ops.gemm_nt(x768, w_fc, o3072, aux=c_in, epi=ops.EPI_DQUICKGELU_D8), lambda: (o3072.clone(),))]
variants = ["0", "8388608", "2048", "4096"]


def one(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for name, fn, grab in cases:
    t = {v: [] for v in variants}
    ref = None
    for v in variants:
        os.environ["VIPANT_GEMM_VARIANT"] = v
        fn(); fn()
        got = grab()
        if ref is None:
            ref = got
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, got)), (name, v)
    torch.cuda.synchronize()
    for _ in range(rounds):
        for v in variants:
            os.environ["VIPANT_GEMM_VARIANT"] = v
            t[v].append(one(fn))
    print(name, "  ".join(f"{v}: {sorted(x)[len(x) // 2]:6.1f} us" for v, x in t.items()), flush=True)
