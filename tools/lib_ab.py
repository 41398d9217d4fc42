"""Interleaved A/B of the hand-written NT contraction against the vendor library's kernel on the K = 768 wide-N shapes of the step
(VERDICT r3 item 2), same operands, same process, alternating launches, HIP-event timing, random data.  Both sides compute the PLAIN
bf16 product (no bias / activation epilogue) so that the comparison is tile against tile; our fused-epilogue launches are listed
beside them.  Measurement only: nothing in the product path calls a library GEMM.   usage: python tools/lib_ab.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
M = 512 * 316
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 9


def ev(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def rb(*s, scale=0.5):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)


for name, N, K in (("c_fc  ", 3072, 768), ("qkv   ", 2304, 768), ("c_proj", 768, 3072), ("dh1   ", 768, 2304)):
    x, w = rb(M, K), rb(N, K, scale=0.03)
    out_a = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    out_b = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ours = lambda: ops.gemm_nt(x, w, out_a, epi=ops.EPI_BF16)
    lib = lambda: torch.mm(x, w.t(), out=out_b)
    for _ in range(3):
        ours(); lib()
    torch.cuda.synchronize()
    ta, tb = [], []
    for _ in range(rounds):
        ta.append(ev(ours)); tb.append(ev(lib))
    ta.sort(); tb.sort()
    err = float((out_a.float() - out_b.float()).abs().max() / out_b.float().abs().max())
    line = f"{name} N={N:5d} K={K:5d}: ours (plain bf16 epilogue) median {ta[len(ta) // 2]:7.1f} us best {ta[0]:7.1f} | library median {tb[len(tb) // 2]:7.1f} us best {tb[0]:7.1f} | max rel diff {err:.1e}"
    if N == 3072:
        bias = torch.randn(N, device=dev)
        code = torch.empty(M, N, dtype=torch.uint8, device=dev)
        fused = lambda: ops.gemm_nt(x, w, out_a, bias=bias, aux=code, epi=ops.EPI_QUICKGELU_D8)
        fused(); torch.cuda.synchronize()
        tf = sorted(ev(fused) for _ in range(rounds))
        line += f" | ours with bias + QuickGELU + 8-bit derivative code: median {tf[len(tf) // 2]:7.1f} us"
    print(line, flush=True)
