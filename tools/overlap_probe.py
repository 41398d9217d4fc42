"""Does an HBM-bound LayerNorm kernel hide under an MFMA-bound contraction when the two are launched on different
HIP streams?  Prints: each alone, back to back on one stream, and concurrently on two streams."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
M, D = 512 * 316, 768
rb = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
x3072, x768, g3072 = rb(M, 3072), rb(M, 768), torch.empty(3072, 768, device=dev)
w_pr, o768 = rb(768, 3072) * 0.02, torch.empty(M, 768, dtype=torch.bfloat16, device=dev)
x = torch.randn(M, D, device=dev); dy = rb(M, D); dres = torch.randn(M, D, device=dev)
gamma = torch.ones(D, device=dev); mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev)
dx = torch.empty(M, D, device=dev); dxb = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
add = rb(M, D)


def tn():
    ops.gemm_tn(x3072, x768, g3072)


def nt():
    ops.gemm_nt(x3072, w_pr, o768, epi=ops.EPI_BF16)


def lnb():
    ops.layernorm_bwd(dy, x, mean, rstd, gamma, dres=dres, dx=dx, dx_bf16=dxb, dgamma=dg, dbeta=db)


def lnf():
    ops.layernorm_fwd(x, gamma, gamma, add=add, want_sum=True)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


side = torch.cuda.Stream()


def both(a, b):
    def run():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            b()
        a()
        main.wait_stream(side)
    return run


for an, a in (("tn dWfc", tn), ("nt dh2", nt)):
    for bn, b in (("ln_bwd", lnb), ("ln_fwd+add", lnf)):
        ta, tb = timeit(a), timeit(b)
        print(f"{an} {ta:7.1f} us | {bn} {tb:7.1f} us | serial {timeit(lambda: (a(), b())):7.1f} us | two streams {timeit(both(a, b)):7.1f} us")
