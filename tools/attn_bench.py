"""Micro-benchmark of the attention kernels at the VA-step shape (b=512, S=316, H=12)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
b, S, H = 512, 316, 12
D = H * 64
qkv = (torch.randn(b * S, 3 * D, device=dev) * 1.0).to(torch.bfloat16)
dout = torch.randn(b * S, D, device=dev).to(torch.bfloat16)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


out, lse = ops.mha_fwd(qkv, b, S, H, False)
fl = 4.0 * b * H * S * S * 64
t = timeit(lambda: ops.mha_fwd(qkv, b, S, H, False))
print(f"fwd {t * 1e3:8.1f} us  {fl / t / 1e9:7.1f} TFLOP/s")
t = timeit(lambda: ops.mha_bwd(qkv, out, dout, lse, b, S, H, False))
print(f"bwd {t * 1e3:8.1f} us  {2.5 * fl / t / 1e9:7.1f} TFLOP/s (5 products)")
