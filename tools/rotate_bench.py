"""Does a contraction run slower when its operands are fresh every launch (as in the step: ~50 GB of activations, every buffer
touched once per pass) than when one buffer set is reused (tools/gemm_bench.py)?  dh2 shape (161 792 x 768 x 3072), bf16 epilogue:
one buffer set vs 24 sets used round-robin; back-to-back launches (no idle gaps) in both cases.
Usage: python tools/rotate_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
M, N, K = 512 * 316, 768, 3072
NSETS = 24
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
xs = [(torch.randn(M, K, device=dev)).to(torch.bfloat16) for _ in range(NSETS)]
os_ = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(NSETS)]


def run(nsets, reps=96):
    for i in range(8):
        ops.gemm_nt(xs[i % nsets], w, os_[i % nsets], epi=ops.EPI_BF16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        ops.gemm_nt(xs[i % nsets], w, os_[i % nsets], epi=ops.EPI_BF16)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for _ in range(3):
    print(f"one buffer set: {run(1):.1f} us   {NSETS} sets round-robin: {run(NSETS):.1f} us   2 sets: {run(2):.1f} us")
