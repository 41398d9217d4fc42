"""Timing of the retrieval-evaluation kernels (vipant_retrieval_ranks) at evaluation-set sizes.
Usage: python tools/retrieval_bench.py   ->  ms per call and the TFLOP/s of the two similarity passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
for n1, k in ((4096, 1), (20000, 1), (20000, 5)):
    n2 = n1 * k
    x1 = torch.nn.functional.normalize(torch.randn(n1, 512, device=dev), dim=-1)
    x2 = torch.nn.functional.normalize(torch.randn(n2, 512, device=dev), dim=-1)
    gold = (torch.arange(n1 * k, device=dev, dtype=torch.int32).reshape(n1, k) if k > 1
            else torch.arange(n1, device=dev, dtype=torch.int32))
    ops.retrieval_ranks(x1, x2, gold, want_top1=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.retrieval_ranks(x1, x2, gold, want_top1=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    flops = 2 * 2.0 * n1 * n2 * 3 * 512        # two passes, hi/lo split = 3x the contraction length
    print(f"N1={n1} N2={n2} G={k}: {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s (MFMA work incl. split)  "
          f"reference would hold {n1 * n2 * 12 / 2**30:.1f} GiB (fp32 sim + int64 argsort)")
