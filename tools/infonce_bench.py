"""InfoNCE kernel group alone (loss + dx1 + dx2 + dlogit_scale) at the 8-GPU global batch: python tools/infonce_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
E, dev = 512, "cuda:0"
x1 = torch.nn.functional.normalize(torch.randn(B, E, device=dev), dim=-1).requires_grad_()
x2 = torch.nn.functional.normalize(torch.randn(B, E, device=dev), dim=-1).requires_grad_()
ls = torch.tensor(2.6593, device=dev, requires_grad=True)
for _ in range(3):
    ops.InfoNCEFn.apply(x1, x2, ls, None, 0, B, 1.0)
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.InfoNCEFn.apply(x1, x2, ls, None, 0, B, 1.0)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
ts.sort()
print(f"B={B}: median {ts[3] * 1e3:.1f} us  best {ts[0] * 1e3:.1f} us  ({6.0 * B * B * E / ts[3] / 1e9:.1f} TFLOP/s of 6 B^2 E); "
      f"workspace {ops.query('vipant_infonce_workspace_bytes', B, E) / 2**20:.1f} MiB")
