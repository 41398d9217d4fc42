"""InfoNCE kernel group alone at B = 4096, E = 512, for rocprofv3 --kernel-trace --stats: 3 warm-up + 20 calls."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
B, E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 512
x1 = torch.nn.functional.normalize(torch.randn(B, E, device=dev), dim=-1).requires_grad_()
x2 = torch.nn.functional.normalize(torch.randn(B, E, device=dev), dim=-1).requires_grad_()
ls = torch.tensor(2.6593, device=dev, requires_grad=True)
for _ in range(3):
    ops.InfoNCEFn.apply(x1, x2, ls, None, 0, B, 1.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.InfoNCEFn.apply(x1, x2, ls, None, 0, B, 1.0)
e1.record()
torch.cuda.synchronize()
print(f"InfoNCE B={B}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
