"""Gradient accuracy of the two InfoNCE paths (row-block kernels / 256 x 256 tile kernels) against an fp64 autograd reference of the
definition: rel-L2 error of dx1, dx2, |loss error|.  Usage: python tools/nce_accuracy.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
for B in (8, 64, 512):
    g = torch.Generator().manual_seed(B)
    a = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1)
    t = torch.nn.functional.normalize(torch.randn(B, 512, generator=g) + 0.7 * a, dim=-1)
    ad, td = a.double().requires_grad_(), t.double().requires_grad_()
    s = torch.tensor(2.6593).double().exp()
    z = s * ad @ td.t()
    lab = torch.arange(B)
    ref = torch.nn.functional.cross_entropy(z, lab) + torch.nn.functional.cross_entropy(z.t(), lab)
    ref.backward()
    for path in ("rows", "tiles"):
        os.environ["VIPANT_NCE_ROWS"] = "1" if path == "rows" else "0"
        a_, t_ = a.to(dev).requires_grad_(), t.to(dev).requires_grad_()
        ls = torch.tensor(2.6593, device=dev, requires_grad=True)
        loss = ops.InfoNCEFn.apply(a_, t_, ls, None, 0, B, 1.0)
        loss.backward()
        e1 = float((a_.grad.cpu().double() - ad.grad).norm() / ad.grad.norm())
        e2 = float((t_.grad.cpu().double() - td.grad).norm() / td.grad.norm())
        print(f"B={B:4d} {path:5s}: |loss err| {abs(float(loss) - float(ref)):.2e}   dx1 rel-L2 {e1:.2e}   dx2 rel-L2 {e2:.2e}")
