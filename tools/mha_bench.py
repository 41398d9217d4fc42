"""Attention core alone (HIP events, one stream): forward and the two-pass backward at the shapes of the towers.
usage: python tools/mha_bench.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd import ops
dev = "cuda:0"


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3


for name, b, S, H, causal in (("audio  b=512 S=316 H=12", 512, 316, 12, False), ("image  b=512 S=50  H=12", 512, 50, 12, False),
                              ("text   b=512 S=77  H=8 causal", 512, 77, 8, True), ("ViT-L  b=256 S=316 H=16", 256, 316, 16, False)):
    qkv = (torch.randn(b * S, 3 * H * 64, device=dev) * 0.5).to(torch.bfloat16)
    out, lse = ops.mha_fwd(qkv, b, S, H, causal)
    dout = torch.randn_like(out)
    fl = 4.0 * b * H * S * S * 64 * (0.5 if causal else 1.0)
    tf = t(lambda: ops.mha_fwd(qkv, b, S, H, causal))
    tb = t(lambda: ops.mha_bwd(qkv, out, dout, lse, b, S, H, causal))
    print("%-32s fwd %7.1f us (%5.2f PF/s)   bwd %7.1f us (%5.2f PF/s on 2.5x fwd work)" % (name, tf, fl / tf / 1e9, tb, 2.5 * fl / tb / 1e9))
