"""Attention forward alone, two builds alternating in ONE process is not possible (one library per process): run once per library,
same box, back to back:  for l in "" kvsplit; do VIPANT_HIP_LIB=... python tools/attn_fwd_ab.py; done"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd import ops
dev = "cuda:0"
b, S, H = 512, 316, 12
qkv = (torch.randn(b * S, 3 * H * 64, device=dev) * 0.5).to(torch.bfloat16)
ref = None
ts = []
for rep in range(7):
    out, lse = ops.mha_fwd(qkv, b, S, H, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.mha_fwd(qkv, b, S, H, False)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
print(os.environ.get("VIPANT_HIP_LIB", "default"), "fwd us:", " ".join("%.1f" % t for t in ts), "checksum %.6f %.6f" % (float(out.float().sum()), float(lse.sum())))
