"""A few launches of the attention kernels at the audio-tower shape (for rocprofv3 passes).  usage: python3 tools/attn_run.py [n]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
b, S, H = 512, 316, 12
qkv = (torch.randn(b * S, 3 * H * 64, device="cuda:0") * 0.5).to(torch.bfloat16)
out, lse = ops.mha_fwd(qkv, b, S, H, False)
dout = torch.randn_like(out)
for _ in range(n):
    ops.mha_fwd(qkv, b, S, H, False)
    ops.mha_bwd(qkv, out, dout, lse, b, S, H, False)
torch.cuda.synchronize()
