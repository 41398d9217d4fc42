import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd import ops
torch.manual_seed(0)
b, S, H = 1, 316, 1
qkv = (torch.randn(b * S, 3 * H * 64, device="cuda:0") * 1.5).to(torch.bfloat16)
out, lse = ops.mha_fwd(qkv, b, S, H, False)
q, k, v = qkv.double().view(b, S, 3, H, 64).permute(2, 0, 3, 1, 4)
s = (q @ k.transpose(-1, -2)) * 0.125
ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(b * S, H * 64)
err = (out.double() - ref).abs()
bad = err > 0.05
print("bad rows:", sorted(set(torch.nonzero(bad)[:, 0].tolist()))[:80])
print("bad cols:", sorted(set(torch.nonzero(bad)[:, 1].tolist())))
r = torch.nonzero(bad)[0].tolist() if bad.any() else None
if r:
    print("row", r[0], "got", out[r[0]].float().tolist()[:16], "ref", ref[r[0]].tolist()[:16])
