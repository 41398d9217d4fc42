import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd import ops
torch.manual_seed(0)
b, S, H = 1, 316, 1
D = H * 64
qkv = (torch.randn(b * S, 3 * D, device="cuda:0") * 1.0).to(torch.bfloat16)
out, lse = ops.mha_fwd(qkv, b, S, H, False)
dout = torch.randn(b * S, D, device="cuda:0").to(torch.bfloat16)
qr = qkv.double().requires_grad_()
q, k, v = qr.view(b, S, 3, H, 64).permute(2, 0, 3, 1, 4)
s = (q @ k.transpose(-1, -2)) * 0.125
ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(b * S, D)
ref.backward(dout.double())
g = qr.grad
for it in range(2):
    dqkv = ops.mha_bwd(qkv, out, dout, lse, b, S, H, False)
    for name, i in (("dq", 0), ("dk", 1), ("dv", 2)):
        got = dqkv[:, i * D:(i + 1) * D].double(); rf = g[:, i * D:(i + 1) * D]
        bad = ~torch.isfinite(got) | ((got - rf).abs() > 0.05 * rf.abs().max())
        rows = sorted(set(torch.nonzero(bad)[:, 0].tolist())); cols = sorted(set(torch.nonzero(bad)[:, 1].tolist()))
        print(it, name, "bad rows", len(rows), rows[:12], "...", rows[-6:], "bad cols", len(cols), cols[:16], "nan", int((~torch.isfinite(got)).sum()))
