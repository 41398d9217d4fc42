"""Attention kernels: accuracy against an fp64 reference and timing of the shipped kernels (the opt-in builds of rounds 3-4 live as text under tools/probes/).
usage: python tools/mha_check.py [tag]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd import ops
dev = "cuda:0"
tag = sys.argv[1] if len(sys.argv) > 1 else ""


def ref_attention(qkv, batch, S, H, causal):
    D = H * 64
    q, k, v = qkv.double().view(batch, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    if causal:
        s = s + torch.full((S, S), float("-inf"), device=s.device, dtype=s.dtype).triu_(1)
    p = torch.softmax(s, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(batch * S, D), torch.logsumexp(s, -1)


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
for b, S, H, causal in ((3, 316, 12, False), (2, 306, 3, False), (2, 320, 2, False), (3, 257, 2, False), (2, 77, 8, True), (2, 50, 12, False),
                        (2, 31, 12, False), (1, 160, 2, False), (2, 100, 2, True), (1, 370, 2, False)):
    qkv = (torch.randn(b * S, 3 * H * 64, device=dev) * 1.5).to(torch.bfloat16)
    out, lse = ops.mha_fwd(qkv, b, S, H, causal)
    qr = qkv.double().requires_grad_()
    ref, rlse = ref_attention(qr, b, S, H, causal)
    dout = torch.randn(b * S, H * 64, device=dev).to(torch.bfloat16)
    ref.backward(dout.double())
    dqkv = ops.mha_bwd(qkv, out, dout, lse, b, S, H, causal)
    dqkv2 = ops.mha_bwd(qkv, out, dout, lse, b, S, H, causal)
    g = qr.grad
    D = H * 64
    errs = [float((dqkv[:, i * D:(i + 1) * D].double() - g[:, i * D:(i + 1) * D]).abs().max() / g[:, i * D:(i + 1) * D].abs().max()) for i in range(3)]
    print(f"{tag} b={b} S={S} H={H} causal={causal}: fwd max err {float((out.double() - ref).abs().max()):.2e} lse err {float((lse.double() - rlse).abs().max()):.2e} "
          f"dq/dk/dv rel max err {errs[0]:.2e} {errs[1]:.2e} {errs[2]:.2e} nan={bool(torch.isnan(dqkv.float()).any())} repro={bool(torch.equal(dqkv, dqkv2))}", flush=True)

for name, b, S, H, causal in (("audio  b=512 S=316 H=12", 512, 316, 12, False), ("ViT-L  b=256 S=316 H=16", 256, 316, 16, False),
                              ("image  b=512 S=50  H=12", 512, 50, 12, False), ("text   b=512 S=77  H=8 causal", 512, 77, 8, True)):
    qkv = (torch.randn(b * S, 3 * H * 64, device=dev) * 0.5).to(torch.bfloat16)
    out, lse = ops.mha_fwd(qkv, b, S, H, causal)
    dout = torch.randn_like(out)
    fl = 4.0 * b * H * S * S * 64 * (0.5 if causal else 1.0)
    ts = sorted(t(lambda: ops.mha_fwd(qkv, b, S, H, causal)) for _ in range(3))
    tb = sorted(t(lambda: ops.mha_bwd(qkv, out, dout, lse, b, S, H, causal)) for _ in range(3))
    print("%s %-32s fwd %7.1f us (%5.2f PF/s)   bwd %7.1f us (%5.2f PF/s on 2.5x fwd work)" % (tag, name, ts[1], fl / ts[1] / 1e9, tb[1], 2.5 * fl / tb[1] / 1e9), flush=True)
