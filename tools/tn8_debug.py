import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vipant_amd import _ffi, ops
from test_fp8_gpu import dequant_mx, rnd
DEV = "cuda:0"
for M, P, Q, vary in [(128, 128, 128, False), (128, 128, 128, True), (256, 256, 256, True), (1000, 768, 768, False), (1000, 768, 768, True)]:
    ra = torch.exp2(torch.randint(-6, 7, (M, 1), device=DEV).float()) if vary else torch.ones(M, 1, device=DEV)
    a = (rnd(M, P, seed=11) * ra).to(torch.bfloat16)
    b = (rnd(M, Q, seed=12) * ra.flip(0)).to(torch.bfloat16)
    qa, sa = ops.quant_e4m3_mx32(a); qb, sb = ops.quant_e4m3_mx32(b)
    c = torch.empty(P, Q, dtype=torch.float32, device=DEV)
    ops.gemm_tn_e4m3(qa, sa, qb, sb, c)
    da, db = dequant_mx(ops, qa, sa).double(), dequant_mx(ops, qb, sb).double()
    ref = da.t() @ db
    d = (c.double() - ref).abs()
    i = int(d.argmax()); pi, qi = i // Q, i % Q
    print(f"M={M} P={P} Q={Q} vary={vary}: max err {float(d.max() / ref.abs().max()):.3e} at ({pi},{qi}) c={float(c[pi, qi]):.6f} ref={float(ref[pi, qi]):.6f}; "
          f"rows p with err>1e-6: {int((d.amax(dim=1) / ref.abs().max() > 1e-6).sum())} cols: {int((d.amax(dim=0) / ref.abs().max() > 1e-6).sum())}")
    # per token-block contributions of the worst element
    contrib = (da[:, pi] * db[:, qi]).view(-1, 32).sum(dim=1) if M % 32 == 0 else None
    if contrib is not None:
        print("   block sums:", [f"{float(x):.4f}" for x in contrib[:8]])
    # hypothesis: products flushed when below 2^-k of the block's largest product?
