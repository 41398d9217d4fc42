"""e4m3 towers: per-parameter gradient of a small stack with the e4m3 weight-gradient contractions on / off (VIPANT_FP8_TN semantics,
switched in-process through ops.FP8_TN) against the bf16 stack."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from types import SimpleNamespace as NS
import gen
import vipant_amd.module as Mod
from vipant_amd import ops
DEV = "cuda:0"
D, layers, b, S = 768, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 31
bb = Mod.TransformerBackbone(NS(layers=layers, skip_attn_mask=True), width=D, ctx_len=None)
w = gen.det_weights("full/768", gen.backbone_shapes(D, layers))
bb.load_state_dict({k[len("encoder."):]: v for k, v in w.items()}, strict=True)
bb = bb.to(DEV)
g = torch.Generator(device=DEV); g.manual_seed(5)
x = torch.randn(b, S, D, generator=g, device=DEV); gy = torch.randn(b, S, D, generator=g, device=DEV)

def run(fp8, tn):
    bb.fp8 = fp8; ops.FP8_TN = tn
    for p in bb.parameters(): p.grad = None
    xi = x.clone().requires_grad_()
    y = bb(xi); y.backward(gy)
    return {k: p.grad.clone() for k, p in bb.named_parameters()}
g0, g1, g2 = run(False, False), run(True, False), run(True, True)
for k in g0:
    r1 = float((g1[k].double() - g0[k].double()).norm() / g0[k].double().norm())
    r2 = float((g2[k].double() - g0[k].double()).norm() / g0[k].double().norm())
    print(f"{k:45s} |bf16| {float(g0[k].norm()):10.4f}  e4m3-NT rel {r1:.3e}   e4m3-NT+TN rel {r2:.3e}  norm ratio {float(g2[k].norm() / g0[k].norm()):.4f}")
