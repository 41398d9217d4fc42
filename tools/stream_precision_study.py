"""Numerics study (CPU, emulation): what would a bf16 residual stream cost in parity?

Three arithmetic models of the audio tower + InfoNCE on identical weights / inputs, all built from the CPU oracle's
functions (oracle/ref_cpu.py) with rounding points inserted:
  A  fp32 everywhere                          -- the reference CPU path (the parity target)
  B  the shipped HIP numerics                 -- every contraction operand and every stored branch activation rounded to
                                                 bf16 (forward and backward), fp32 accumulation, fp32 residual stream and
                                                 fp32 gradient stream, fp32 LayerNorm / softmax / loss
  C  B + bf16 residual stream                 -- x <- bf16(x + branch) after every residual add, and the gradient stream
                                                 rounded to bf16 at the same points (what the reference's fp16 autocast does
                                                 with 3 more mantissa bits: clip/model.py:157-160, cvap/module/val.py:253-257)
  D  B + bf16 GRADIENT stream only            -- forward as B (loss and features unchanged); the gradient of the residual
                                                 stream is rounded to bf16 after every residual-gradient add
  E  D + fp16 FORWARD stream                  -- x <- fp16(x + branch) after every residual add: the reference's own autocast
                                                 stream precision (clip/model.py:157-160), LayerNorm statistics in fp32
Usage: python tools/stream_precision_study.py [b] [T] [F] [layers]
Results are recorded in profiles/r2_stream_precision.md.
"""
import os
import sys

import torch
import torch.nn.functional as Fn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bwd):
        ctx.bwd = bwd
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return (g.bfloat16().float() if ctx.bwd else g), None


class _RoundF16(torch.autograd.Function):    # forward rounded to fp16 (the reference's autocast stream), gradient to bf16
    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class _RoundBwd(torch.autograd.Function):       # forward untouched, gradient rounded to bf16
    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def rnd(x, on=True, bwd=True):
    if on == "grad":
        return _RoundBwd.apply(x)
    if on == "f16":
        return _RoundF16.apply(x)
    return _Round.apply(x, bwd) if on else x


def linear(x, w, b, q):
    y = rnd(x, q) @ rnd(w, q).t()
    return y + b if b is not None else y


def block(x, sd, p, H, q, qs):
    h = rnd(R.layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]), q)
    b, S, D = x.shape
    qkv = rnd(linear(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"], q), q)
    qh, kh, vh = (t.reshape(b, S, H, 64).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
    s = (qh * 0.125) @ kh.transpose(-1, -2)
    pr = rnd(torch.softmax(s, dim=-1), q)
    o = rnd((pr @ vh).permute(0, 2, 1, 3).reshape(b, S, D), q)
    y1 = rnd(linear(o, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], q), q)
    x = rnd(x + y1, qs)
    h = rnd(R.layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]), q)
    u = rnd(linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"], q), q)
    g = rnd(R.quick_gelu(u), q)
    y2 = rnd(linear(g, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"], q), q)
    return rnd(x + y2, qs)


def tower(aud, sd, layers, stride, pr, q, qs):
    pos = R.interp_clip_vp_embedding(sd["misc.positional_embedding"], pr)
    w = sd["pre_encoder.conv1.weight"].mean(1, keepdim=True)
    x = Fn.conv2d(rnd(aud, q, False), rnd(w, q), stride=tuple(stride))
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    c = sd["misc.class_embedding"] + torch.zeros(x.shape[0], 1, x.shape[-1])
    x = torch.cat([c, x], dim=1) + pos[: x.shape[1] + 1]
    x = rnd(R.layer_norm(x, sd["pre_encoder.ln.weight"], sd["pre_encoder.ln.bias"]), qs)
    for i in range(layers):
        x = block(x, sd, f"encoder.resblocks.{i}.", 12, q, qs)
    h = rnd(R.layer_norm(x[:, 0, :], sd["post_encoder.ln.weight"], sd["post_encoder.ln.bias"]), q)
    return R.l2_normalize(h @ rnd(sd["post_encoder.proj"], q))


def main():
    b, T, Fq, L = (int(v) for v in (sys.argv[1:5] + [32, 256, 64, 12][len(sys.argv) - 1:]))
    stride, S, pr = R.vit_position_resolution([T, Fq], 32, [16, 24])
    w = gen.det_weights(f"study/{L}/{S}", gen.vit_head_shapes(768, L, 512, S))
    aud = gen.det_randn("study/aud", (b, 1, T, Fq))
    img = R.l2_normalize(gen.det_randn("study/img", (b, 512)))
    out = {}
    for name, q, qs in (("A fp32", False, False), ("B bf16 operands, fp32 stream", True, False), ("C bf16 operands, bf16 stream", True, True),
                        ("D bf16 operands, fp32 stream, bf16 GRADIENT stream", True, "grad"),
                        ("E bf16 operands, fp16 stream (fp32 LN statistics), bf16 GRADIENT stream", True, "f16")):
        sd = {k: v.clone().requires_grad_() for k, v in w.items()}
        ls = torch.tensor(2.6592600, requires_grad=True)
        feat = tower(aud, sd, L, stride, pr, q, qs)
        loss = R.ce_loss_head(img, feat, ls)
        loss.backward()
        out[name] = (float(loss), feat.detach(), {k: v.grad.clone() for k, v in sd.items()}, float(ls.grad))
    la, fa, ga, da = out["A fp32"]
    print(f"b={b} {T}x{Fq} S={S} L={L}: loss(A) = {la:.6f}")
    for name in list(out)[1:]:
        l, f, g, d = out[name]
        gn = torch.tensor([float(g[k].norm() / ga[k].norm()) for k in g])
        ge = torch.tensor([float((g[k] - ga[k]).norm() / ga[k].norm()) for k in g])
        print(f"  {name}: |loss - A| = {abs(l - la):.2e}; feature max err / max = {float((f - fa).abs().max() / fa.abs().max()):.2e}; "
              f"min cos = {float(Fn.cosine_similarity(f, fa, dim=-1).min()):.6f}; grad-norm ratio in [{float(gn.min()):.4f}, {float(gn.max()):.4f}]; "
              f"grad rel-L2 err median {float(ge.median()):.3e} max {float(ge.max()):.3e}; dlogit_scale rel err {abs(d - da) / abs(da):.2e}")


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
