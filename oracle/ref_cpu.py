"""CPU oracle for the VIP-ANT bimodal contrastive training step.

TEST INFRASTRUCTURE ONLY.  This module is a plain PyTorch fp32 restatement of
the reference's hot path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
the product path (`vipant_amd/`) never does and fails loudly when the HIP
library is missing.

Parity pinning: every function below is checked against outputs of the
reference itself, imported in the build container (`tests/golden/make_golden.py`
-> `tests/golden/*.npz`, compared in `tests/test_oracle_golden.py`).  The
reference ships no tests or golden vectors of its own.

Each function cites the reference file:line it follows (paths relative to the
reference repository root).  Weights are passed as flat dicts that use the
reference's state_dict key names.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------- primitives
def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """clip/model.py:154-160 -- LayerNorm computed in fp32, cast back to the input dtype."""
    orig = x.dtype
    xf = x.float()
    mu = xf.mean(-1, keepdim=True)
    var = ((xf - mu) ** 2).mean(-1, keepdim=True)
    y = (xf - mu) * torch.rsqrt(var + eps) * weight.float() + bias.float()
    return y.to(orig)


def quick_gelu(x: Tensor) -> Tensor:
    """clip/model.py:163-165."""
    return x * torch.sigmoid(1.702 * x)


def causal_mask(ctx_len: int) -> Tensor:
    """cvap/module/val.py:484-491 -- additive mask, -inf strictly above the diagonal."""
    m = torch.full((ctx_len, ctx_len), float("-inf"))
    return m.triu_(1)


def multi_head_attention(x: Tensor, sd: SD, prefix: str, n_head: int, attn_mask: Optional[Tensor]) -> Tensor:
    """cvap/module/val.py:511-517 (nn.MultiheadAttention, packed in_proj, dropout 0).

    x is batch-first [b, S, D] here; the reference runs seq-first [S, b, D], which is
    the same math on permuted storage (cvap/module/encoder/clip_head.py:108-110).
    """
    b, S, D = x.shape
    dh = D // n_head
    qkv = x @ sd[prefix + "in_proj_weight"].t() + sd[prefix + "in_proj_bias"]
    q, k, v = qkv.chunk(3, dim=-1)

    def heads(t):
        return t.reshape(b, S, n_head, dh).permute(0, 2, 1, 3)

    q, k, v = heads(q) * (dh ** -0.5), heads(k), heads(v)
    s = q @ k.transpose(-1, -2)
    if attn_mask is not None:
        s = s + attn_mask[:S, :S]
    p = torch.softmax(s.float(), dim=-1).to(x.dtype)
    o = (p @ v).permute(0, 2, 1, 3).reshape(b, S, D)
    return o @ sd[prefix + "out_proj.weight"].t() + sd[prefix + "out_proj.bias"]


def residual_attention_block(x: Tensor, sd: SD, prefix: str, n_head: int, attn_mask: Optional[Tensor]) -> Tensor:
    """cvap/module/val.py:519-522 -- pre-LN block."""
    h = layer_norm(x, sd[prefix + "ln_1.weight"], sd[prefix + "ln_1.bias"])
    x = x + multi_head_attention(h, sd, prefix + "attn.", n_head, attn_mask)
    h = layer_norm(x, sd[prefix + "ln_2.weight"], sd[prefix + "ln_2.bias"])
    h = h @ sd[prefix + "mlp.c_fc.weight"].t() + sd[prefix + "mlp.c_fc.bias"]
    h = quick_gelu(h)
    h = h @ sd[prefix + "mlp.c_proj.weight"].t() + sd[prefix + "mlp.c_proj.bias"]
    return x + h


def transformer_backbone(x: Tensor, sd: SD, prefix: str, layers: int, width: int,
                         ctx_len: Optional[int], skip_attn_mask: bool) -> Tensor:
    """cvap/module/val.py:468-494; heads = width // 64 (val.py:474)."""
    n_head = width // 64
    mask = None
    if not skip_attn_mask and ctx_len is not None:
        mask = causal_mask(ctx_len).to(x.dtype)
    for i in range(layers):
        x = residual_attention_block(x, sd, f"{prefix}resblocks.{i}.", n_head, mask)
    return x


# --------------------------------------------------------------------------- encoders
def vit_position_resolution(resolution, patch_size, stride) -> Tuple[list, int, Tuple[int, int]]:
    """cvap/module/val.py:148-167."""
    stride = stride or patch_size
    if isinstance(stride, int):
        stride = [stride] * 2
    stride = list(stride)
    if isinstance(patch_size, int):
        patch_size = [patch_size] * 2
    patch_size = list(patch_size)
    if isinstance(resolution, int):
        nrow = ncol = resolution // patch_size[0]
    else:
        nrow = (resolution[0] - patch_size[0]) // stride[0] + 1
        ncol = (resolution[1] - patch_size[1]) // stride[1] + 1
    return stride, nrow * ncol + 1, (nrow, ncol)


def interp_clip_vp_embedding(old_pos_emb: Tensor, pos_resolution, old_pos_resolution=None, bop: int = 1) -> Tensor:
    """cvap/module/val.py:524-556 -- bilinear re-gridding of a visual positional table."""
    num_pos, pos_dim = old_pos_emb.shape[-2:]
    if int(np.prod(pos_resolution)) + 1 == num_pos:
        return old_pos_emb
    if old_pos_resolution is None:
        h = w = int(np.sqrt(num_pos - bop))
    else:
        h, w = old_pos_resolution
    grid = old_pos_emb[bop:].reshape(-1, h, w, pos_dim).permute(0, 3, 1, 2)
    if tuple(grid.shape[-2:]) == tuple(pos_resolution):
        return old_pos_emb
    new = F.interpolate(grid, tuple(pos_resolution), mode="bilinear", align_corners=False)
    new = new.permute(0, 2, 3, 1).flatten(1, 2)
    return torch.cat((old_pos_emb[:bop], new.view(-1, pos_dim)), dim=0)


def interp_conv_weight_spatial(w: Tensor, patch_shape) -> Tensor:
    """cvap/module/val.py:182-190."""
    if tuple(w.shape[-2:]) != tuple(patch_shape):
        w = F.interpolate(w, tuple(patch_shape), mode="bilinear", align_corners=False)
    return w


def vit_pre_encoder(x: Tensor, sd: SD, stride: Sequence[int], pos: Tensor, cls: Tensor) -> Tensor:
    """cvap/module/val.py:228-259 -- patch conv (+channel-mean kernel for non-RGB input),
    cls token, positional table, ln_pre.  x: [b, C, T, F] -> [b, S, D]."""
    assert x.dim() == 4
    w = sd["pre_encoder.conv1.weight"]
    x = x.to(w.dtype)
    if x.shape[1] != 3:
        if x.shape[1] != w.shape[1]:
            w = w.mean(1, keepdim=True)
        x = F.conv2d(x, w, bias=None, stride=tuple(stride))
    else:
        x = F.conv2d(x, w, bias=None, stride=tuple(stride))
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    c = cls.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
    x = torch.cat([c, x], dim=1)
    x = x + pos[: x.shape[1]].to(x.dtype)
    return layer_norm(x, sd["pre_encoder.ln.weight"], sd["pre_encoder.ln.bias"])


def vit_post_encoder(x: Tensor, sd: SD) -> Tensor:
    """cvap/module/val.py:288-289 -- LN(cls row) @ proj."""
    h = layer_norm(x[:, 0, :], sd["post_encoder.ln.weight"], sd["post_encoder.ln.bias"])
    return h @ sd["post_encoder.proj"]


def gpt_pre_encoder(tokens: Tensor, sd: SD, pos: Tensor) -> Tuple[Tensor, Tensor]:
    """cvap/module/val.py:109-122 -- returns (x [b,L,D], eot index [b])."""
    eot = tokens.argmax(dim=-1)
    x = sd["pre_encoder.token_embedding.weight"][tokens]
    return x + pos[: x.shape[1]].to(x.dtype), eot


def gpt_post_encoder(x: Tensor, sd: SD, eot: Tensor) -> Tensor:
    """cvap/module/val.py:136-146."""
    h = layer_norm(x, sd["post_encoder.ln.weight"], sd["post_encoder.ln.bias"])
    return h[torch.arange(h.shape[0]), eot] @ sd["post_encoder.proj"]


def l2_normalize(x: Tensor) -> Tensor:
    """cvap/module/encoder/clip_head.py:117-118 (no eps)."""
    return x / x.norm(dim=-1, keepdim=True)


def vit_head_forward(x: Tensor, sd: SD, *, width: int, layers: int, stride: Sequence[int],
                     position_resolution: Tuple[int, int], normalized: bool = True) -> Tensor:
    """MetaHead.forward for CLIPAudioHead / CLIPImageHead (clip_head.py:98-120)."""
    pos = interp_clip_vp_embedding(sd["misc.positional_embedding"], position_resolution)
    h = vit_pre_encoder(x, sd, stride, pos, sd["misc.class_embedding"])
    h = transformer_backbone(h, sd, "encoder.", layers, width, None, True)
    h = vit_post_encoder(h, sd)
    return l2_normalize(h) if normalized else h


def text_head_forward(tokens: Tensor, sd: SD, *, width: int = 512, layers: int = 12, ctx_len: int = 77,
                      normalized: bool = True) -> Tensor:
    """MetaHead.forward for CLIPTextHead (clip_head.py:98-120, 249-292)."""
    pos = sd["misc.positional_embedding"]
    h, eot = gpt_pre_encoder(tokens, sd, pos)
    h = transformer_backbone(h, sd, "encoder.", layers, width, ctx_len, False)
    h = gpt_post_encoder(h, sd, eot)
    return l2_normalize(h) if normalized else h


# --------------------------------------------------------------------------- losses
def ce_loss_head(x1: Tensor, x2: Tensor, logit_scale: Tensor, scale_max: Optional[float] = None,
                 normalized: bool = True) -> Tensor:
    """cvap/module/decoder/loss_head.py:265-284 -- symmetric InfoNCE: sum of the two CE means."""
    if not normalized:
        x1, x2 = l2_normalize(x1), l2_normalize(x2)
    s = logit_scale.exp().clamp(max=scale_max if scale_max else float("inf"))
    z1 = s * x1 @ x2.t()
    z2 = s * x2 @ x1.t()
    labels = torch.arange(x1.shape[0])
    return F.cross_entropy(z1, labels) + F.cross_entropy(z2, labels)


def infonce_manual(x1: Tensor, x2: Tensor, logit_scale: float, scale_max: Optional[float] = None):
    """Closed-form loss and gradients of `ce_loss_head` (fp64), used to check the HIP kernel
    without autograd.  Returns (loss, dx1, dx2, dlogit_scale)."""
    a, t = x1.double(), x2.double()
    B = a.shape[0]
    s_raw = math.exp(logit_scale)
    clamped = scale_max is not None and s_raw > scale_max
    s = scale_max if clamped else s_raw
    c = a @ t.t()
    z = s * c
    rl = torch.logsumexp(z, dim=1)
    cl = torch.logsumexp(z, dim=0)
    diag = z.diagonal()
    loss = (rl - diag).mean() + (cl - diag).mean()
    dz = (torch.exp(z - rl[:, None]) + torch.exp(z - cl[None, :]) - 2 * torch.eye(B, dtype=z.dtype)) / B
    dx1 = s * dz @ t
    dx2 = s * dz.t() @ a
    dls = torch.zeros((), dtype=z.dtype) if clamped else (dz * z).sum()
    return loss, dx1, dx2, dls


def valce_loss_head(x1, x2, x3, scales: Dict[str, Tensor], *, va: bool, lv: bool, al: bool,
                    scale_max: Optional[float] = None) -> Tensor:
    """cvap/module/decoder/loss_head.py:475-495."""
    loss = 0.0
    if x1 is not None and x2 is not None and va:
        loss = loss + ce_loss_head(x1, x2, scales["va"], scale_max)
    if x1 is not None and x3 is not None and lv:
        loss = loss + ce_loss_head(x1, x3, scales["lv"], scale_max)
    if x2 is not None and x3 is not None and al:
        loss = loss + ce_loss_head(x2, x3, scales["al"], scale_max)
    return loss


def retrieval_metrics(ranks: Tensor, nsample=None, msg: str = "") -> str:
    """cvap/module/decoder/loss_head.py:68-78."""
    nsample = nsample or ranks.shape[0]
    hit = lambda k: torch.where(ranks < k)[0].shape[0] / nsample * 100.0
    med, avg = ranks.median() + 1, ranks.mean() + 1
    return f"{msg}: R@1 {hit(1):2.2f} R5 {hit(5):2.2f} R10 {hit(10):2.2f} R50 {hit(50):2.2f} MED {med:2.2f} AVG {avg:2.2f}"


def retrieval_eval(x1s: Tensor, x2s: Tensor, k: int = 5) -> str:
    """cvap/module/decoder/loss_head.py:80-107: best rank among a clip's k captions; rank of a caption's clip."""
    n = x1s.shape[0]
    pos12 = (x1s @ x2s.t()).argsort(descending=True).argsort()          # pos12[i, j] = rank of column j in row i
    ranks = pos12.reshape(n, n, k)[torch.arange(n), torch.arange(n)].min(-1)[0].float()
    msg_12 = retrieval_metrics(ranks, msg="A->T")
    pos21 = (x2s @ x1s.t()).argsort(descending=True).argsort()
    ranks = pos21[torch.arange(n * k), torch.arange(n).repeat_interleave(k)].float()
    return f"{msg_12}\n{retrieval_metrics(ranks, msg='T->A')}"


def _class_stats(top1, ids, sample_by_classname, classname_by_sample, nsample, msg) -> str:
    """cvap/module/decoder/loss_head.py:176-232 at k = 1 (nearest neighbour in the gold class or not)."""
    tp = {}
    for idx, nb in enumerate(top1):
        cname = classname_by_sample.get(ids[idx], "")
        if cname not in sample_by_classname:
            sample_by_classname[cname] = []                  # the reference's defaultdict grows on lookup
        tp.setdefault(cname, {}).setdefault(ids[idx], 0)
        tp[cname][ids[idx]] += int(ids[nb] in sample_by_classname[cname])
    p = r = p_cls = r_cls = 0.0
    for cname, per_sample in tp.items():
        nrel = len(sample_by_classname[cname])
        pc = sum(v / 1 for v in per_sample.values()); rc = sum(v / nrel for v in per_sample.values())
        p += pc; r += rc; p_cls += pc / nrel; r_cls += rc / nrel
    nclass = len(sample_by_classname)
    return (f"{msg}: P@1 {p / nsample * 100:2.2f} R@1 {r / nsample * 100:2.2f} "
            f"mAP {p_cls / nclass * 100:2.2f} mAR {r_cls / nclass * 100:2.2f}")


def retrieval_report(x1s: Tensor, x2s: Tensor, ids=None, gold_file=None) -> str:
    """cvap/module/decoder/loss_head.py:109-244 -- LossHead.report: equal-size, 1-vs-5 and shape-mismatch branches,
    per-class nearest-neighbour statistics when a gold file is given."""
    import json
    n1, n2 = x1s.shape[0], x2s.shape[0]
    ref = stats = ""
    if n1 == n2:
        labels = torch.arange(n1).unsqueeze(-1)
        ind_12 = (x1s @ x2s.t()).argsort(descending=True)
        ind_21 = (x2s @ x1s.t()).argsort(descending=True)
        r12 = torch.where(ind_12 == labels)[1]
        r21 = torch.where(ind_21 == labels)[1]
        t = lambda r, k: torch.where(r < k)[0].shape[0] / n1 * 100.0
        p12 = f"I->A: t1 = {t(r12, 1):2.2f} t5 = {t(r12, 5):2.2f}"
        p21 = f"A->I: t1 = {t(r21, 1):2.2f} t5 = {t(r21, 5):2.2f}"
        if gold_file is not None:
            by_class, by_sample = {}, {}
            with open(gold_file) as fr:
                for iline, line in enumerate(fr):
                    if iline + 1 > n1:
                        break
                    rec = json.loads(line)
                    key = " ".join(rec["labels"])
                    by_class.setdefault(key, []).append(rec["id"]); by_sample[rec["id"]] = key
            m12 = _class_stats(ind_12[:, 0].tolist(), ids, by_class, by_sample, n1, "I->A")
            m21 = _class_stats(ind_21[:, 0].tolist(), ids, by_class, by_sample, n1, "A->I")
            stats = f"\n{m12} {m21}\n"
    elif n1 * 5 == n2:
        pos12 = (x1s @ x2s.t()).argsort(descending=True).argsort()
        r12 = pos12.reshape(n1, n1, 5)[torch.arange(n1), torch.arange(n1)]                  # [n1, 5]
        t1 = (r12 < 1).sum(-1).sum() / (1 * n1) * 100.0
        t5 = (r12 < 5).sum(-1).sum() / (5 * n1) * 100.0
        p12 = f"A->T: t1 = {t1:2.2f} t5 = {t5:2.2f} mR = {r12.min(-1)[0].float().mean() + 1:2.2f}"
        pos21 = (x2s @ x1s.t()).argsort(descending=True).argsort()
        r21 = pos21[torch.arange(n2), torch.arange(n1).repeat_interleave(5)]
        q = lambda k: torch.where(r21 < k)[0].shape[0] / n2 * 100.0
        p21 = f"T->A: t1 = {q(1):2.2f} t5 = {q(5):2.2f} mR = {r21.float().mean() + 1:2.2f}"
        ref = f"\nREFERENCE\n{retrieval_eval(x1s, x2s)}"
    else:
        p12, p21 = f"{x1s.shape}x{x2s.shape}", "-"
    return f"{stats}{p12} {p21} @ {n1}{ref}"


def zero_shot_report(audios: Tensor, labels: Tensor, text: Tensor, label_map=None) -> str:
    """cvap/module/decoder/loss_head.py:371-407 -- the zero-shot branch of ClassificationHead.report."""
    ind = (audios @ text.t()).argsort(descending=True)
    predictions = ind[:, :1]
    if isinstance(label_map, dict):
        predictions = torch.tensor([label_map[x] for x in predictions.flatten().tolist()]).view(predictions.shape)
    precision = (predictions == labels.unsqueeze(-1)).sum() / audios.shape[0] * 100.0
    return f"A->T: p1 = {precision:2.2f} @ {audios.shape[0]}"


# --------------------------------------------------------------------------- worker glue
def cvalp_forward(images, audios, text, *, image_sd=None, audio_sd=None, text_sd=None,
                  audio_cfg=None, image_cfg=None, text_cfg=None, loss="ce", scales=None,
                  loss_flags=None, scale_max=None) -> Tensor:
    """cvap/model/cvalp.py:34-62 without data_parallel (one replica, full batch)."""
    image_features = audio_features = text_features = None
    dummy_image = images is not None and list(images.shape[1:]) == [1, 1, 1]
    if images is not None and image_sd is not None and not dummy_image:
        image_features = vit_head_forward(images, image_sd, **image_cfg)
    elif images is not None:
        image_features = l2_normalize(images) if not dummy_image else images
    if audios is not None and audio_sd is not None:
        audio_features = vit_head_forward(audios, audio_sd, **audio_cfg)
    dummy_text = list(text.shape[1:]) == [1] if text is not None else True
    if text is not None and text_sd is not None and not dummy_text:
        text_features = text_head_forward(text, text_sd, **text_cfg)
    elif text is not None:
        text_features = l2_normalize(text) if not dummy_text else text
    if loss == "ce":
        return ce_loss_head(image_features, audio_features, scales["logit_scale"], scale_max)
    if dummy_image:
        image_features = None
    return valce_loss_head(image_features, audio_features, text_features, scales,
                           scale_max=scale_max, **loss_flags)


# --------------------------------------------------------------------------- optimizer
def adjust_learning_rate(step: int, *, epochs: int, steps_per_epoch: int, warmup_epoch: float,
                         batch_size: int, lr_weight: float, lr_bias: float) -> Tuple[float, float]:
    """cvap/module/lars.py:9-22 -- returns (lr for ndim>1 group, lr for ndim<2 group)."""
    max_steps = epochs * steps_per_epoch
    warmup_steps = int(warmup_epoch * steps_per_epoch)
    base_lr = batch_size / 256
    if step < warmup_steps:
        lr = base_lr * step / warmup_steps
    else:
        step -= warmup_steps
        max_steps -= warmup_steps
        q = 0.5 * (1 + math.cos(math.pi * step / max_steps))
        end_lr = base_lr * 0.001
        lr = base_lr * q + end_lr * (1 - q)
    return lr * lr_weight, lr * lr_bias


def lars_step(p: Tensor, g: Tensor, mu: Tensor, lr: float, *, weight_decay: float = 1e-6,
              momentum: float = 0.9, eta: float = 0.001) -> Tuple[Tensor, Tensor]:
    """cvap/module/lars.py:43-72 for one tensor; weight decay and trust ratio only when ndim >= 2.
    Returns (new_p, new_mu)."""
    dp = g
    if p.ndim >= 2:
        dp = dp + weight_decay * p
        pn, un = torch.norm(p), torch.norm(dp)
        q = torch.where(pn > 0.0, torch.where(un > 0, eta * pn / un, torch.ones_like(pn)), torch.ones_like(pn))
        dp = dp * q
    mu = mu * momentum + dp
    return p - lr * mu, mu


# --------------------------------------------------------------------------- init helpers
def audio_head_shapes(width: int, layers: int, embed_dim: int, S: int, in_channels: int = 3,
                      patch: int = 32) -> Dict[str, Tuple[int, ...]]:
    """State-dict layout of CLIPAudioHead / CLIPImageHead."""
    shapes = {
        "pre_encoder.conv1.weight": (width, in_channels, patch, patch),
        "pre_encoder.ln.weight": (width,), "pre_encoder.ln.bias": (width,),
        "post_encoder.ln.weight": (width,), "post_encoder.ln.bias": (width,),
        "post_encoder.proj": (width, embed_dim),
        "misc.positional_embedding": (S, width), "misc.class_embedding": (width,),
    }
    for i in range(layers):
        p = f"encoder.resblocks.{i}."
        shapes.update({
            p + "attn.in_proj_weight": (3 * width, width), p + "attn.in_proj_bias": (3 * width,),
            p + "attn.out_proj.weight": (width, width), p + "attn.out_proj.bias": (width,),
            p + "ln_1.weight": (width,), p + "ln_1.bias": (width,),
            p + "mlp.c_fc.weight": (4 * width, width), p + "mlp.c_fc.bias": (4 * width,),
            p + "mlp.c_proj.weight": (width, 4 * width), p + "mlp.c_proj.bias": (width,),
            p + "ln_2.weight": (width,), p + "ln_2.bias": (width,),
        })
    return shapes
