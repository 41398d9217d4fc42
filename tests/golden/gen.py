"""Deterministic synthetic weights / inputs shared by `make_golden.py` (which feeds them to the
imported reference) and by the parity tests (which feed them to the oracle and the HIP path).

Fixtures therefore hold only the reference's OUTPUTS plus checksums of these generated inputs;
the inputs themselves are regenerated from names, never stored.  Everything is CPU
`torch.Generator` based, so it is identical in the build container and on the GPU box (same image).
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import torch


def det_randn(name: str, shape, scale: float = 1.0, shift: float = 0.0) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(name.encode()) + 1213)
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32) * scale + shift


def checksum(t: torch.Tensor) -> Tuple[float, float]:
    t = t.double()
    return float(t.sum()), float((t * t).sum())


def vit_head_shapes(width: int, layers: int, embed_dim: int, S: int, in_channels: int = 3,
                    patch: int = 32) -> Dict[str, Tuple[int, ...]]:
    shapes = {
        "pre_encoder.conv1.weight": (width, in_channels, patch, patch),
        "pre_encoder.ln.weight": (width,), "pre_encoder.ln.bias": (width,),
        "post_encoder.ln.weight": (width,), "post_encoder.ln.bias": (width,),
        "post_encoder.proj": (width, embed_dim),
        "misc.positional_embedding": (S, width), "misc.class_embedding": (width,),
    }
    shapes.update(backbone_shapes(width, layers))
    return shapes


def text_head_shapes(width: int, layers: int, embed_dim: int, ctx_len: int = 77,
                     vocab: int = 49408) -> Dict[str, Tuple[int, ...]]:
    shapes = {
        "pre_encoder.token_embedding.weight": (vocab, width),
        "post_encoder.ln.weight": (width,), "post_encoder.ln.bias": (width,),
        "post_encoder.proj": (width, embed_dim),
        "misc.positional_embedding": (ctx_len + 1, width), "misc.class_embedding": (width,),
    }
    shapes.update(backbone_shapes(width, layers))
    return shapes


def backbone_shapes(width: int, layers: int) -> Dict[str, Tuple[int, ...]]:
    shapes = {}
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


def det_weights(tag: str, shapes: Dict[str, Tuple[int, ...]]) -> Dict[str, torch.Tensor]:
    """Random but well-conditioned weights: matrices ~ N(0, 1/fan_in), LN gains ~ 1 +- 0.1,
    biases / embeddings small.  Scales are chosen so activations stay O(1) through 12 layers."""
    out = {}
    for k, shp in shapes.items():
        name = f"{tag}/{k}"
        if k.endswith("ln.weight") or k.endswith("ln_1.weight") or k.endswith("ln_2.weight"):
            out[k] = det_randn(name, shp, 0.1, 1.0)
        elif k.endswith("bias"):
            out[k] = det_randn(name, shp, 0.02)
        elif k.endswith("conv1.weight"):
            fan_in = shp[2] * shp[3]          # effective 1-channel kernel after the channel mean
            out[k] = det_randn(name, shp, fan_in ** -0.5)
        elif k.endswith("token_embedding.weight"):
            out[k] = det_randn(name, shp, 0.02)
        elif k.endswith("positional_embedding") or k.endswith("class_embedding"):
            out[k] = det_randn(name, shp, shp[-1] ** -0.5)
        elif k.endswith("proj") and len(shp) == 2 and "post_encoder" in k:
            out[k] = det_randn(name, shp, shp[0] ** -0.5)
        elif len(shp) == 2:
            out[k] = det_randn(name, shp, shp[1] ** -0.5)
        else:
            out[k] = det_randn(name, shp, 0.02)
    return out


def det_tokens(tag: str, b: int, L: int = 77, lo: int = 8) -> torch.Tensor:
    """Synthetic CLIP token rows: [SOT, ids..., EOT, 0...]; EOT is the row max."""
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(tag.encode()) + 77)
    lens = torch.randint(lo, L + 1, (b,), generator=g)
    lens[0] = L                                     # at least one row fills the batch width
    toks = torch.zeros(b, L, dtype=torch.int64)
    for i in range(b):
        n = int(lens[i])
        toks[i, 0] = 49406
        toks[i, 1:n - 1] = torch.randint(1, 49406, (n - 2,), generator=g)
        toks[i, n - 1] = 49407
    return toks
