"""Encoder building blocks with the reference's registry names, constructor signatures, forward
keyword protocol and state_dict keys (cvap/module/val.py), computing through the HIP autograd nodes of
`vipant_amd.ops`.  torch.nn modules appear only as PARAMETER CONTAINERS (same shapes, names and default
initialisation as the reference, so seeds and checkpoints carry over); their forward() is never called.

Differences that are deliberate and MI355X-motivated:
  * activations are batch-first token-major [b, S, D] end to end (`TransformerBackbone.batch_first = True`), so
    MetaHead's two permute copies (clip_head.py:108-110) disappear;
  * the residual stream is fp16 inside a transformer stack (`running.stream_dtype`, the reference's own autocast precision;
    `fp32` selectable) and fp32 at the stack's boundaries; its gradient is bf16; MFMA operands are bf16, accumulation fp32.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from .. import ops
from ..registry import Registry

ENCODER_MODULES_REGISTRY = Registry("ENCODER_MODULES")
ENCODER_MODULES_REGISTRY.__doc__ = "Registry for encoder modules."


def build_encoder_module(cfg, **kwargs):
    """cvap/module/val.py:17-18."""
    return ENCODER_MODULES_REGISTRY.get(cfg.name)(cfg, **kwargs)


class LayerNorm(nn.LayerNorm):
    """Parameter container for ln_* (clip/model.py:154-160); the fp32-statistics LayerNorm itself is
    vipant_layernorm_fwd/bwd."""

    def forward(self, x):  # pragma: no cover - guard against accidental eager use
        raise RuntimeError("LayerNorm parameters are consumed by the fused HIP path; do not call the module")


class QuickGELU(nn.Module):
    """Placeholder keeping `mlp.gelu` in the module tree (clip/model.py:163-165); fused into the c_fc epilogue."""

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("QuickGELU is fused into the c_fc contraction epilogue")


class MetaEncoder(nn.Module):
    """cvap/module/val.py:20-32."""

    def __init__(self):
        super().__init__()
        self.position_resolution = None
        self.mask = None

    @property
    def hp(self):
        return []

    @hp.setter
    def hp(self, hp):
        pass


class Miscellanea(MetaEncoder):
    """Positional / class embedding container (cvap/module/val.py:34-51)."""

    def __init__(self, cfg, position_resolution=None, **kwargs):
        super().__init__()
        if position_resolution is not None:
            width = position_resolution[-1]
            self.position_resolution = position_resolution[:-1]
            positions = int(np.prod(self.position_resolution)) + 1
        else:
            self.position_resolution = None
            width, positions = 0, 0
        scale = width ** -0.5 if width > 0 else 0
        self.positional_embedding = nn.Parameter(scale * torch.randn(positions, width))
        self.class_embedding = nn.Parameter(scale * torch.randn(width))

    def initialize_parameters(self):
        pass


@ENCODER_MODULES_REGISTRY.register()
class AddonEncoder(nn.Module):
    """Identity hook between pre-encoder / backbone / post-encoder (cvap/module/val.py:53-61)."""

    def __init__(self, cfg, **kwargs):
        super().__init__()

    def forward(self, x, **kwargs):
        return x


@ENCODER_MODULES_REGISTRY.register()
class CLIPMisc(Miscellanea):
    """cvap/module/val.py:63-92."""

    def replace_modules(self, reference, keep_hp=False):
        self.positional_embedding, self.class_embedding = reference.positional_embedding, reference.class_embedding
        if not keep_hp:
            self.position_resolution = reference.position_resolution

    @property
    def hp(self):
        return [self.position_resolution]

    @hp.setter
    def hp(self, hp):
        (self.position_resolution,) = hp

    @property
    def pos_embedding(self):
        return interp_clip_vp_embedding(self.positional_embedding, self.position_resolution)

    @property
    def cls_embedding(self):
        return self.class_embedding


@ENCODER_MODULES_REGISTRY.register()
class GPTPreEncoder(MetaEncoder):
    """Token embedding + positional table (cvap/module/val.py:94-122); records the EOT index in `.mask`."""

    def __init__(self, cfg, width=512, ctx_len=77, **kwargs):
        super().__init__()
        self.position_resolution = (ctx_len, width)
        self.token_embedding = nn.Embedding(cfg.vocab_size, width)
        self.initialize_parameters()

    def initialize_parameters(self):
        nn.init.normal_(self.token_embedding.weight, std=0.02)

    @property
    def dtype(self):
        return self.token_embedding.weight.dtype

    def forward(self, x, positional_embedding=None, class_embedding=None, **kwargs):
        b, L = x.shape
        out, self.mask = ops.embed_tokens(x, self.token_embedding.weight, positional_embedding)
        return out.view(b, L, -1)


@ENCODER_MODULES_REGISTRY.register()
class GPTPostEncoder(MetaEncoder):
    """ln_final, EOT gather, text projection (cvap/module/val.py:124-146)."""
    fuses_normalization = True

    def __init__(self, cfg, width=512, embed_dim=512, **kwargs):
        super().__init__()
        scale = width ** -0.5
        self.ln = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, embed_dim))

    def initialize_parameters(self):
        pass

    def readout_rows(self, mask=None):
        """The one row per item this read-out takes: the end-of-text index the pre-encoder recorded."""
        return mask

    def forward(self, x, positional_embedding=None, class_embedding=None, mask=None, normalized=False, **kwargs):
        b, L, D = x.shape
        if L == 1:          # the stack already returned the read-out rows
            mask = None
        return ops.ReadoutFn.apply(x.reshape(b * L, D), mask, b, L,
                                   *ops.no_tape((self.ln.weight, self.ln.bias, self.proj)), bool(normalized))


def _vit_position_resolution(input_resolution, patch_size, stride):
    """Patch-grid geometry (cvap/module/val.py:148-167)."""
    stride = stride or patch_size
    stride = [stride] * 2 if isinstance(stride, int) else list(stride)
    patch_size = [patch_size] * 2 if isinstance(patch_size, int) else list(patch_size)
    if isinstance(input_resolution, int):
        nrow = ncol = input_resolution // patch_size[0]
    else:
        nrow = (input_resolution[0] - patch_size[0]) // stride[0] + 1
        ncol = (input_resolution[1] - patch_size[1]) // stride[1] + 1
    return stride, nrow * ncol + 1, (nrow, ncol)


def _bilinear_resize(t: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate(mode="bilinear", align_corners=False) for the one-off weight re-gridding at load time
    (cvap/module/val.py:173-189, 545-550).  Init-time only, never on the training path; runs on the tensor's
    device through torch (weight surgery is host logic)."""
    return torch.nn.functional.interpolate(t, tuple(int(s) for s in size), mode="bilinear", align_corners=False)


def interp_conv_weight_channel(conv_weight, input_shape):
    """cvap/module/val.py:169-180."""
    if conv_weight.shape[1] != input_shape[1]:
        shape = (conv_weight.shape[0], input_shape[1])
        conv_weight = _bilinear_resize(conv_weight.permute(2, 3, 0, 1), shape).permute(2, 3, 0, 1)
    return conv_weight


def interp_conv_weight_spatial(conv_weight, patch_shape):
    """cvap/module/val.py:182-190."""
    if tuple(conv_weight.shape[-2:]) != tuple(patch_shape):
        conv_weight = _bilinear_resize(conv_weight, patch_shape)
    return conv_weight


def interp_clip_vp_embedding(old_pos_emb, pos_resolution, old_pos_resolution=None, bop=1):
    """Re-grid a visual positional table (cvap/module/val.py:524-556); identity when sizes already agree,
    which is always the case on the training path."""
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
    new = _bilinear_resize(grid, pos_resolution).permute(0, 2, 3, 1).flatten(1, 2)
    return torch.cat((old_pos_emb[:bop], new.view(-1, pos_dim)), dim=0)


@ENCODER_MODULES_REGISTRY.register()
class ViTPreEncoder(MetaEncoder):
    """Patch conv + cls token + positional table + ln_pre (cvap/module/val.py:192-259)."""

    def __init__(self, cfg, width=768, resolution=224, **kwargs):
        super().__init__()
        self.stride, _, self.position_resolution = _vit_position_resolution(resolution, cfg.patch_size, cfg.stride)
        self.position_resolution += (width,)
        self.conv1 = nn.Conv2d(in_channels=cfg.in_channels, out_channels=width, kernel_size=cfg.patch_size,
                               stride=self.stride, bias=False)
        self.patch_size = self.conv1.weight.shape[-2:]
        self.ln = LayerNorm(width)

    def initialize_parameters(self):
        pass

    def replace_modules(self, reference, keep_hp=False):
        self.conv1, self.ln = reference.conv1, reference.ln
        if not keep_hp:
            self.stride, self.patch_size, self.position_resolution = \
                reference.stride, reference.patch_size, reference.position_resolution

    @property
    def hp(self):
        return [self.stride, self.patch_size, self.position_resolution]

    @hp.setter
    def hp(self, hp):
        (self.stride, self.patch_size, self.position_resolution) = hp

    @property
    def dtype(self):
        return self.conv1.weight.dtype

    def forward(self, x, positional_embedding=None, class_embedding=None, **kwargs):
        assert x.dim() == 4, f"expect 4d `x` but get x.dim == {x.dim()}"
        w = self.conv1.weight
        if x.shape[1] != 3:
            w = interp_conv_weight_spatial(w, self.patch_size)       # no-op unless the kernel was re-gridded
        b = x.shape[0]
        out = ops.PatchEmbedFn.apply(x.to(torch.float32), w, class_embedding, positional_embedding,
                                     self.ln.weight, self.ln.bias, tuple(self.stride))
        return out.view(b, -1, out.shape[-1])


@ENCODER_MODULES_REGISTRY.register()
class ViTPostEncoder(MetaEncoder):
    """ln_post(cls row) @ proj (cvap/module/val.py:261-290)."""
    fuses_normalization = True

    def __init__(self, cfg, width=768, embed_dim=512, **kwargs):
        super().__init__()
        scale = width ** -0.5
        self.ln = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, embed_dim))

    def initialize_parameters(self):
        pass

    def readout_rows(self, mask=None):
        """The one row per item this read-out takes: the class token."""
        return "first"

    def forward(self, x, positional_embedding=None, class_embedding=None, position_resolution=None,
                require_feature=False, normalized=False, **kwargs):
        if require_feature:
            raise NotImplementedError("require_feature (encoder-decoder captioning) is outside the contrastive path")
        b, S, D = x.shape
        return ops.ReadoutFn.apply(x.reshape(b * S, D), None, b, S,
                                   *ops.no_tape((self.ln.weight, self.ln.bias, self.proj)), bool(normalized))


class ResidualAttentionBlock(nn.Module):
    """Parameter layout of one pre-LN block (cvap/module/val.py:496-522): attn (packed in_proj + out_proj),
    ln_1, mlp.{c_fc, gelu, c_proj}, ln_2.  Compute lives in ops.BackboneFn."""

    def __init__(self, d_model: int, n_head: int, attn_mask=None, skip_attn_mask: bool = True):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", nn.Linear(d_model, d_model * 4)),
            ("gelu", QuickGELU()),
            ("c_proj", nn.Linear(d_model * 4, d_model)),
        ]))
        self.ln_2 = LayerNorm(d_model)
        self.skip_attn_mask = skip_attn_mask
        self.causal = (not skip_attn_mask) and attn_mask is not None

    def flat_params(self):
        sd = dict(self.named_parameters())
        return [sd[n] for n in ops.BLOCK_PARAM_NAMES]

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("blocks are executed by TransformerBackbone through ops.BackboneFn")


@ENCODER_MODULES_REGISTRY.register()
class TransformerBackbone(MetaEncoder):
    """cvap/module/val.py:468-494.  heads = width // 64; causal mask iff ctx_len is set and not skip_attn_mask."""

    def __init__(self, cfg, width=512, ctx_len=77, **kwargs):
        super().__init__()
        self.batch_first = True     # token-major batch-first kernels: MetaHead skips its permutes
        self.ctx_len = ctx_len
        heads = width // 64
        attn_mask = self.build_attention_mask()
        self.resblocks = nn.Sequential(*[
            ResidualAttentionBlock(width, heads, attn_mask, cfg.skip_attn_mask) for _ in range(cfg.layers)
        ])
        self.causal = (not cfg.skip_attn_mask) and attn_mask is not None
        self.grad_sync = None       # set by vipant_amd.parallel for data-parallel replicas
        self.recompute_mlp = False  # `running.recompute_mlp`: do not keep the [M, 4D] MLP activations for the backward
        self.fp8 = False            # `running.fp8_gemm`: e4m3 operands in the NT contractions of the trainable blocks (configs[4])
        self.stream_f16 = True      # `running.stream_dtype` (fp16 | fp32): residual stream inside the stack in the reference's autocast precision
        self.last_block_rows = True     # `running.last_block_rows`: honour `rows=` (the last block on the read-out rows only; exact)

    def build_attention_mask(self):
        """Marker only: the -inf upper-triangular mask (val.py:484-491) is applied inside the attention kernel."""
        return None if self.ctx_len is None else "causal"

    def forward(self, x, rows=None, **kwargs):
        """`rows` (None | "first" | int64 [b]): the one row per item the caller is going to read (MetaHead asks its post-encoder):
        the stack then returns [b, 1, D] -- those rows -- and evaluates its last block on them alone (ops.BackboneFn)."""
        b, S, D = x.shape
        if self.causal and S > self.ctx_len:
            raise ValueError(f"sequence length {S} exceeds ctx_len {self.ctx_len}")
        params = [p for blk in self.resblocks for p in blk.flat_params()]
        if not self.last_block_rows or len(params) == 0:
            rows = None
        out = ops.BackboneFn.apply(x.reshape(b * S, D), b, S, self.causal, self.grad_sync, self.recompute_mlp, self.fp8,
                                   self.stream_f16, rows, *ops.no_tape(params))
        return out.view(b, S if rows is None else 1, D)
