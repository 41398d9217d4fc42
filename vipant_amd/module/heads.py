"""Encoding heads: the reference's four-part abstraction pre_encoder -> backbone -> post_encoder (+ misc
parameters) with its registries, builders, state_dict keys and weight-remapping entry points
(cvap/module/encoder/clip_head.py, audio_head.py:20-134, image_head.py:17-23, text_head.py:14-20,
cvap/module/__init__.py:21-36)."""
from __future__ import annotations

import re
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from ..registry import Registry
from .val import build_encoder_module, interp_clip_vp_embedding, interp_conv_weight_spatial, _bilinear_resize

IMAGE_HEADS_REGISTRY = Registry("IMAGE_HEADS")
AUDIO_HEADS_REGISTRY = Registry("AUDIO_HEADS")
TEXT_HEADS_REGISTRY = Registry("TEXT_HEADS")


def build_image_head(cfg, **kwargs):
    return IMAGE_HEADS_REGISTRY.get(cfg.name)(cfg, **kwargs)


def build_audio_head(cfg, **kwargs):
    return AUDIO_HEADS_REGISTRY.get(cfg.name)(cfg, **kwargs)


def build_text_head(cfg, **kwargs):
    return TEXT_HEADS_REGISTRY.get(cfg.name)(cfg, **kwargs)


def _pair(x):
    return list(x) if isinstance(x, (list, tuple)) else [x, x]


def position_resolution(input_resolution, patch_size, stride):
    """cvap/module/encoder/audio_head.py:28-40 -- (nrow, ncol) of the patch grid; nrow is the time axis."""
    input_resolution, patch_size = _pair(input_resolution), _pair(patch_size)
    stride = _pair(stride or patch_size)
    nrow = (input_resolution[0] - patch_size[0]) // stride[0] + 1
    ncol = (input_resolution[1] - patch_size[1]) // stride[1] + 1
    return nrow, ncol


def load_pos_embedding(state_dict, old_dict, new_dict, key, bop, old_pos_shape, new_pos_shape, use_slice=True):
    """Adapt a stored audio positional table to a new spectrogram length (audio_head.py:89-134): identical
    grid -> as is; same number of mel columns and fewer frames -> slice (first frames, or starting at frame row 6
    when the stored clip is longer); otherwise bilinear re-gridding."""
    add_leading_dim = False
    old_pos_emb = state_dict[key]
    if old_pos_emb.dim() == 3:
        assert old_pos_emb.shape[0] == 1
        old_pos_emb = old_pos_emb.squeeze(0)
        add_leading_dim = True
    num_pos, pos_dim = old_pos_emb.shape[-2:]
    num_pos_required = int(np.prod(new_pos_shape))
    old_pos_shape, new_pos_shape = tuple(old_pos_shape), tuple(new_pos_shape)
    if new_pos_shape == old_pos_shape:
        new_pos_emb = old_pos_emb
    elif use_slice and new_pos_shape[-1] == old_pos_shape[-1] and num_pos_required + bop <= num_pos:
        extra = old_pos_shape[-2] - new_pos_shape[-2]
        if extra == 0:
            new_pos_emb = old_pos_emb[:num_pos_required + bop]
        else:
            start = 6 * old_pos_shape[-1] + bop
            new_pos_emb = torch.cat((old_pos_emb[:bop], old_pos_emb[start:start + num_pos_required]), 0)
    else:
        grid = old_pos_emb[bop:].reshape((-1,) + old_pos_shape + (pos_dim,)).permute(0, 3, 1, 2)
        new = _bilinear_resize(grid, new_pos_shape).permute(0, 2, 3, 1).flatten(1, 2)
        new_pos_emb = torch.cat((old_pos_emb[:bop], new.view(-1, pos_dim)), dim=0)
    old_dict[key] = new_pos_emb.unsqueeze(0) if add_leading_dim else new_pos_emb
    new_keys, old_keys = set(new_dict.keys()), set(old_dict.keys())
    new_dict.update(old_dict)
    return new_keys - old_keys, old_keys - new_keys


class MetaHead(nn.Module):
    """cvap/module/encoder/clip_head.py:25-120."""

    def __init__(self, cfg, **kwargs):
        super().__init__()
        kwargs.pop("keep_hp", False)
        kwargs.pop("reference", None)
        kwargs.pop("shared_modules", [])
        kwargs.update({"width": cfg.width, "embed_dim": cfg.embed_dim, "ctx_len": cfg.ctx_len,
                       "resolution": cfg.resolution})
        # construction order follows the reference so a seed yields the same from-scratch initialisation
        self.encoder = build_encoder_module(cfg.encoder, **kwargs)
        self.pre_encoder = build_encoder_module(cfg.pre_encoder, **kwargs)
        self.post_encoder = build_encoder_module(cfg.post_encoder, **kwargs)
        self.pre_encoder_addon = build_encoder_module(cfg.pre_encoder_addon, **kwargs)
        self.post_encoder_addon = build_encoder_module(cfg.post_encoder_addon, **kwargs)
        pos_res = (self.pre_encoder.position_resolution or self.encoder.position_resolution
                   or self.post_encoder.position_resolution)
        kwargs.update({"position_resolution": pos_res})
        self.misc = build_encoder_module(cfg.misc, **kwargs)

    def replace_modules(self, shared_modules=[], reference=None, keep_hp=False, **kwargs):
        """Share sub-modules with a reference head (clip_head.py:71-96)."""
        if len(shared_modules) < 1 or reference is None:
            return []
        ref_modules = []
        for module in ["encoder", "pre_encoder", "post_encoder", "misc"]:
            if module not in shared_modules:
                continue
            ref_modules.append(module)
            mine, theirs = getattr(self, module), getattr(reference, module)
            if hasattr(mine, "replace_modules"):
                mine.replace_modules(theirs, keep_hp=keep_hp)
            else:
                hp = mine.hp
                setattr(self, module, theirs)
                if keep_hp:
                    getattr(self, module).hp = hp
        return ref_modules

    def forward(self, x, *args, **kwargs):
        kwargs.update({"positional_embedding": self.misc.pos_embedding, "class_embedding": self.misc.cls_embedding,
                       "position_resolution": self.misc.position_resolution})
        x = self.pre_encoder(x, **kwargs)                        # (N, L, D)
        x = self.pre_encoder_addon(x, **kwargs)
        x = x.permute(1, 0, 2) if not self.encoder.batch_first else x
        # the read-out takes one row per item (class / end-of-text token): tell the stack, which then evaluates its last block on
        # those rows only and returns them (exact; `running.last_block_rows`).  Only when nothing sits between stack and read-out.
        rows = None
        if (self.encoder.batch_first and hasattr(self.post_encoder, "readout_rows") and not kwargs.get("require_feature", False)
                and type(self.post_encoder_addon).__name__ == "AddonEncoder"):
            rows = self.post_encoder.readout_rows(mask=self.pre_encoder.mask)
        x = self.encoder(x, **dict(kwargs, rows=rows)) if rows is not None else self.encoder(x, **kwargs)
        x = x.permute(1, 0, 2) if not self.encoder.batch_first else x
        mask = self.pre_encoder.mask
        x = self.post_encoder_addon(x, **kwargs)
        x = self.post_encoder(x, mask=mask, **kwargs)
        if kwargs.get("normalized", False) and not getattr(self.post_encoder, "fuses_normalization", False):
            from .. import ops
            x = ops.l2_normalize(x)
        return x


def _remap_clip_visual(state_dict):
    """CLIP VisualTransformer keys -> head keys (clip_head.py:128-144 / 195-211)."""
    pre_keys, post_keys, misc_keys = {"conv1.weight"}, {"proj"}, {"positional_embedding", "class_embedding"}
    out = OrderedDict()
    for k, v in state_dict.items():
        if k in pre_keys:
            k = f"pre_encoder.{k}"
        elif k in post_keys:
            k = f"post_encoder.{k}"
        elif k in misc_keys:
            k = f"misc.{k}"
        else:
            k = re.sub(r"^transformer\.", "encoder.", k)
            k = re.sub(r"^ln_pre\.", "pre_encoder.ln.", k)
            k = re.sub(r"^ln_post\.", "post_encoder.ln.", k)
        out[k] = v
    return out


def _finish_load(head, old_dict):
    new_dict = head.state_dict()
    new_keys, old_keys = set(new_dict.keys()), set(old_dict.keys())
    new_dict.update(old_dict)
    head.load_state_dict(new_dict)
    return new_keys - old_keys, old_keys - new_keys


class CLIPImageHead(MetaHead):
    """cvap/module/encoder/clip_head.py:122-166 (TransformerBackbone branch)."""

    def copy_state_dict(self, state_dict):
        return _finish_load(self, _remap_clip_visual(state_dict))


class CLIPAudioHead(MetaHead):
    """cvap/module/encoder/clip_head.py:168-247."""

    def from_pretrained(self, state_dict, cfg, *args, **kwargs):
        """Initialise from a VA-pretrained audio head; re-slice / re-grid its positional table (:172-191)."""
        key = "misc.positional_embedding"
        new_dict = self.state_dict()
        old_dict = {k: v for k, v in state_dict.items() if k != key}
        new_pos_shape = self.misc.position_resolution
        old_pos_shape = position_resolution(cfg.model.audio.resolution, cfg.model.audio.pre_encoder.patch_size,
                                            cfg.model.audio.pre_encoder.stride)
        if state_dict[key].shape[0] in {50, 197}:
            state_dict[key] = interp_clip_vp_embedding(state_dict.pop(key), old_pos_shape)
        n_o, o_n = load_pos_embedding(state_dict, old_dict, new_dict, key, 1, old_pos_shape, new_pos_shape)
        self.load_state_dict(new_dict)
        return n_o, o_n

    def copy_state_dict(self, state_dict):
        """CLIP visual tower -> audio head: key remap, positional grid and conv kernel re-gridding (:193-247)."""
        old_dict = _remap_clip_visual(state_dict)
        pos_key = "misc.positional_embedding"
        old_dict[pos_key] = interp_clip_vp_embedding(old_dict.pop(pos_key), self.misc.position_resolution)
        new_dict = self.state_dict()
        conv_key = "pre_encoder.conv1.weight"
        conv_weight = interp_conv_weight_spatial(old_dict[conv_key], new_dict[conv_key].shape[-2:])
        use_mean = new_dict[conv_key].shape[1] != 1
        old_dict[conv_key] = conv_weight if use_mean else conv_weight.mean(1, keepdim=True)
        return _finish_load(self, old_dict)


class CLIPTextHead(MetaHead):
    """cvap/module/encoder/clip_head.py:249-292."""

    def initialize_parameters(self):
        pass

    def copy_state_dict(self, state_dict):
        pre_keys, misc_keys = {"token_embedding.weight"}, {"positional_embedding"}
        old_dict = OrderedDict()
        for k, v in state_dict.items():
            if k in pre_keys:
                k = f"pre_encoder.{k}"
            elif k in misc_keys:
                k = f"misc.{k}"
            else:
                k = re.sub(r"^transformer\.", "encoder.", k)
                k = re.sub(r"^ln_final\.", "post_encoder.ln.", k)
                k = re.sub(r"^text_projection", "post_encoder.proj", k)
            old_dict[k] = v
        new_dict = self.state_dict()
        pos_key = "misc.positional_embedding"
        old_num, new_num = old_dict[pos_key].shape[0], new_dict[pos_key].shape[0]
        if old_num >= new_num:
            old_dict[pos_key] = old_dict.pop(pos_key)[:new_num]
        else:
            new_dict[pos_key][:old_num] = old_dict.pop(pos_key)
            old_dict[pos_key] = new_dict[pos_key]
        return _finish_load(self, old_dict)


class DummyHead(nn.Module):
    """cvap/module/__init__.py:21-36 -- stands in for an absent modality (VA script: +model/text=dummy)."""

    def __init__(self, cfg, **kwargs):
        super().__init__()

    def from_pretrained(self, state_dict, cfg, *args, **kwargs):
        pass

    def copy_state_dict(self, state_dict):
        return {}, {}

    def replace_modules(self, **kwargs):
        return []

    def forward(self, x, *args, **kwargs):
        return None


for _reg in (IMAGE_HEADS_REGISTRY, AUDIO_HEADS_REGISTRY, TEXT_HEADS_REGISTRY):
    _reg.register(DummyHead)
IMAGE_HEADS_REGISTRY.register(CLIPImageHead)
AUDIO_HEADS_REGISTRY.register(CLIPAudioHead)
TEXT_HEADS_REGISTRY.register(CLIPTextHead)
