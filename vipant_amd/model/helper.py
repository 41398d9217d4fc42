"""Checkpoint / CLIP weight loading with the reference's return contracts (cvap/model/helper.py:10-50)."""
from __future__ import annotations

import os
from collections import OrderedDict

import torch

__all__ = ["load_checkpoint", "load_clip"]

_CLIP_FILES = {"ViT-B32": "ViT-B-32.pt", "ViT-B/32": "ViT-B-32.pt", "ViT-B16": "ViT-B-16.pt", "ViT-B/16": "ViT-B-16.pt"}


def load_checkpoint(cfg, echo):
    """-> (local_cfg, image_sd, audio_sd, text_sd, loss_sd); `model` in the .pth is a 2- or 4-tuple
    (cvap/model/helper.py:10-30, save format cvap/monitor/cvalp.py:302-309)."""
    model_file = f"{cfg.model_root}/{cfg.model_name}/{cfg.model_file}"
    if not os.path.isfile(model_file):      # the reference degrades to from-scratch when there is no file (helper.py:15-17)
        echo(f"Failed to load the checkpoint `{model_file}`")
        return (None,) * 5
    # an existing file that cannot be unpickled is an error, not a reason to train from scratch silently (reference
    # .pth files pickle an omegaconf DictConfig: reading them needs omegaconf importable)
    checkpoint = torch.load(model_file, map_location="cpu", weights_only=False)
    echo(f"Loading from {model_file}")
    local_cfg = _as_config(checkpoint["cfg"])
    nmodule = len(checkpoint["model"])
    if nmodule == 2:
        audio_head_sd, loss_head_sd = checkpoint["model"]
        return local_cfg, None, audio_head_sd, None, loss_head_sd
    if nmodule == 4:
        image_head_sd, audio_head_sd, text_head_sd, loss_head_sd = checkpoint["model"]
        return local_cfg, image_head_sd, audio_head_sd, text_head_sd, loss_head_sd
    raise ValueError(f"I don't know how to parse the checkpoint: # module is {nmodule}.")


def _as_config(obj):
    """The stored run config with attribute access: this repo saves a plain dict (Monitor.save), the reference an
    omegaconf DictConfig (cvap/monitor/cvalp.py:302-309)."""
    from ..config import Config, to_config
    if isinstance(obj, Config):
        return obj
    if not isinstance(obj, dict):
        try:
            from omegaconf import OmegaConf
            obj = OmegaConf.to_container(obj, resolve=True)
        except ImportError:
            return obj
    return to_config(obj)


def _read_clip_state_dict(path):
    try:
        return torch.jit.load(path, map_location="cpu").state_dict()      # OpenAI releases are TorchScript archives
    except Exception:
        obj = torch.load(path, map_location="cpu", weights_only=False)
        return obj.get("state_dict", obj) if isinstance(obj, dict) else obj.state_dict()


def load_clip(local_cfg, cfg, echo):
    """-> (from_scratch, image_head_sd, text_head_sd, model).  Splits a CLIP state dict into the visual tower
    (keys without the `visual.` prefix) and everything else except `logit_scale` (cvap/model/helper.py:32-50).
    No network here: a missing file means from-scratch initialisation, exactly as the reference degrades."""
    try:
        rcfg = cfg.running
        fname = _CLIP_FILES.get(rcfg.clip_model_name, f"{rcfg.clip_model_name}.pt")
        path = os.path.join(rcfg.clip_model_root, fname)
        sd = _read_clip_state_dict(path)
        image_head_sd = OrderedDict((k[len("visual."):], v.float()) for k, v in sd.items() if k.startswith("visual."))
        text_head_sd = OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in sd.items()
                                   if not k.startswith("visual") and k not in ("logit_scale", "input_resolution",
                                                                               "context_length", "vocab_size"))
        if local_cfg is not None:
            image_head_sd = None
        return False, image_head_sd, text_head_sd, None
    except Exception as e:
        echo(f"Will learn from scratch because: {e}")
        return True, None, None, None
