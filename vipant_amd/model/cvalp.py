"""CVALP worker: the tri-modal contrastive model both launch scripts select (`worker=CVALP`), with the
reference's build / forward / state-dict contract (cvap/model/cvalp.py:28-267).

One replica per process: the reference's three `data_parallel(head, ...)` calls per step (replicate, scatter,
thread, gather; cvalp.py:41-56) become plain calls of the local heads; the global-batch loss semantics they
implied are provided by the feature all-gather inside the loss head (vipant_amd.parallel).
"""
from __future__ import annotations

import re
import os
from collections import OrderedDict

import torch
import torch.distributed as dist
from torch import nn

from .. import ops
from ..module import build_audio_head, build_image_head, build_loss_head, build_text_head
from .helper import load_checkpoint, load_clip


class CVALP(nn.Module):
    def __init__(self, cfg, echo):
        super().__init__()
        self.cfg = cfg
        self.echo = echo

    # ------------------------------------------------------------------ forward (cvalp.py:34-62)
    def forward(self, images, audios, text, *args, **kwargs):
        kwargs = {"normalized": self.loss_head.normalized, "names": kwargs.get("names", None)}
        return self.loss_from_features(*self.features(images, audios, text, **kwargs), **kwargs)

    def features(self, images, audios, text, **kwargs):
        """The three towers (cvalp.py:37-59) -> (image, audio, text) features, None for a modality that is absent or dummy.
        Split from the loss so that the trainer can run the towers in micro-batches under one global-batch loss."""
        kwargs = {"normalized": self.loss_head.normalized, "names": kwargs.get("names", None)}
        image_features = audio_features = text_features = None
        dummy_image = images is not None and list(images.shape[1:]) == [1, 1, 1]
        side = None
        if images is not None and self.image_head is not None and not dummy_image:
            if self._frozen(self.image_head) and images.is_cuda and os.environ.get("VIPANT_TOWER_OVERLAP", "1") == "1":
                # a frozen tower has no place in the autograd tape and no dependence on the trainable tower: it runs on a
                # side stream, so its small launches (M = 50 b tokens, short last rounds) fill in beside the audio tower's
                side = self._side_stream(images.device)
                side.wait_stream(torch.cuda.current_stream(images.device))
                with torch.cuda.stream(side), torch.no_grad():
                    image_features = self.image_head(images, **kwargs)
            else:
                image_features = self.image_head(images, **kwargs)
        elif images is not None:                       # pre-computed un-normalised features
            if self.loss_head.normalized and not dummy_image:
                images = ops.l2_normalize(images)
            image_features = images
        if audios is not None and self.audio_head is not None:
            audio_features = self.audio_head(audios, **kwargs)
        dummy_text = list(text.shape[1:]) == [1] if text is not None else True
        if text is not None and self.text_head is not None and not dummy_text:
            text_features = self.text_head(text, **kwargs)
        elif text is not None:
            if self.loss_head.normalized and not dummy_text:
                text = ops.l2_normalize(text)
            text_features = text
        if side is not None:                           # join before the loss reads the image features
            torch.cuda.current_stream(images.device).wait_stream(side)
            image_features.record_stream(torch.cuda.current_stream(images.device))
        if dummy_image:
            image_features = None                      # "dummy images will be ignored"
        if dummy_text and text is not None:
            text_features = None
        return image_features, audio_features, text_features

    def loss_from_features(self, image_features, audio_features, text_features, **kwargs):
        kwargs = {"normalized": self.loss_head.normalized, "names": kwargs.get("names", None)}
        return self.loss_head(image_features, audio_features, text_features, **kwargs)

    @staticmethod
    def _frozen(head) -> bool:
        return not any(p.requires_grad for p in head.parameters())

    def _side_stream(self, device):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def encode_image(self, image, *args, **kwargs):
        return self.image_head(image, **kwargs)

    def encode_audio(self, audio, *args, **kwargs):
        return self.audio_head(audio, **kwargs)

    def encode_text(self, text, *args, **kwargs):
        return self.text_head(text, **kwargs)

    # ------------------------------------------------------------------ state (cvalp.py:82-101)
    def collect_audio_state_dict(self):
        return self.collect_state_dict()

    def collect_state_dict(self):
        return (
            (self.image_head.state_dict() if self.image_head is not None and not self.cfg.model.image.freeze
             else OrderedDict()),
            self.audio_head.state_dict(),
            (self.text_head.state_dict() if self.text_head is not None and not self.cfg.model.text.freeze
             else OrderedDict()),
            self.loss_head.state_dict(),
        )

    def report(self, gold_file=None, **kwargs):
        if self.training:
            return self.loss_head.stats(**kwargs) if hasattr(self.loss_head, "stats") else ""
        if not dist.is_initialized() or dist.get_rank() == 0:
            return self.loss_head.report(gold_file=gold_file, **kwargs)
        return ""

    # ------------------------------------------------------------------ build (cvalp.py:103-267)
    def _device(self):
        if not torch.cuda.is_available():       # parameter construction / state-dict surgery can be inspected on a CPU
            return torch.device("cpu")          # host; forward() on CPU tensors raises (no CPU fallback)
        # the entry point selects this process's GPU (train.py: LOCAL_RANK); `cfg.rank` is the GLOBAL rank and is only
        # used for logging / rank-0 duties, it must not pick the device on a multi-node run
        return torch.device("cuda", torch.cuda.current_device())

    def build(self, **kwargs):
        tunable_params = dict()
        loss_kwargs = {k: v for k, v in kwargs.items() if k in ("negatives",)}
        if self.cfg.eval:
            local_cfg, _, audio_head_sd, _, loss_head_sd = load_checkpoint(self.cfg, self.echo)
            from_scratch, image_head_sd, text_head_sd, _ = load_clip(None, self.cfg, self.echo)
            self.image_head = build_image_head(self.cfg.model.image)
            if image_head_sd is not None:
                self.image_head.copy_state_dict(image_head_sd)
            self.audio_head = build_audio_head(self.cfg.model.audio)
            if audio_head_sd is not None:
                self.audio_head.load_state_dict(audio_head_sd)
            self.text_head = build_text_head(self.cfg.model.text)
            if text_head_sd is not None:
                self.text_head.copy_state_dict(text_head_sd)
            self.loss_head = build_loss_head(self.cfg.model.loss, **loss_kwargs)
            if loss_head_sd is not None:
                self.loss_head.load_state_dict(loss_head_sd)
        elif self.cfg.running.siamese.alive:
            tunable_params = self._build_siamese_backbone(**loss_kwargs)
        else:
            tunable_params = self._build_separate_backbone(**loss_kwargs)
        self.to(self._device())
        return tunable_params

    def _tunable(self, lmodules=(), amodules=()):
        icfg, acfg, tcfg = self.cfg.model.image, self.cfg.model.audio, self.cfg.model.text
        tunable = {f"loss_head.{k}": v for k, v in self.loss_head.named_parameters()}
        if not icfg.freeze and self.image_head is not None:
            tunable.update({f"image_head.{k}": v for k, v in self.image_head.named_parameters()})
        elif self.image_head is not None:
            shared = set(amodules) | set(lmodules)
            pattern = "|".join([rf"^{m}\." for m in shared])
            tunable.update({f"image_head.{k}": v for k, v in self.image_head.named_parameters()
                            if pattern != "" and re.match(pattern, k)})
            self.echo("Freeze image encoder" + (f" (excl. shared modules: {shared})." if shared else "."))
        if not acfg.freeze:
            pattern = "|".join([rf"^{m}\." for m in amodules])
            tunable.update({f"audio_head.{k}": v for k, v in self.audio_head.named_parameters()
                            if pattern == "" or not re.match(pattern, k)})
        else:
            self.echo("Freeze audio encoder.")
        if not tcfg.freeze and self.text_head is not None:
            pattern = "|".join([rf"^{m}\." for m in lmodules])
            tunable.update({f"text_head.{k}": v for k, v in self.text_head.named_parameters()
                            if pattern == "" or not re.match(pattern, k)})
        elif self.text_head is not None:
            self.echo("Freeze text encoder.")
        return tunable

    def _init_msg(self, what, src, n_o):
        msg = f" except {n_o}" if len(n_o) > 0 else ""
        self.echo(f"Initialize {what} encoder from `{src}`{msg}.")

    def _build_siamese_backbone(self, **kwargs):
        """AT fine-tuning layout (run_bimodal_at.sh): VA-pretrained audio head, frozen CLIP text head, image head
        destroyed when `running.imagine=False` (cvalp.py:130-215)."""
        cfg = self.cfg
        local_cfg, _, audio_head_sd, _, loss_head_sd = load_checkpoint(cfg, self.echo)
        from_scratch, image_head_sd, text_head_sd, _ = load_clip(None, cfg, self.echo)
        self.image_head = build_image_head(cfg.model.image)
        if not from_scratch and not cfg.model.image.from_scratch:
            n_o, _ = self.image_head.copy_state_dict(image_head_sd)
            self._init_msg("image", "image_head", n_o)
        if cfg.running.frame_emb is not None or not cfg.running.imagine:
            self.image_head = None
            self.echo("Destory image encoder.")
        scfg = cfg.running.siamese
        amodules = set(scfg.amodules)
        skw = {"shared_modules": amodules, "reference": self.image_head, "keep_hp": scfg.keep_hp}
        self.audio_head = build_audio_head(cfg.model.audio, **skw)
        if not cfg.model.audio.from_scratch:
            if local_cfg is not None:
                n_o, _ = self.audio_head.from_pretrained(audio_head_sd, local_cfg)
                self._init_msg("audio", "audio_head", n_o)
            elif not from_scratch:
                n_o, _ = self.audio_head.copy_state_dict(image_head_sd)
                self._init_msg("audio", "image_head", n_o)
            else:
                self.echo("Have to learn from scratch.")
        ref_modules = self.audio_head.replace_modules(**skw)
        self.echo(f"A: audio_head.modules referring to image_head.modules: {ref_modules}.")
        lmodules = set(scfg.lmodules)
        skw.update({"shared_modules": lmodules})
        self.text_head = build_text_head(cfg.model.text, **skw)
        if not from_scratch and not cfg.model.text.from_scratch:
            src = text_head_sd if cfg.model.text.from_text else image_head_sd
            n_o, _ = self.text_head.copy_state_dict(src)
            self._init_msg("text", "text_head" if cfg.model.text.from_text else "image_head", n_o)
        ref_modules = self.text_head.replace_modules(**skw)
        self.echo(f"T:  text_head.modules referring to image_head.modules: {ref_modules}.")
        if cfg.running.text_emb is not None or len(self.text_head.state_dict()) == 0:
            self.text_head = None
            self.echo("Destory text encoder.")
        self.loss_head = build_loss_head(cfg.model.loss, **kwargs)
        return self._tunable(lmodules, amodules)

    def _build_separate_backbone(self, **kwargs):
        """VA pre-training layout (run_bimodal_va.sh): frozen image head, trainable audio head initialised from
        the CLIP visual tower when available (cvalp.py:217-267)."""
        cfg = self.cfg
        from_scratch, image_head_sd, text_head_sd, _ = load_clip(None, cfg, self.echo)
        self.image_head = build_image_head(cfg.model.image)
        if not from_scratch and not cfg.model.image.from_scratch:
            n_o, _ = self.image_head.copy_state_dict(image_head_sd)
            self._init_msg("image", "image_head", n_o)
        if cfg.running.frame_emb is not None:
            self.image_head = None
            self.echo("Destory image encoder.")
        self.audio_head = build_audio_head(cfg.model.audio)
        if not from_scratch and not cfg.model.audio.from_scratch:
            n_o, _ = self.audio_head.copy_state_dict(image_head_sd)
            self._init_msg("audio", "image_head", n_o)
        self.text_head = build_text_head(cfg.model.text)
        if not from_scratch and not cfg.model.text.from_scratch:
            n_o, _ = self.text_head.copy_state_dict(text_head_sd)
            self._init_msg("text", "text_head", n_o)
        if len(self.text_head.state_dict()) == 0:
            self.text_head = None
            self.echo("Destory text encoder.")
        self.loss_head = build_loss_head(cfg.model.loss, **kwargs)
        return self._tunable()
