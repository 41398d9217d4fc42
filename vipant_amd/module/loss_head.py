"""Loss heads with the reference's registry / forward contract (cvap/module/decoder/loss_head.py).

Training path: `CELossHead.forward` = symmetric InfoNCE through the fused HIP kernels (ops.InfoNCEFn); in a
replica group the features are first all-gathered so every rank scores the GLOBAL batch -- the semantics of
the reference's dp mode, where the loss head sees the gathered batch (cvap/model/cvalp.py:41-61).
Eval path: `infer` caches normalised features, `report` computes retrieval top-1 / top-5 (host side).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from .. import ops, parallel
from ..registry import Registry

LOSS_HEADS_REGISTRY = Registry("LOSS_HEADS")


def build_loss_head(cfg, **kwargs):
    return LOSS_HEADS_REGISTRY.get(cfg.name)(cfg, **kwargs)


class LossHead(nn.Module):
    """cvap/module/decoder/loss_head.py:25-244 (feature cache + equal-size retrieval report)."""

    def __init__(self):
        super().__init__()
        self.reduce = False
        self.normalized = True

    def copy_state_dict(self, state_dict):
        pass

    def infer(self, x1, x2, *args, **kwargs):
        if not hasattr(self, "x1s") or not hasattr(self, "x2s") or not hasattr(self, "ids"):
            self.x1s, self.x2s, self.ids = [], [], []
        if not kwargs.get("normalized", False):
            x1, x2 = ops.l2_normalize(x1), ops.l2_normalize(x2)
        self.x1s.append(x1.detach().float().cpu())
        self.x2s.append(x2.detach().float().cpu())
        names = kwargs.get("names", None)
        if names is not None:
            self.ids.extend(names)
        return None

    @staticmethod
    def _ranks(sim: np.ndarray, gold: np.ndarray) -> np.ndarray:
        """Position of the gold column in a descending sort of each row (loss_head.py:116-118) without sorting."""
        g = sim[np.arange(sim.shape[0]), gold][:, None]
        return (sim > g).sum(1)

    def report(self, gold_file=None):
        x1s, x2s = torch.cat(self.x1s).numpy(), torch.cat(self.x2s).numpy()
        n1, n2 = x1s.shape[0], x2s.shape[0]
        if n1 == n2:
            sim = x1s @ x2s.T
            r12 = self._ranks(sim, np.arange(n1))
            r21 = self._ranks(sim.T, np.arange(n1))
            pct = lambda r, k: float((r < k).sum()) / n1 * 100.0
            p_12 = f"I->A: t1 = {pct(r12, 1):2.2f} t5 = {pct(r12, 5):2.2f}"
            p_21 = f"A->I: t1 = {pct(r21, 1):2.2f} t5 = {pct(r21, 5):2.2f}"
        elif n1 * 5 == n2:   # 1 audio vs 5 captions (loss_head.py:135-168)
            sim = x1s @ x2s.T
            r12 = np.stack([self._ranks(sim, np.arange(n1) * 5 + j) for j in range(5)], 1)
            t1 = float((r12 < 1).sum()) / n1 * 100.0
            t5 = float((r12 < 5).sum()) / (5 * n1) * 100.0
            p_12 = f"A->T: t1 = {t1:2.2f} t5 = {t5:2.2f} mR = {r12.min(1).astype(np.float32).mean() + 1:2.2f}"
            r21 = self._ranks(sim.T, np.repeat(np.arange(n1), 5))
            p_21 = (f"T->A: t1 = {float((r21 < 1).sum()) / n2 * 100.0:2.2f} t5 = {float((r21 < 5).sum()) / n2 * 100.0:2.2f} "
                    f"mR = {r21.astype(np.float32).mean() + 1:2.2f}")
        else:
            p_12, p_21 = f"{tuple(x1s.shape)}x{tuple(x2s.shape)}", "-"
        del self.x1s, self.x2s, self.ids
        return f"{p_12} {p_21} @ {n1}"


@LOSS_HEADS_REGISTRY.register()
class CELossHead(LossHead):
    """Symmetric InfoNCE with a learnable temperature (cvap/module/decoder/loss_head.py:246-284)."""

    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.logit_scale = (nn.Parameter(torch.ones([]) * np.log(1 / 0.07)) if cfg.scaling
                            else torch.ones([], requires_grad=False) * np.log(1 / 1))
        self.scale_max = cfg.scale_max or float("inf")
        self.reduce = False
        # replica groups: "global" = all-gather features, B-way negatives (reference dp semantics);
        # "local" = per-rank negatives, gradients averaged (what the reference's ddp mode would compute)
        self.negatives = kwargs.get("negatives", "global")

    def copy_state_dict(self, state_dict):
        key = "logit_scale"
        new_dict = self.state_dict()
        if key in new_dict and key in state_dict:
            new_dict.update({key: state_dict[key]})
        self.load_state_dict(new_dict)

    def forward(self, x1, x2, *args, **kwargs):
        if not self.training:
            if not dist.is_initialized() or dist.get_rank() == 0:
                return self.infer(x1, x2, *args, **kwargs)
            return None
        if not kwargs.get("normalized", False):
            x1, x2 = ops.L2NormFn.apply(x1), ops.L2NormFn.apply(x2)
        scale_max = 0.0 if self.scale_max == float("inf") else float(self.scale_max)
        ls = self.logit_scale if isinstance(self.logit_scale, nn.Parameter) else self.logit_scale.to(x1.device)
        b = x1.shape[0]
        world = parallel.world_size()
        if world > 1 and self.negatives == "global":
            x1g, x2g = parallel.all_gather_features(x1, x2)
            return ops.InfoNCEFn.apply(x1g, x2g, ls, scale_max, parallel.rank() * b, b, 1.0)
        return ops.InfoNCEFn.apply(x1, x2, ls, scale_max, 0, b, 1.0)


@LOSS_HEADS_REGISTRY.register()
class VALCELossHead(LossHead):
    """Sum of pairwise InfoNCE heads over {vision, audio, language} (loss_head.py:421-495); the AT script
    enables only `al`."""

    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.loss_head_va = self.loss_head_lv = self.loss_head_al = None
        self._total_loss = {}
        if cfg.va:
            self.loss_head_va = CELossHead(cfg, **kwargs)
            self._total_loss.update({"va": 0.})
        if cfg.lv:
            self.loss_head_lv = CELossHead(cfg, **kwargs)
            self._total_loss.update({"lv": 0.})
        if cfg.al:
            self.loss_head_al = CELossHead(cfg, **kwargs)
            self._total_loss.update({"al": 0.})

    def copy_state_dict(self, state_dict):
        pass

    def _pairs(self, x1, x2, x3):
        return (("va", self.loss_head_va, x1, x2), ("lv", self.loss_head_lv, x1, x3), ("al", self.loss_head_al, x2, x3))

    def infer(self, x1, x2, x3, *args, **kwargs):
        for _, head, a, b in self._pairs(x1, x2, x3):
            if a is not None and b is not None and head is not None:
                head.infer(a, b, *args, **kwargs)
        return 0.0

    def stats(self, nstep=1, **kwargs):
        return " ".join(f"{k} {float(v) / nstep:.3f}" for k, v in self._total_loss.items())

    def report(self, gold_file=None):
        out = []
        for tag, head in (("VA", self.loss_head_va), ("LV", self.loss_head_lv), ("AL", self.loss_head_al)):
            if head is not None:
                out.append(f"{tag}: " + head.report(gold_file))
        return "\n" + "\n".join(out).strip()

    def forward(self, x1, x2, x3, *args, **kwargs):
        """v: x1; a: x2; l: x3."""
        if not self.training:
            if not dist.is_initialized() or dist.get_rank() == 0:
                return self.infer(x1, x2, x3, *args, **kwargs)
            return None
        loss = 0.0
        for key, head, a, b in self._pairs(x1, x2, x3):
            if a is not None and b is not None and head is not None:
                term = head(a, b, *args, **kwargs)
                self._total_loss[key] += term.detach()
                loss = loss + term
        return loss


class DummyLossHead(nn.Module):
    def __init__(self, cfg, **kwargs):
        super().__init__()

    def forward(self, *args, **kwargs):
        return None
