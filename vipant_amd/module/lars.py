"""LARS optimizer + learning-rate schedule of the reference (cvap/module/lars.py), stepping through the fused
multi-tensor HIP kernel (vipant_lars_step): two launches per step instead of ~10 per tensor."""
from __future__ import annotations

import math

import torch

from .. import ops, parallel

__all__ = ["exclude_bias_or_norm", "adjust_learning_rate", "LARS"]


def exclude_bias_or_norm(p):
    return p.ndim < 2


def adjust_learning_rate(cfg, optimizer, dataloader, step):
    """Linear warm-up then cosine decay to 0.1 % of base; base = batch / 256 (cvap/module/lars.py:9-22).

    `cfg.batch_size` (= `running.batch_size`) is what ONE process's loader yields.  In the reference's dp mode that
    process feeds every GPU, so it is the whole batch; with one replica per GPU the whole batch is `world` times it, and
    the schedule follows the whole batch -- N replicas take exactly the step of one process fed the concatenated batch."""
    max_steps = cfg.epochs * len(dataloader)
    warmup_steps = int(cfg.warmup_epoch * len(dataloader))
    base_lr = cfg.batch_size * parallel.world_size() / 256
    if step < warmup_steps:
        lr = base_lr * step / warmup_steps
    else:
        step -= warmup_steps
        max_steps -= warmup_steps
        q = 0.5 * (1 + math.cos(math.pi * step / max_steps))
        end_lr = base_lr * 0.001
        lr = base_lr * q + end_lr * (1 - q)
    optimizer.param_groups[0]["lr"] = lr * cfg.lr_weight
    optimizer.param_groups[1]["lr"] = lr * cfg.lr_bias


class LARS(torch.optim.Optimizer):
    """Same constructor and per-tensor rule as cvap/module/lars.py:24-72; `step()` gathers every tensor that has a
    gradient into one vipant_lars_step call.  Momentum buffers live in `self.state[p]['mu']` as in the reference."""

    def __init__(self, params, lr, weight_decay=0, momentum=0.9, eta=0.001, weight_decay_filter=None,
                 lars_adaptation_filter=None):
        defaults = dict(lr=lr, weight_decay=weight_decay, momentum=momentum, eta=eta,
                        weight_decay_filter=weight_decay_filter, lars_adaptation_filter=lars_adaptation_filter)
        super().__init__(params, defaults)
        self._fused = None
        self._fused_key = None

    @torch.no_grad()
    def step(self):
        plist, glist, lrs, adapt = [], [], [], []
        g0 = self.param_groups[0]
        for g in self.param_groups:
            if (g["weight_decay"], g["momentum"], g["eta"]) != (g0["weight_decay"], g0["momentum"], g0["eta"]):
                raise ValueError("fused LARS needs the same weight_decay / momentum / eta in every group")
            for p in g["params"]:
                if p.grad is None:
                    continue
                wd_on = g["weight_decay_filter"] is None or not g["weight_decay_filter"](p)
                ad_on = g["lars_adaptation_filter"] is None or not g["lars_adaptation_filter"](p)
                if wd_on != ad_on:
                    raise ValueError("fused LARS applies weight decay and the trust ratio to the same tensors")
                plist.append(p); glist.append(p.grad.contiguous()); lrs.append(g["lr"]); adapt.append(ad_on)
        if not plist:
            return
        key = tuple((p.data_ptr(), a) for p, a in zip(plist, adapt))
        if self._fused_key != key:
            self._fused = ops.LarsState([p.data for p in plist], adapt)
            for p, mu in zip(plist, self._fused.mu):
                st = self.state[p]
                if "mu" in st:
                    mu.copy_(st["mu"])
                st["mu"] = mu
            self._fused_key = key
        self._fused.step(glist, lrs, g0["weight_decay"], g0["momentum"], g0["eta"])
        # the kernel wrote the parameters through raw pointers: tell torch, so that everything keyed on a tensor's version sees the
        # update -- above all the bf16 weight cache of no-grad forwards (evaluation between epochs, the feature pass of
        # `running.micro_batch`), which would otherwise keep serving the weights of the first step it saw
        for p in plist:
            torch.autograd.graph.increment_version(p)
