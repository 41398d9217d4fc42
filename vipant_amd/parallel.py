"""Data-parallel replicas: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on ROCm; "gloo" in
the CPU tests).  The path shards by samples; there are exactly two exchange steps:

  1. all-gather of the two [b, E] feature blocks before the loss, so that every rank scores the GLOBAL batch
     (the reference's dp-mode semantics, cvap/model/cvalp.py:41-61).  Every rank evaluates the full B x B loss and
     the fused kernel emits gradients only for the rank's own rows, so the backward of the gather is a slice --
     no reduce-scatter;
  2. gradient all-reduce (SUM): each rank holds dL/d(theta) through its own samples only.  The transformer stack
     hands each layer's gradients over as ONE flat fp32 buffer the moment that layer's backward is finished
     (ops.BackboneFn), so the reduction of layer l overlaps the backward of layers < l on a side stream.
     `logit_scale` is different: every rank computes its complete gradient, so it is never reduced in
     global-negatives mode.

     `running.comm_overlap` picks when the buckets go out: "block" (default) -- each block's bucket as soon as it is complete,
     overlapping the backward of the blocks below; "step" -- all of a step's block buckets as ONE all-reduce after the backward
     (nothing shares the chip with the backward, the reduction is exposed).  DESIGN.md section 6 has the numbers behind the default.

Device-agnostic on purpose: the same code runs under gloo on CPU tensors in tests/test_parallel_cpu.py.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def rank() -> int:
    return dist.get_rank() if is_dist() else 0


def active() -> bool:
    """Are the exchange steps on?  True in any group of more than one replica.  VIPANT_FORCE_COLLECTIVES=1 turns them on in
    a single-rank group as well, so the RCCL call path (stream hand-off, flat gather, bucket all-reduce) can be exercised on
    a one-GPU box, where it must be the identity (tests/test_replicas_gpu.py)."""
    return is_dist() and (dist.get_world_size() > 1 or os.environ.get("VIPANT_FORCE_COLLECTIVES") == "1")


class _AllGatherRows(torch.autograd.Function):
    """[b, C] per rank -> [world * b, C]; backward keeps this rank's rows (see module docstring)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        w = world_size()
        out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        if x.is_cuda and dist.get_backend() == "gloo":       # gloo cannot gather device tensors: stage through the host
            host = torch.empty(out.shape, dtype=x.dtype)      # (single-GPU replica tests only; RCCL takes the direct path)
            dist.all_gather(list(host.chunk(w, dim=0)), x.cpu())
            out.copy_(host)
        elif dist.get_backend() == "nccl":                    # RCCL: one flat collective, no per-rank output list
            dist.all_gather_into_tensor(out, x)
        else:
            dist.all_gather(list(out.chunk(w, dim=0)), x)
        ctx.b = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        r = rank()
        return g[r * ctx.b:(r + 1) * ctx.b].contiguous()


def all_gather_features(x1: torch.Tensor, x2: torch.Tensor):
    """One collective for both modalities: gathers [b, 2E], returns the two [B, E] halves (contiguous)."""
    E = x1.shape[1]
    both = _AllGatherRows.apply(torch.cat([x1, x2], dim=1))
    return both[:, :E].contiguous(), both[:, E:].contiguous()


class GradSync:
    """SUM all-reduce of gradient buckets on a side stream: per block, overlapping the backward (`overlap="block"`), or all of a
    step's block buckets as one collective when the backward is over (`overlap="step"`)."""

    def __init__(self, group=None, overlap: str = "block"):
        assert overlap in ("block", "step"), overlap
        self.group = group
        self.overlap = overlap
        self.handles: List = []
        self.buffers: List[torch.Tensor] = []
        self.pairs: List = []
        self.deferred: List = []            # overlap == "step": (flat, views, params) held back until wait()
        self.stream: Optional[torch.cuda.Stream] = None

    def _comm_stream(self, device):
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=device)
        return self.stream

    def reduce_async(self, flat: torch.Tensor, views=None, params=None):
        """Start the SUM all-reduce of one bucket.  `views` / `params` (optional, same length) name the slices of
        `flat` that are the gradients of `params`: autograd usually CLONES a gradient it is handed while other
        references to it exist, so after the reduction `wait()` copies the reduced slices over whatever tensor ended
        up in `param.grad`."""
        if self.overlap == "step" and active():
            self.deferred.append((flat, views, params))
            return
        self._start(flat, views, params)

    def _start(self, flat: torch.Tensor, views=None, params=None):
        if not active():
            return
        if views is not None:
            self.pairs.extend(zip(params, views))
        if flat.is_cuda:
            comm = self._comm_stream(flat.device)
            comm.wait_stream(torch.cuda.current_stream(flat.device))    # bucket is complete on the compute stream
            with torch.cuda.stream(comm):
                self.handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            flat.record_stream(comm)
        else:
            self.handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.buffers.append(flat)

    def reduce_params(self, params: Iterable[torch.nn.Parameter]):
        """Bucket the (small) gradients that did not come through a layer bucket: patch embedding, read-out.  Started
        asynchronously like the layer buckets; `wait()` hands the reduced slices back as the parameters' `.grad`."""
        seen, plist = set(), []
        for p in params:
            if p.grad is not None and id(p) not in seen:
                seen.add(id(p)); plist.append(p)
        if not active() or not plist:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in plist])
        views, off = [], 0
        for p in plist:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.reduce_async(flat, views, plist)

    def wait(self):
        """Block until every bucket is reduced, then make the reduced slices the parameters' gradients.

        A parameter can own several slices -- a siamese shared encoder runs the stack twice over the same weights, once per
        tower, and each run hands over its own bucket -- so the slices of one parameter are summed; the first one is
        installed as `.grad` by reference (no 352 MB copy-back per step), the rest are added to it."""
        back = []
        if self.deferred:                    # overlap == "step": one flat buffer, one collective; the views move into it
            held, self.deferred = self.deferred, []
            if len(held) == 1:
                self._start(*held[0])
            else:
                flat = torch.cat([f for f, _, _ in held])
                views, params, off = [], [], 0
                for f, vs, ps in held:
                    if not vs:          # a bare bucket: reduce_async's in-place contract -- the reduced slice goes back into it
                        back.append((f, flat[off:off + f.numel()]))
                    for v, p in zip(vs or [], ps or []):
                        o = off + v.storage_offset() - f.storage_offset()
                        views.append(flat[o:o + v.numel()].view(v.shape))
                        params.append(p)
                    off += f.numel()
                self._start(flat, views, params)
        for h in self.handles:
            h.wait()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        for f, red in back:
            f.copy_(red.view_as(f))
        first = set()
        for p, v in self.pairs:
            if p.grad is None:
                continue
            if id(p) not in first:
                first.add(id(p))
                p.grad = v
            else:
                p.grad.add_(v)
        self.handles, self.buffers, self.pairs = [], [], []
