"""Trainer loop with the reference's Monitor contract (`Monitor(cfg, echo, device).learn()`), reproducing
the step semantics of VALMonitor.epoch / VAMonitor.epoch (cvap/monitor/cvalp.py:172-272, cvap/monitor/cvap.py:160-244):
per step -- H2D, LARS learning-rate schedule, zero_grad(set_to_none), forward, backward, optimizer step, counters,
`samples/s` log line, periodic eval + checkpoint.

Deliberate deviations (documented in DESIGN.md):
  * bf16 MFMA operands with fp32 accumulation and an fp32 residual stream need no loss scaling, so the reference's
    fp16 autocast + GradScaler pair (cvalp.py:123, 201-205) has no counterpart here;
  * one process per GPU: `mode=ddp` is served by vipant_amd.parallel (feature all-gather + bucketed gradient
    all-reduce); the reference's ddp branch cannot start (train.py:35 names an undefined Monitor);
  * the reference's dataset builders (cvap/data/*) are outside the hot-path scope; batches come from a synthetic
    loader with the same tensor contract unless a loader is injected through `dataloader=`.
"""
from __future__ import annotations

import os
import time
from collections import defaultdict

import numpy as np
import torch

from . import parallel
from .model import build_main_model
from .module import LARS, adjust_learning_rate, exclude_bias_or_norm

SOT, EOT = 49406, 49407


class SyntheticLoader:
    """Batches with the tensor contract of the reference collators (cvap/data/audioset_clf.py:122-152,
    cvap/monitor/cvalp.py:136-154): (images, audios [b, T, F], text [b, L] i64, labels, names).  Seeded per rank and
    per step, reproducible.  `running.dp_chunk` (set by train.py for `mode=dp` under N replicas): the loader
    stands for the reference's ONE loader whose batch data_parallel scatters -- every replica draws the same global batch of
    N x batch_size samples and keeps its contiguous chunk."""

    def __init__(self, cfg, steps, with_text, device_rank=0):
        self.cfg, self.steps, self.with_text, self.rank = cfg, int(steps), with_text, device_rank
        rcfg = cfg.running
        self.b = int(rcfg.batch_size)
        self.chunk = None
        if rcfg.get("dp_chunk", False) and parallel.world_size() > 1:
            self.chunk = (parallel.rank(), self.b)
            self.b, self.rank = self.b * parallel.world_size(), 0
        self.T, self.F = int(rcfg.max_audio_len), int(rcfg.num_mel_bins)
        self.precomputed = bool(rcfg.get("precomputed_image", False)) or rcfg.get("frame_emb", None) is not None
        self.res = int(cfg.running.resolution) if not isinstance(cfg.running.resolution, (list, tuple)) else 224
        self.embed = int(cfg.running.embed_dim)

    def __len__(self):
        return self.steps

    def __iter__(self):
        for step in range(self.steps):
            g = torch.Generator().manual_seed(1213 + 1000 * self.rank + step)
            audios = torch.randn(self.b, self.T, self.F, generator=g)
            if self.with_text:
                images = torch.zeros(self.b, 1, 1, 1)
                lens = torch.randint(8, 78, (self.b,), generator=g)
                L = int(lens.max())
                text = torch.zeros(self.b, L, dtype=torch.int64)
                for i in range(self.b):
                    n = int(lens[i])
                    text[i, 0] = SOT
                    text[i, 1:n - 1] = torch.randint(1, SOT, (n - 2,), generator=g)
                    text[i, n - 1] = EOT
            else:
                images = (torch.randn(self.b, self.embed, generator=g) if self.precomputed
                          else torch.randn(self.b, 3, self.res, self.res, generator=g))
                text = torch.zeros(self.b, 1, dtype=torch.int64)
            labels = torch.zeros(self.b, dtype=torch.int64)
            names = [f"synthetic-{self.rank}-{step}-{i}" for i in range(self.b)]
            if self.chunk is not None:
                sl = slice(self.chunk[0] * self.chunk[1], (self.chunk[0] + 1) * self.chunk[1])
                images, audios, text, labels, names = images[sl], audios[sl], text[sl], labels[sl], names[sl]
            yield images, audios, text, labels, names


class Monitor(object):
    with_text = True      # VALMonitor (AT); VAMonitor sets False

    def __init__(self, cfg, echo, device, dataloader=None):
        self.cfg, self.echo, self.device = cfg, echo, device
        self.dataloader = dataloader if dataloader is not None else self.build_data()
        self.evalloader = self.testloader = None
        self.gold_file = None
        model = build_main_model(cfg, echo)
        negatives = cfg.running.get("negatives", "global")
        tunable_params = model.build(negatives=negatives)
        self.model = model
        self.grad_sync = None
        if parallel.active():
            # `running.comm_overlap`: block (default: each block's bucket is reduced while the blocks below run their backward) | step
            self.grad_sync = parallel.GradSync(overlap=str(cfg.running.get("comm_overlap", "block")))
            for head in (model.audio_head, model.image_head, model.text_head):
                if head is not None and hasattr(head, "encoder"):
                    head.encoder.grad_sync = self.grad_sync
        if cfg.running.get("recompute_mlp", False):
            for head in (model.audio_head, model.image_head, model.text_head):
                if head is not None and hasattr(head, "encoder"):
                    head.encoder.recompute_mlp = True
        if cfg.running.get("grad_stream", None) is not None:        # bf16 (default) | fp32 master of the stream gradient
            from . import ops
            ops.GRAD_STREAM_F32 = str(cfg.running.grad_stream).lower() in ("fp32", "float32")
        # the residual stream inside every transformer stack: fp16 (the reference's autocast precision; default) or fp32;
        # LayerNorm statistics are fp32 either way
        stream_f16 = str(cfg.running.get("stream_dtype", "fp16")).lower() in ("fp16", "float16", "half")
        # `running.last_block_rows` (default on): a tower's last block is evaluated on the rows its read-out takes (class / end-of-text
        # token) -- exact, see ops.BackboneFn; off: the full block, as the reference computes it before discarding the other rows
        last_rows = bool(cfg.running.get("last_block_rows", True)) and os.environ.get("VIPANT_LAST_BLOCK_ROWS", "1") != "0"
        for head in (model.audio_head, model.image_head, model.text_head):
            if head is not None and hasattr(head, "encoder"):
                head.encoder.stream_f16 = stream_f16
                head.encoder.last_block_rows = last_rows
        if cfg.running.get("fp8_gemm", False):      # BASELINE.json configs[4]: e4m3 operands in the audio tower's NT contractions
            if model.audio_head is not None and hasattr(model.audio_head, "encoder"):
                model.audio_head.encoder.fp8 = True
        self.model.train(not cfg.eval)
        self.build_optimizer(tunable_params)

    def build_data(self):
        rcfg = self.cfg.running
        steps = rcfg.get("synthetic_steps", 8)
        loader = SyntheticLoader(self.cfg, steps, self.with_text, max(self.cfg.rank, 0))
        self.echo(f"Instantiate main dataloader from `synthetic': total {len(loader)} batches.")
        return loader

    # ------------------------------------------------------------------ cvalp.py:106-130
    def learn(self):
        if not self.model.training:
            self.echo("Evaluating started...")
            with torch.no_grad():
                report = self.infer(self.dataloader, samples=self.cfg.running.eval_samples, gold_file=self.gold_file)
                self.echo(f"{report}")
                return None
        self.echo("Training started...")
        self.last_time = 0.
        self.total_loss = 0
        self.total_step = 0
        self.total_inst = 0
        self.start_time = time.time()
        for iepoch in range(int(self.cfg.optimizer.epochs)):
            self.epoch(iepoch)

    # ------------------------------------------------------------------ cvalp.py:136-154
    def make_batch(self, batch):
        images = torch.as_tensor(batch[0]).to(self.device, non_blocking=True)
        audios = torch.as_tensor(batch[1]).to(self.device, non_blocking=True).unsqueeze(1)
        text = torch.as_tensor(batch[2]).to(self.device, non_blocking=True)
        labels = torch.as_tensor(batch[3]).to(self.device, non_blocking=True)
        return images, audios, text, labels, batch[-1]

    def timeit(self, time_dict, key=None, show=False):
        if self.cfg.rank > 0:
            return
        if show:
            report = " ".join(f"{k} {np.mean(v):.2f}" for k, v in time_dict.items())
            self.echo(f"Time (s): {report}; # step {self.total_step} # sample {self.total_inst}")
            return
        if key is None:
            self.last_time = time.time()
        else:
            this_time = time.time()
            time_dict[key].append(this_time - self.last_time)
            self.last_time = this_time

    # ------------------------------------------------------------------ one optimisation step
    def _forward_backward_micro(self, images, audios, text, mb):
        """One global-batch step with the towers run `mb` samples at a time (`running.micro_batch`), for batches whose
        activations do not fit at once (BASELINE.json configs[4]: ViT-L, 1024 clips per GPU).  The objective is untouched --
        every sample is still scored against the whole batch:
          1. features of all micro-batches, nothing kept for a backward;
          2. the loss over the concatenated features -> d loss / d features (and the loss head's own gradients);
          3. every micro-batch again, this time kept, and back-propagated from its slice of (2).
        The parameter gradients equal the one-pass gradients up to fp32 summation order; the price is one extra tower forward."""
        n = audios.shape[0]
        cuts = [(i, min(i + mb, n)) for i in range(0, n, mb)]
        parts = []
        with torch.no_grad():
            for a, b in cuts:
                parts.append(self.model.features(images[a:b], audios[a:b], text[a:b] if text is not None else None))
        feats = [None if parts[0][m] is None else torch.cat([p[m] for p in parts]).detach() for m in range(3)]
        heads = (self.model.image_head, self.model.audio_head, self.model.text_head)
        live = [m for m in range(3) if feats[m] is not None and heads[m] is not None
                and any(p.requires_grad for p in heads[m].parameters())]
        for m in live:
            feats[m].requires_grad_()
        loss = self.model.loss_from_features(*feats)
        loss.backward()
        if not live:                       # every tower frozen: only the loss head trains, nothing to re-run
            return loss
        # Pass 3 under data-parallel replicas: the per-block reduction buckets of the stacks are switched off for the micro-batches
        # -- each one would start a full set of all-reduces (n_micro x the traffic and n_micro flat gradient copies) -- autograd
        # accumulates locally instead and `step()` reduces the accumulated gradients ONCE (all parameters through reduce_params).
        # Towers that are frozen keep their pass-1 features: only the live modalities are run again.
        stacks = [h.encoder for h in heads if h is not None and hasattr(h, "encoder")]
        saved_sync = [st.grad_sync for st in stacks]
        for st in stacks:
            st.grad_sync = None
        try:
            for a, b in cuts:
                again = self.model.features(images[a:b] if 0 in live else None, audios[a:b] if 1 in live else None,
                                            text[a:b] if (2 in live and text is not None) else None)
                torch.autograd.backward([again[m] for m in live], [feats[m].grad[a:b] for m in live])
        finally:
            for st, gs in zip(stacks, saved_sync):
                st.grad_sync = gs
        self._micro_step = True
        return loss

    def step(self, images, audios, text):
        """zero_grad -> forward -> backward -> (replica gradient reduction) -> optimizer (cvalp.py:200-205)."""
        self.optimizer.zero_grad(set_to_none=True)
        mb = int(self.cfg.running.get("micro_batch", 0) or 0)
        if 0 < mb < audios.shape[0]:
            loss = self._forward_backward_micro(images, audios, text if self.with_text else None, mb)
        else:
            loss = self.model(images, audios, text if self.with_text else None)
            from . import ops
            loss.backward(gradient=ops.unit_grad(loss.device) if loss.is_cuda and loss.dim() == 0 and loss.dtype == torch.float32 else None)
        if self.grad_sync is not None:
            micro = getattr(self, "_micro_step", False)         # micro-batched step: nothing went through a per-block bucket
            self._micro_step = False
            rest = [p for p in self.params if micro or not getattr(p, "_vipant_bucketed", False)]
            if self.model.loss_head is not None and self.cfg.running.get("negatives", "global") == "global":
                skip = {id(p) for p in self.model.loss_head.parameters()}      # complete on every rank already
                rest = [p for p in rest if id(p) not in skip]
            self.grad_sync.reduce_params(rest)      # one more bucket behind the per-block ones already in flight
            self.grad_sync.wait()
        self.optimizer.step()
        return loss

    # ------------------------------------------------------------------ cvalp.py:172-272
    def epoch(self, iepoch):
        all_time = defaultdict(list)
        self.timeit(all_time)
        nchunk = parallel.world_size()
        ocfg = self.cfg.optimizer
        warmup_step_rate = max(int(ocfg.warmup_steps) // 20, 1)      # cvalp.py:177
        for step, batch in enumerate(self.dataloader, start=iepoch * len(self.dataloader)):
            images, audios, text, _, _ = self.make_batch(batch)
            self.timeit(all_time, key="data")
            if ocfg.use_lars:
                adjust_learning_rate(ocfg, self.optimizer, self.dataloader, step)
            # linear warm-up of a torch.optim optimizer (cvalp.py:185-198); always applied at the very first step
            warmup = (not ocfg.use_lars) and ocfg.warmup and (self.total_step + 1) <= ocfg.warmup_steps
            if warmup and ((self.total_step + 1) % warmup_step_rate == 0 or self.total_step == 0):
                ratio = (self.total_step + 1) / ocfg.warmup_steps
                for param_group in self.optimizer.param_groups:
                    param_group["lr"] = ratio * param_group["initial_lr"]
                lrs = " ".join(f"{g['lr']:.2e}" for g in self.optimizer.param_groups)
                self.echo(f"warmup lr: {lrs} @ {self.total_step}")
            loss = self.step(images, audios, text)
            if not ocfg.use_lars and ocfg.batch_sch and not warmup:
                self.scheduler.step()               # after all warm-up is completed (cvalp.py:207-209)
            self.timeit(all_time, key="model")
            self.total_step += 1
            self.total_loss += loss.detach()
            self.total_inst += images.shape[0] * nchunk
            if self.cfg.rank <= 0 and self.total_step % self.cfg.running.peep_rate == 0:
                lr_w = self.optimizer.param_groups[0]["lr"]
                lr_b = self.optimizer.param_groups[1]["lr"]
                msg = self.model.report(**{"nstep": self.total_step})
                self.echo(
                    f"epoch {iepoch:>4} step {self.total_step}\t"
                    f"lr_w {lr_w:.2e} lr_b {lr_b:.2e} loss {float(self.total_loss) / self.total_step:.3f} "
                    f"{msg} {self.total_inst / (time.time() - self.start_time):.2f} samples/s"
                )
            if self.total_step % self.cfg.running.save_rate == 0 or (
                    self.cfg.running.save_epoch and self.total_step % len(self.dataloader) == 0):
                for loader, samples in ((self.evalloader, self.cfg.running.eval_samples),
                                        (self.testloader, self.cfg.running.test_samples)):
                    if loader is None:
                        continue
                    self.model.train(False)
                    with torch.no_grad():
                        report = self.infer(loader, samples=samples, iepoch=iepoch, gold_file=self.gold_file)
                    self.model.train(True)
                    if report != "":
                        self.echo(f"{report}")
                if self.cfg.rank <= 0:
                    self.save()
            self.timeit(all_time, key="report")
        if not ocfg.use_lars and not ocfg.batch_sch:
            self.scheduler.step()                   # per-epoch schedule (cvalp.py:270-271)
        self.timeit(all_time, show=True)

    # ------------------------------------------------------------------ cvalp.py:274-300
    def infer(self, dataloader, samples=float("inf"), iepoch=0, gold_file=None):
        nsample, nchunk = 0, 1
        start_time = time.time()
        for ibatch, batch in enumerate(dataloader):
            if nsample >= samples:
                break
            images, audios, text, _, names = self.make_batch(batch)
            self.model(images, audios, text if self.with_text else None, names=names)
            nsample += images.shape[0] * nchunk
        self.echo(f"# sample {nsample}; {nsample / (time.time() - start_time):.2f} samples/s")
        return self.model.report(gold_file=gold_file)

    # ------------------------------------------------------------------ cvalp.py:302-309
    def save(self):
        from .config import to_plain
        out_dir = f"{self.cfg.alias_root}/{self.cfg.model_name}"
        os.makedirs(out_dir, exist_ok=True)
        fsave = f"{out_dir}/{self.total_step:08d}.pth"
        self.echo(f"Saving the checkpoint to {fsave}")
        torch.save({"cfg": to_plain(self.cfg), "model": self.model.collect_audio_state_dict()}, fsave)

    # ------------------------------------------------------------------ cvalp.py:311-348
    def build_optimizer(self, tunable_params={}):
        if not self.model.training:
            return
        self.params = list(tunable_params.values())
        for k, v in self.model.named_parameters():
            if k not in tunable_params:
                v.requires_grad = False
        ntotal = sum(p.numel() for p in self.model.parameters())
        ntune = sum(p.numel() for p in self.model.parameters() if p.requires_grad)
        self.echo(f"# param {ntotal / 1e6:.2f}M # tunable {ntune / 1e6:.2f}M.")
        # transformer-stack parameters are reduced per layer by the backward itself (parallel.GradSync)
        for head in (self.model.audio_head, self.model.image_head, self.model.text_head):
            if head is not None and hasattr(head, "encoder"):
                for p in head.encoder.parameters():
                    p._vipant_bucketed = True
        param_groups = [
            {"params": [p for p in self.params if p.ndim > 1]},
            {"params": [p for p in self.params if p.ndim < 2]},
        ]
        ocfg = self.cfg.optimizer
        if ocfg.use_lars:
            self.optimizer = LARS(param_groups, lr=0., weight_decay=ocfg.weight_decay,
                                  weight_decay_filter=exclude_bias_or_norm, lars_adaptation_filter=exclude_bias_or_norm)
            # `running.batch_size` is PER PROCESS here (one process per GPU), while the reference's dp mode feeds every GPU from
            # one process: say which global batch and base learning rate this launch trains with (module/lars.py)
            world = parallel.world_size()
            self.echo(f"LARS schedule: per-process batch {ocfg.batch_size} x {world} replica(s) = global batch "
                      f"{ocfg.batch_size * world}, base lr {ocfg.batch_size * world / 256:g}")
        else:
            # cvalp.py:338-342: any torch.optim optimizer + lr scheduler named in the config (`optimizer.optimizer`,
            # `optimizer.scheduler`).  The update itself runs through torch's own optimizer kernels -- only LARS, the default
            # of both launch scripts, has a fused HIP step; forward / backward are the HIP path either way.
            ocfg_opt, ocfg_sch = ocfg.optimizer, ocfg.scheduler
            self.optimizer = getattr(torch.optim, ocfg_opt[0])(param_groups, **dict(ocfg_opt[1]))
            self.scheduler = getattr(torch.optim.lr_scheduler, ocfg_sch[0])(self.optimizer, **dict(ocfg_sch[1]))
        if self.cfg.verbose:
            self.echo("Gradienting The Following Parameters:")
            for k, v in self.model.named_parameters():
                if v.requires_grad:
                    self.echo(f"{k} {tuple(v.size())}")


class VALMonitor(Monitor):
    """AT fine-tuning trainer (cvap/monitor/cvalp.py:23)."""
    with_text = True


class VAMonitor(Monitor):
    """VA pre-training trainer (cvap/monitor/cvap.py:21)."""
    with_text = False


__all__ = ["Monitor", "VAMonitor", "VALMonitor", "SyntheticLoader"]
