"""Loss heads with the reference's registry / forward contract (cvap/module/decoder/loss_head.py).

Training path: `CELossHead.forward` = symmetric InfoNCE through the fused HIP kernels (ops.InfoNCEFn); in a
replica group the features are first all-gathered so every rank scores the GLOBAL batch -- the semantics of
the reference's dp mode, where the loss head sees the gathered batch (cvap/model/cvalp.py:41-61).
Eval path: `infer` caches normalised features on the device, `report` ranks them with the fused retrieval kernels.
"""
from __future__ import annotations

import json
from collections import defaultdict

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
    """cvap/module/decoder/loss_head.py:25-244: feature cache (`infer`) and the retrieval report.

    The reference sorts every row of `x1s @ x2s.t()` and looks up the gold column; here the ranks come from
    `ops.retrieval_ranks` (fused similarity tiles + counting on the GPU, nothing N x N is stored).  What is left on
    the host is the formatting of a few counters, done in the reference's own dtypes so the strings agree."""

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
            x1, x2 = ops.l2_normalize(x1.detach().float()), ops.l2_normalize(x2.detach().float())
        self.x1s.append(x1.detach().float())
        self.x2s.append(x2.detach().float())
        names = kwargs.get("names", None)
        if names is not None:
            self.ids.extend(names)
        return None

    @staticmethod
    def _gold_cluster(gold_file, nsample, verbose=False):
        """loss_head.py:48-66: class name ("labels" joined by blanks) <-> sample ids, first `nsample` records."""
        sample_by_classname, classname_by_sample = defaultdict(list), defaultdict(str)
        with open(gold_file, "r") as fr:
            for iline, line in enumerate(fr):
                if iline + 1 > nsample:
                    break
                record = json.loads(line)
                key = " ".join(record["labels"])
                sample_by_classname[key].append(record["id"])
                classname_by_sample[record["id"]] = key
        return sample_by_classname, classname_by_sample

    @staticmethod
    def retrieval_metrics(ranks, nsample=None, msg=""):
        """loss_head.py:68-78; `ranks` is a float32 vector as in the reference (median = lower median)."""
        ranks = torch.as_tensor(ranks).float().cpu()
        nsample = nsample or ranks.shape[0]
        hit = lambda k: int((ranks < k).sum()) / nsample * 100.
        med, avg = ranks.median() + 1, ranks.mean() + 1
        return (f"{msg}: R@1 {hit(1):2.2f} R5 {hit(5):2.2f} R10 {hit(10):2.2f} R50 {hit(50):2.2f} "
                f"MED {med:2.2f} AVG {avg:2.2f}")

    @staticmethod
    def _one_v_k_ranks(x1s, x2s, k=5):
        """ranks of the k captions of every clip (x1 -> x2, [n1, k]) and of every caption's clip (x2 -> x1, [n1 * k])."""
        n1 = x1s.shape[0]
        gold12 = torch.arange(n1 * k, dtype=torch.int32).reshape(n1, k)
        gold21 = torch.arange(n1, dtype=torch.int32).repeat_interleave(k)
        return ops.retrieval_ranks(x1s, x2s, gold12).cpu(), ops.retrieval_ranks(x2s, x1s, gold21).cpu()

    @staticmethod
    def retrieval_eval(x1s, x2s, k=5, _ranks=None):
        """loss_head.py:80-107: best-caption rank per clip and clip rank per caption."""
        r12, r21 = _ranks if _ranks is not None else LossHead._one_v_k_ranks(x1s, x2s, k)
        msg_12 = LossHead.retrieval_metrics(r12.min(-1)[0], msg="A->T")
        msg_21 = LossHead.retrieval_metrics(r21, msg="T->A")
        return f"{msg_12}\n{msg_21}"

    def _class_stats(self, top1, sample_by_classname, classname_by_sample, nsample, msg, k=1):
        """loss_head.py:176-232 (topk_overlap + pnr at k = 1): precision / recall of the nearest neighbour by class."""
        stats = defaultdict(dict)
        for idx, neighbor in enumerate(top1):
            sample = self.ids[idx]
            classname = classname_by_sample[sample]
            true_neighbors = sample_by_classname[classname]
            sample_stat = stats.get(classname, {})
            this_stat = sample_stat.get(sample, [0] * 2)
            if self.ids[neighbor] in true_neighbors:
                this_stat[0] += 1
            sample_stat[sample] = this_stat
            stats[classname] = sample_stat
        p = r = p_cls = r_cls = 0.
        nclass = len(sample_by_classname)
        for classname, class_stats in stats.items():
            nrelevant = len(sample_by_classname[classname])
            pc = rc = 0
            for sample, (tp, _) in class_stats.items():
                p += tp / k; r += tp / nrelevant
                pc += tp / k; rc += tp / nrelevant
            p_cls += pc / nrelevant
            r_cls += rc / nrelevant
        p, r = (p / nsample) * 100, (r / nsample) * 100
        p_cls, r_cls = (p_cls / nclass) * 100, (r_cls / nclass) * 100
        return f"{msg}: P@{k} {p:2.2f} R@{k} {r:2.2f} mAP {p_cls:2.2f} mAR {r_cls:2.2f}"

    def report(self, gold_file=None):
        x1s, x2s = torch.cat(self.x1s), torch.cat(self.x2s)
        n1, n2 = x1s.shape[0], x2s.shape[0]
        ref_metric = ""
        top12 = top21 = None
        if n1 == n2:
            gold = torch.arange(n1, dtype=torch.int32)
            r12, top12 = ops.retrieval_ranks(x1s, x2s, gold, want_top1=True)
            r21, top21 = ops.retrieval_ranks(x2s, x1s, gold, want_top1=True)
            pct = lambda r, k: int((r < k).sum()) / n1 * 100.
            p_12 = f"I->A: t1 = {pct(r12, 1):2.2f} t5 = {pct(r12, 5):2.2f}"
            p_21 = f"A->I: t1 = {pct(r21, 1):2.2f} t5 = {pct(r21, 5):2.2f}"
        elif n1 * 5 == n2:   # 1 clip vs 5 captions (loss_head.py:135-168)
            r12, r21 = self._one_v_k_ranks(x1s, x2s, 5)
            t1 = (r12 < 1).sum(-1).sum() / (1 * n1) * 100.     # float32 tensors, as in the reference
            t5 = (r12 < 5).sum(-1).sum() / (5 * n1) * 100.
            mean = r12.min(-1)[0].float().mean() + 1
            p_12 = f"A->T: t1 = {t1:2.2f} t5 = {t5:2.2f} mR = {mean:2.2f}"
            t1 = int((r21 < 1).sum()) / n2 * 100.
            t5 = int((r21 < 5).sum()) / n2 * 100.
            mean = r21.float().mean() + 1
            p_21 = f"T->A: t1 = {t1:2.2f} t5 = {t5:2.2f} mR = {mean:2.2f}"
            ref_metric = self.retrieval_eval(x1s, x2s, _ranks=(r12, r21))
            gold_file = None
        else:
            p_12, p_21 = f"{x1s.shape}x{x2s.shape}", "-"
            gold_file = None
        msg_12 = msg_21 = ""
        if gold_file is not None:
            by_class, by_sample = self._gold_cluster(gold_file, n1)
            msg_12 = self._class_stats(top12.tolist(), by_class, by_sample, n1, "I->A")
            msg_21 = self._class_stats(top21.tolist(), by_class, by_sample, n1, "A->I")
        del self.x1s, self.x2s, self.ids
        msg = "" if msg_12 == msg_21 == "" else f"\n{msg_12} {msg_21}\n"
        ref = "" if ref_metric == "" else f"\nREFERENCE\n{ref_metric}"
        return f"{msg}{p_12} {p_21} @ {n1}{ref}"


def zero_shot_report(audios, labels, text, label_map=None) -> str:
    """Zero-shot classification report: the `text is not None` branch of ClassificationHead.report
    (cvap/module/decoder/loss_head.py:371-407), as used by cvap/monitor/esc50_clf.py:260-325.

    audios [n, E] audio features, text [c, E] one feature per class prompt, labels int [n]; the prediction is the
    arg-max prompt (mapped through `label_map` when prompts and labels use different ids).  The reference sorts all of
    `audios @ text.t()`; here the arg-max comes from the fused retrieval kernel (`top1`), nothing n x c is stored."""
    audios, text = audios.detach().float().contiguous(), text.detach().float().contiguous()
    n = audios.shape[0]
    gold = torch.zeros((n,), dtype=torch.int32)
    _, top1 = ops.retrieval_ranks(audios, text, gold, want_top1=True)
    predictions = top1.long()
    if isinstance(label_map, dict):
        predictions = torch.tensor([label_map[x] for x in predictions.tolist()], device=predictions.device)
    labels = torch.as_tensor(labels, device=predictions.device).long()
    precision = (predictions == labels).sum() / n * 100.      # float32 tensor, as in the reference
    return f"A->T: p1 = {precision:2.2f} @ {n}"


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
        # "local" = per-rank negatives, objective = mean over ranks of the per-rank loss (what the reference's ddp mode
        # would compute): every gradient leaves the kernel scaled by 1/world and the replicas SUM-reduce
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
        if parallel.active() and self.negatives == "global":
            x1g, x2g = parallel.all_gather_features(x1, x2)
            return ops.InfoNCEFn.apply(x1g, x2g, ls, scale_max, parallel.rank() * b, b, 1.0)
        return ops.InfoNCEFn.apply(x1, x2, ls, scale_max, 0, b, 1.0 / parallel.world_size() if parallel.active() else 1.0)


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
