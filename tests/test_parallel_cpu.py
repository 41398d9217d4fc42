"""Replica logic (vipant_amd/parallel.py) under torch.distributed "gloo", world_size 2, on CPU tensors.

The towers here are a tiny differentiable stand-in and the loss is the CPU oracle, because HIP kernels cannot run in
this container; what is under test is the exchange scheme itself: feature all-gather whose backward is a slice,
SUM all-reduce of per-layer gradient buckets, no reduction of logit_scale -- and the invariant it must deliver:
N replicas on a split batch == one process on the whole batch (the reference's dp-mode semantics)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as R


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tower(x, w):
    f = torch.tanh(x @ w)
    return f / f.norm(dim=-1, keepdim=True)


def _data(B=16, Din=24, E=64):
    g = torch.Generator().manual_seed(7)
    xa, xt = torch.randn(B, Din, generator=g), torch.randn(B, Din, generator=g)
    wa, wt = torch.randn(Din, E, generator=g) * 0.3, torch.randn(Din, E, generator=g) * 0.3
    return xa, xt, wa, wt


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vipant_amd import parallel
        assert parallel.world_size() == world and parallel.rank() == rank
        xa, xt, wa0, wt0 = _data()
        B = xa.shape[0]
        b = B // world
        sl = slice(rank * b, (rank + 1) * b)
        res = {}
        # ---- global negatives: gather features, loss on the full batch, slice backward, SUM reduce
        wa, wt = wa0.clone().requires_grad_(), wt0.clone().requires_grad_()
        ls = torch.tensor(2.0, requires_grad=True)
        fa, ft = _tower(xa[sl], wa), _tower(xt[sl], wt)
        ga, gt = parallel.all_gather_features(fa, ft)
        assert ga.shape == (B, 64) and torch.allclose(ga[sl], fa.detach())
        loss = R.ce_loss_head(ga, gt, ls)
        loss.backward()
        sync = parallel.GradSync()
        flat = torch.cat([wa.grad.reshape(-1), wt.grad.reshape(-1)])
        sync.reduce_async(flat)
        sync.wait()
        res["global"] = (loss.detach(), flat.clone(), ls.grad.clone())
        # ---- local negatives (cfg3: "DDP grad all-reduce only"): mean of per-rank losses
        wa, wt = wa0.clone().requires_grad_(), wt0.clone().requires_grad_()
        ls2 = torch.tensor(2.0, requires_grad=True)
        loss_l = R.ce_loss_head(_tower(xa[sl], wa), _tower(xt[sl], wt), ls2) / world
        loss_l.backward()
        sync.reduce_params([wa, wt, ls2]); sync.wait()
        res["local"] = (wa.grad.clone(), wt.grad.clone(), ls2.grad.clone())
        if rank == 0:
            torch.save(res, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_replicas_equal_one_process(tmp_path):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    xa, xt, wa0, wt0 = _data()
    # single process, whole batch
    wa, wt = wa0.clone().requires_grad_(), wt0.clone().requires_grad_()
    ls = torch.tensor(2.0, requires_grad=True)
    loss = R.ce_loss_head(_tower(xa, wa), _tower(xt, wt), ls)
    loss.backward()
    gl, gflat, gls = res["global"]
    assert torch.allclose(gl, loss.detach(), atol=1e-6)
    assert torch.allclose(gflat, torch.cat([wa.grad.reshape(-1), wt.grad.reshape(-1)]), atol=1e-6)
    assert torch.allclose(gls, ls.grad, atol=1e-6)          # complete on every rank without any reduction
    # local negatives == mean over ranks of the per-rank objective
    wa, wt = wa0.clone().requires_grad_(), wt0.clone().requires_grad_()
    ls = torch.tensor(2.0, requires_grad=True)
    tot = 0
    for r in range(2):
        sl = slice(r * 8, (r + 1) * 8)
        tot = tot + R.ce_loss_head(_tower(xa[sl], wa), _tower(xt[sl], wt), ls) / 2
    tot.backward()
    la, lt, lls = res["local"]
    assert torch.allclose(la, wa.grad, atol=1e-6) and torch.allclose(lt, wt.grad, atol=1e-6)
    assert torch.allclose(lls, ls.grad, atol=1e-6)


def test_single_process_is_a_no_op():
    from vipant_amd import parallel
    assert parallel.world_size() == 1 and parallel.rank() == 0
    s = parallel.GradSync()
    t = torch.ones(4)
    s.reduce_async(t); s.wait()
    assert torch.equal(t, torch.ones(4))


class _BucketedFn(torch.autograd.Function):
    """Mimics ops.BackboneFn: the gradients of several parameters are views of one flat buffer whose all-reduce is
    started INSIDE backward, before autograd has stored (and usually cloned) them into .grad."""

    @staticmethod
    def forward(ctx, x, sync, w1, w2):
        ctx.save_for_backward(x, w1, w2)
        ctx.sync, ctx.params = sync, (w1, w2)
        return (x @ w1) @ w2

    @staticmethod
    def backward(ctx, g):
        x, w1, w2 = ctx.saved_tensors
        flat = torch.empty(w1.numel() + w2.numel())
        v1, v2 = flat[:w1.numel()].view_as(w1), flat[w1.numel():].view_as(w2)
        v2.copy_((x @ w1).t() @ g)
        v1.copy_(x.t() @ (g @ w2.t()))
        ctx.sync.reduce_async(flat, [v1, v2], list(ctx.params))
        return None, None, v1, v2


def _bucket_worker(rank, world, port, out, overlap):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vipant_amd import parallel
        from vipant_amd.module import adjust_learning_rate
        g = torch.Generator().manual_seed(3)
        x = torch.randn(8, 5, generator=g)
        w1 = torch.nn.Parameter(torch.randn(5, 6, generator=g)); w2 = torch.nn.Parameter(torch.randn(6, 4, generator=g))
        sync = parallel.GradSync(overlap=overlap)      # "step": the buckets are held back and go out as ONE collective in wait()
        _BucketedFn.apply(x[rank * 4:rank * 4 + 4], sync, w1, w2).sum().backward()
        sync.wait()
        res = [w1.grad.clone(), w2.grad.clone()]
        # siamese towers sharing an encoder: the stack runs twice over the SAME parameters in one step, every run hands
        # over its own bucket; the reduced slices of a parameter must be summed, not overwrite each other
        w1.grad = w2.grad = None
        y = torch.randn(8, 5, generator=g)
        out2 = _BucketedFn.apply(x[rank * 4:rank * 4 + 4], sync, w1, w2).sum() \
            + 2.0 * _BucketedFn.apply(y[rank * 4:rank * 4 + 4], sync, w1, w2).sum()
        out2.backward()
        sync.wait()
        res += [w1.grad.clone(), w2.grad.clone()]
        # the LR schedule follows the whole batch: per-process batch x number of replicas (lars.py:9-22 with dp's batch)
        opt = type("O", (), {"param_groups": [{}, {}]})()
        ocfg = type("C", (), dict(epochs=2, warmup_epoch=0, batch_size=64, lr_weight=0.2, lr_bias=0.0048))()
        adjust_learning_rate(ocfg, opt, range(10), 0)
        res.append(torch.tensor(opt.param_groups[0]["lr"]))
        if rank == 0:
            torch.save(res, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("overlap", ["block", "step"])
def test_bucket_reduced_inside_backward_reaches_param_grad(tmp_path, overlap):
    """`running.comm_overlap`: per-block buckets reduced as they are handed over, or all of a step's buckets concatenated into one
    collective when the backward is over (two buckets over the same parameters in the siamese case): the same gradients."""
    out = str(tmp_path / "b.pt")
    mp.spawn(_bucket_worker, args=(2, _free_port(), out, overlap), nprocs=2, join=True)
    g1, g2, s1, s2, lr = torch.load(out)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(8, 5, generator=g)
    w1 = torch.randn(5, 6, generator=g).requires_grad_(); w2 = torch.randn(6, 4, generator=g).requires_grad_()
    ((x @ w1) @ w2).sum().backward()
    assert torch.allclose(g1, w1.grad, atol=1e-5) and torch.allclose(g2, w2.grad, atol=1e-5)
    y = torch.randn(8, 5, generator=g)
    w1.grad = w2.grad = None
    (((x @ w1) @ w2).sum() + 2.0 * ((y @ w1) @ w2).sum()).backward()
    assert torch.allclose(s1, w1.grad, atol=1e-4) and torch.allclose(s2, w2.grad, atol=1e-4)
    assert abs(float(lr) - (2 * 64 / 256) * 0.2) < 1e-7


def _bare_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vipant_amd import parallel
        res = {}
        for overlap in ("block", "step"):
            sync = parallel.GradSync(overlap=overlap)
            a, b = torch.full((5,), float(rank + 1)), torch.full((3,), 10.0 * (rank + 1))
            w = torch.nn.Parameter(torch.zeros(2))
            w.grad = torch.full((2,), 100.0 * (rank + 1))
            flat = w.grad.clone()
            sync.reduce_async(a)                            # bare buckets: reduced IN PLACE, whatever the overlap mode
            sync.reduce_async(flat, [flat.view_as(w)], [w])
            sync.reduce_async(b)
            sync.wait()
            res[overlap] = (a.clone(), b.clone(), w.grad.clone())
        if rank == 0:
            torch.save(res, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bare_buckets_are_reduced_in_place_in_both_overlap_modes(tmp_path):
    out = str(tmp_path / "bare.pt")
    mp.spawn(_bare_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    for overlap in ("block", "step"):
        a, b, g = res[overlap]
        assert torch.equal(a, torch.full((5,), 3.0)) and torch.equal(b, torch.full((3,), 30.0)) and torch.equal(g, torch.full((2,), 300.0))
