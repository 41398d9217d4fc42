"""Two replicas of the REAL HIP training step against one process on the whole batch (MI355X, single GPU).

RCCL refuses two ranks on one device, so the two processes share cuda:0 and talk over gloo; everything else is the
product path: local heads, feature all-gather -> global-batch InfoNCE with gradients for the local rows only,
per-block gradient buckets reduced inside the backward, LARS step.  Invariant: the parameters after one
step equal those of a single process fed the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

OV = ("+running=bimodal worker=CVALP mode=ddp eval=False num_gpus=1 +model/image=vit_val +model/audio=vit_val "
      "+model/text=dummy +model/loss=ce +optimizer=standard +running/audio=default "
      "model.audio.pre_encoder.stride=[16,24] model.image.encoder.layers=2 running.audio.max_len=256 "
      "running.audio.num_mel_bins=64 running.epochs=2 running.frame_emb=synthetic "
      "running.synthetic_steps=2 running.save_epoch=False optimizer.warmup_epoch=1").split()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(B):
    g = torch.Generator().manual_seed(99)
    return torch.randn(B, 512, generator=g), torch.randn(B, 1, 256, 64, generator=g)


def _run(rank, world, port, out, backend="gloo"):
    if world > 1 or backend == "nccl":
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if backend == "nccl":                       # single-rank RCCL group with the exchange steps forced on
            os.environ["VIPANT_FORCE_COLLECTIVES"] = "1"
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vipant_amd.config import compose
        from vipant_amd.module import adjust_learning_rate
        from vipant_amd.monitor import VAMonitor
        B = 16
        b = B // world
        cfg = compose(OV + [f"running.batch_size={b}"])
        cfg.rank = 0
        torch.manual_seed(cfg.seed)
        mon = VAMonitor(cfg, lambda *_: None, torch.device("cuda:0"))
        img, aud = _batch(B)
        sl = slice(rank * b, (rank + 1) * b)
        adjust_learning_rate(cfg.optimizer, mon.optimizer, mon.dataloader, 1)
        # same LR as the single-process run: the schedule scales with the PER-PROCESS batch size in the reference
        for gparam in mon.optimizer.param_groups:
            gparam["lr"] = gparam["lr"] * world
        loss = mon.step(img[sl].cuda(), aud[sl].cuda(), None)
        torch.cuda.synchronize()
        if rank == 0:
            sd = {k: v.detach().cpu() for k, v in mon.model.named_parameters() if v.requires_grad}
            torch.save({"loss": float(loss.detach()), "params": sd}, out)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_replicas_match_single_process(tmp_path):
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    mp.spawn(_run, args=(1, 0, one), nprocs=1, join=True)
    mp.spawn(_run, args=(2, _free_port(), two), nprocs=2, join=True)
    a, b = torch.load(one), torch.load(two)
    assert abs(a["loss"] - b["loss"]) < 1e-5, (a["loss"], b["loss"])
    assert a["params"].keys() == b["params"].keys()
    for k in a["params"]:
        pa, pb = a["params"][k], b["params"][k]
        err = float((pa - pb).abs().max())
        assert err <= 1e-6 + 2e-4 * float(pa.abs().max()), (k, err)


@pytest.mark.timeout(600)
def test_rccl_call_path_single_rank_is_identity(tmp_path):
    """The RCCL code path itself (flat feature all-gather, per-block bucket all-reduce on the side stream, copy-back into
    .grad, small-parameter bucket) on the real library: a one-rank `nccl` group with the collectives forced on must give
    exactly the step of a process without a group."""
    one, rc = str(tmp_path / "one.pt"), str(tmp_path / "rccl.pt")
    mp.spawn(_run, args=(1, 0, one), nprocs=1, join=True)
    mp.spawn(_run, args=(1, _free_port(), rc, "nccl"), nprocs=1, join=True)
    a, b = torch.load(one), torch.load(rc)
    assert a["loss"] == b["loss"], (a["loss"], b["loss"])
    for k in a["params"]:
        assert torch.equal(a["params"][k], b["params"][k]), k

