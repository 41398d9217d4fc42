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


def _run(rank, world, port, out, backend="gloo", extra=()):
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
        B = int(os.environ.get("VIPANT_TEST_GLOBAL_BATCH", "16"))
        b = B // world
        cfg = compose(OV + [f"running.batch_size={b}"] + list(extra))
        cfg.rank = 0
        torch.manual_seed(cfg.seed)
        mon = VAMonitor(cfg, lambda *_: None, torch.device("cuda:0"))
        img, aud = _batch(B)
        sl = slice(rank * b, (rank + 1) * b)
        adjust_learning_rate(cfg.optimizer, mon.optimizer, mon.dataloader, 1)      # follows the whole batch: b * world
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


@pytest.mark.timeout(900)
def test_two_replicas_match_single_process_with_e4m3_contractions(tmp_path):
    """`running.fp8_gemm` under replicas.  The e4m3 forms share one scale per 32 consecutive token rows (round 6), so replicas equal one
    process bit for bit only when a rank's token count is a multiple of 32 -- true of every per-GPU batch in BASELINE.json (512 or 1024
    clips x 316 tokens); at this test's 8 clips x 31 tokens a rank's blocks sit differently, and the invariant holds to the budget below
    (the e4m3 rounding of a few boundary rows), as the loss and every parameter do."""
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    fp8 = ("running.fp8_gemm=True",)
    mp.spawn(_run, args=(1, 0, one, "gloo", fp8), nprocs=1, join=True)
    mp.spawn(_run, args=(2, _free_port(), two, "gloo", fp8), nprocs=2, join=True)
    a, b = torch.load(one), torch.load(two)
    assert abs(a["loss"] - b["loss"]) < 1e-5, (a["loss"], b["loss"])
    for k in a["params"]:
        pa, pb = a["params"][k], b["params"][k]
        err = float((pa - pb).abs().max())
        assert err <= 1e-6 + 2e-4 * float(pa.abs().max()), (k, err)


@pytest.mark.timeout(900)
def test_two_replicas_e4m3_on_block_aligned_ranks(tmp_path, monkeypatch):
    """The claim of DESIGN.md section 6 (i): when a rank's token count is a multiple of 32 -- here 32 clips x 31 tokens = 31 blocks, as
    512 x 316 and 1024 x 316 are -- a rank's e4m3 quantisation blocks are the blocks the one-process run has at the same rows, and
    two replicas take the one-process step as tightly as the bf16 towers do (fp32 summation order of the reduced gradients)."""
    monkeypatch.setenv("VIPANT_TEST_GLOBAL_BATCH", "64")
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    fp8 = ("running.fp8_gemm=True",)
    mp.spawn(_run, args=(1, 0, one, "gloo", fp8), nprocs=1, join=True)
    mp.spawn(_run, args=(2, _free_port(), two, "gloo", fp8), nprocs=2, join=True)
    a, b = torch.load(one), torch.load(two)
    assert abs(a["loss"] - b["loss"]) < 1e-6, (a["loss"], b["loss"])
    worst = 0.0
    for k in a["params"]:
        pa, pb = a["params"][k], b["params"][k]
        worst = max(worst, float((pa - pb).abs().max()) / max(float(pa.abs().max()), 1e-30))
    print(f"e4m3, block-aligned ranks: loss {a['loss']:.7f} / {b['loss']:.7f}, worst parameter difference {worst:.2e} of its scale")
    assert worst <= 2e-5, worst


@pytest.mark.timeout(900)
def test_two_replicas_with_micro_batches_match_single_process(tmp_path):
    """`running.micro_batch` under replicas (the micro-batches accumulate locally, the accumulated gradients are reduced ONCE; a
    parameter's reduced slices are summed): two replicas running two micro-batches each equal one process on the whole batch."""
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    mp.spawn(_run, args=(1, 0, one), nprocs=1, join=True)
    mp.spawn(_run, args=(2, _free_port(), two, "gloo", ("running.micro_batch=4",)), nprocs=2, join=True)
    a, b = torch.load(one), torch.load(two)
    assert abs(a["loss"] - b["loss"]) < 1e-5, (a["loss"], b["loss"])
    for k in a["params"]:
        pa, pb = a["params"][k], b["params"][k]
        err = float((pa - pb).abs().max())
        assert err <= 1e-6 + 2e-4 * float(pa.abs().max()), (k, err)


@pytest.mark.timeout(900)
def test_eight_replicas_match_single_process(tmp_path):
    """World 8, the replica count of BASELINE.json configs[3] / configs[4] (per-rank batch 2 here): same invariant."""
    one, eight = str(tmp_path / "one.pt"), str(tmp_path / "eight.pt")
    mp.spawn(_run, args=(1, 0, one), nprocs=1, join=True)
    mp.spawn(_run, args=(8, _free_port(), eight), nprocs=8, join=True)
    a, b = torch.load(one), torch.load(eight)
    assert abs(a["loss"] - b["loss"]) < 1e-5, (a["loss"], b["loss"])
    for k in a["params"]:
        pa, pb = a["params"][k], b["params"][k]
        err = float((pa - pb).abs().max())
        assert err <= 1e-6 + 2e-4 * float(pa.abs().max()), (k, err)


@pytest.mark.timeout(900)
def test_four_replicas_match_single_process(tmp_path):
    """The same invariant at world 4 (per-rank batch 4: row offsets 0 / 4 / 8 / 12 of the global similarity matrix -- strips that do
    not start on a multiple of 8 -- four gradient buckets per block in flight): four replicas sharing the GPU over gloo take the step
    of one process fed the concatenated batch."""
    one, four = str(tmp_path / "one.pt"), str(tmp_path / "four.pt")
    mp.spawn(_run, args=(1, 0, one), nprocs=1, join=True)
    mp.spawn(_run, args=(4, _free_port(), four), nprocs=4, join=True)
    a, b = torch.load(one), torch.load(four)
    assert abs(a["loss"] - b["loss"]) < 1e-5, (a["loss"], b["loss"])
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



# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[2]: AT fine-tuning, "DDP grad all-reduce only" -- every replica scores its own b-way negatives and the
# objective is the mean over replicas of the per-replica loss (what the reference's ddp mode would compute).
AT_OV = ("+running=trimodal monitor=VALMonitor worker=CVALP mode=ddp eval=False num_gpus=1 +model/image=vit_val "
         "+model/audio=vit_val +model/text=transformer_val +model/loss=ce_val +optimizer=standard +running/audio=default "
         "model.audio.pre_encoder.stride=[16,24] running.siamese.alive=True running.imagine=False model.loss.va=False "
         "model.image.encoder.layers=2 model.text.encoder.layers=2 running.audio.max_len=256 running.audio.num_mel_bins=64 "
         "running.batch_size=8 running.epochs=2 running.synthetic_steps=1 running.save_epoch=False optimizer.warmup_epoch=1 "
         "+running.negatives=local").split()


def _run_at_local(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vipant_amd.config import compose
        from vipant_amd.module import adjust_learning_rate
        from vipant_amd.monitor import VALMonitor
        cfg = compose(AT_OV + ([f"seed={os.environ['VIPANT_TEST_SEED']}"] if os.environ.get("VIPANT_TEST_SEED") else [])
                      + [f"running.last_block_rows={os.environ.get('VIPANT_TEST_LAST_ROWS', 'True')}"])
        cfg.rank = rank                              # the synthetic loader seeds per rank: every replica has its own batch
        torch.cuda.set_device(0)
        torch.manual_seed(cfg.seed)
        mon = VALMonitor(cfg, lambda *_: None, torch.device("cuda:0"))
        images, audios, text, _, _ = mon.make_batch(next(iter(mon.dataloader)))
        init = {k: v.detach().cpu().clone() for k, v in mon.model.state_dict().items()}
        adjust_learning_rate(cfg.optimizer, mon.optimizer, mon.dataloader, 1)
        lrs = [g["lr"] for g in mon.optimizer.param_groups]
        loss = mon.step(images, audios, text)        # the product path: nothing here rescales losses, gradients or LR
        torch.cuda.synchronize()
        after = {k: v.detach().cpu().clone() for k, v in mon.model.named_parameters() if v.requires_grad}
        torch.save({"loss": float(loss.detach()), "init": init, "after": after, "lrs": lrs,
                    "audios": audios.cpu(), "text": text.cpu()}, f"{out}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("rows", [True, False], ids=["rows", "fullblock"])
def test_two_replicas_local_negatives_match_oracle_mean_objective(tmp_path, monkeypatch, rows):
    """cfg3 product path (VALMonitor.step, `running.negatives=local`) on two replicas against the CPU oracle taking one
    LARS step on  mean_r InfoNCE(audio_r, text_r).  Biases, LayerNorm parameters and logit_scale have no trust ratio
    (lars.py:58-66), so a SUM-instead-of-mean reduction or a per-replica LR would show as a factor `world` in their update."""
    from oracle import ref_cpu as R
    out = str(tmp_path / "at")
    monkeypatch.setenv("VIPANT_TEST_LAST_ROWS", str(rows))
    mp.spawn(_run_at_local, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    for k in r0["after"]:                            # replicas stay in lock-step
        assert torch.equal(r0["after"][k], r1["after"][k]), k
    assert not torch.equal(r0["audios"], r1["audios"])
    lw, lb = R.adjust_learning_rate(1, epochs=2, steps_per_epoch=1, warmup_epoch=1, batch_size=16, lr_weight=0.2, lr_bias=0.0048)
    assert abs(r0["lrs"][0] - lw) < 1e-12 and abs(r0["lrs"][1] - lb) < 1e-12
    init = r0["init"]
    asd = {k[len("audio_head."):]: v.clone().requires_grad_() for k, v in init.items() if k.startswith("audio_head.")}
    tsd = {k[len("text_head."):]: v for k, v in init.items() if k.startswith("text_head.")}
    ls = init["loss_head.loss_head_al.logit_scale"].clone().requires_grad_()
    stride, S, pr = R.vit_position_resolution([256, 64], 32, [16, 24])
    total, per_rank = 0.0, []
    for r in (r0, r1):
        lr_ = R.cvalp_forward(torch.zeros(8, 1, 1, 1), r["audios"], r["text"], audio_sd=asd, text_sd=tsd, loss="valce",
                              scales={"al": ls}, loss_flags=dict(va=False, lv=False, al=True),
                              audio_cfg=dict(width=768, layers=2, stride=stride, position_resolution=pr),
                              text_cfg=dict(width=512, layers=2, ctx_len=77))
        per_rank.append(float(lr_))
        total = total + lr_ / 2
    total.backward()
    assert abs(r0["loss"] - per_rank[0]) < 5e-3 and abs(r1["loss"] - per_rank[1]) < 5e-3, (r0["loss"], r1["loss"], per_rank)
    ref = {f"audio_head.{k}": v for k, v in asd.items()}
    ref["loss_head.loss_head_al.logit_scale"] = ls
    assert set(ref) == set(r0["after"])
    dirs = []
    for k, p in ref.items():
        p_ref, _ = R.lars_step(p.detach(), p.grad, torch.zeros_like(p), lw if p.ndim > 1 else lb)
        d_ref, d_hip = (p_ref - p.detach()).double(), (r0["after"][k] - p.detach()).double()
        if float(d_ref.norm()) == 0.0:
            assert float(d_hip.norm()) == 0.0, k
            continue
        ratio = float(d_hip.norm() / d_ref.norm())
        assert abs(ratio - 1) < 6e-2, (k, ratio)                         # bf16 towers: a few % on a gradient norm; never 2x
        dirs.append(float((d_hip - d_ref).norm() / d_ref.norm()))
        if os.environ.get("VIPANT_TEST_VERBOSE"): print("DIR %-60s %.4f" % (k, dirs[-1]))
    dirs.sort()
    # update direction: bf16 towers against the fp32 oracle at b = 8.  Every tensor's error is dominated by ONE shared draw -- the
    # error of d loss / d features, the feature noise amplified by the logit scale -- so the median moves as a whole with the
    # rounding pattern: observed over six seeds (VIPANT_TEST_SEED) 0.021 ... 0.044 with the full last block and 0.021 ... 0.066 with
    # `running.last_block_rows` (this seed: 0.044 / 0.066, the noisiest of the six in both modes); the noisiest tensor
    # (near-cancelling LayerNorm-weight / bias gradients) 0.08 ... 0.23
    print("update-direction error vs the fp32 oracle: median %.4f max %.4f" % (dirs[len(dirs) // 2], dirs[-1]))
    # the read-out-row path (exact up to rounding: dh + dq rows are rounded to bf16 twice) has 1.2x its
    # worst observed seed; the full last block 1.2x ITS worst: 0.0506 since the loss takes the row-block kernels at this batch
    # (round 4) -- the two InfoNCE paths are equally accurate against fp64 (dx rel-L2 2.1e-3 both, tools/nce_accuracy.py), the
    # statistic moves with any change of rounding pattern
    assert dirs[len(dirs) // 2] < (8e-2 if rows else 6e-2) and dirs[-1] < 0.3, (rows, dirs[len(dirs) // 2], dirs[-1])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("script", ["va", "at"])
def test_bench_two_ranks_share_the_gpu(script):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), on the one GPU of this box:
    the ranks share cuda:0 and talk over gloo (`VIPANT_DIST_BACKEND=gloo`; RCCL refuses two ranks on one device).  Checks the
    N > 1 code path of the bench end to end -- group set-up, feature all-gather, bucketed gradient reduction, barrier, max-over-
    ranks timing, the one JSON line -- not its speed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VIPANT_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "16", "--frames", "256", "--mels", "64", "--layers", "2", "--no-cpu-baseline", "--script", script]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=800, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]                 # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 32 and out["scaling"] == "weak"
    assert out["value"] > 0 and abs(out["value"] - 32 / (out["ms_per_step"] * 1e-3)) < 1e-2 * out["value"]
    import math
    nneg = 32 if script == "va" else 16            # va: global (all-gathered) negatives; at: every rank scores its own 16
    assert math.isfinite(out["loss"]) and abs(out["loss"] - 2 * math.log(nneg)) < 1.0, out["loss"]


@pytest.mark.timeout(900)
def test_bench_gpus_flag_starts_its_own_replicas():
    """`python bench.py --gpus 2` with NO launcher around it (the driver's 1-GPU command shape with another N): the process starts
    two replicas itself (vipant_amd/launch.py: child torch.distributed.run, the parent never touches the GPU) and relays rank 0's
    line.  Over gloo the two replicas share this box's GPU; with the RCCL backend the same command must refuse -- non-zero exit
    naming the GPU count -- instead of measuring one GPU under the label of two."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
           "--frames", "256", "--mels", "64", "--layers", "2", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=800, env=dict(clean, VIPANT_DIST_BACKEND="gloo"), cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["rccl"] == "gloo" and out["config"]["global_batch"] == 32
    if torch.cuda.device_count() < 2:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=clean, cwd=root)
        assert res.returncode != 0 and not [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
        assert "asked for 2 GPUs but 1 is visible" in res.stderr, res.stderr[-1000:]


@pytest.mark.timeout(900)
def test_train_entry_dp_mode_starts_replicas_and_takes_the_one_process_step(tmp_path):
    """`num_gpus=2 mode=dp` (run_bimodal_va.sh:23) is the reference's dp step: ONE loader batch of `running.batch_size`, split over
    the GPUs, loss over the whole batch (cvap/model/cvalp.py:41-61).  train.py starts the two replicas itself; their two steps
    must be the steps of `num_gpus=1 mode=dp` on the same batch size: same logged losses, same weights in the checkpoint."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env = dict(clean, VIPANT_DIST_BACKEND="gloo")

    def run(ngpu, name):
        ov = (f"+running=bimodal worker=CVALP port=1 num_gpus={ngpu} mode=dp num_proc=2 eval=False verbose=False "
              f"alias_root={tmp_path} model_name={name} +model/image=vit_val +model/audio=vit_val +model/text=dummy +model/loss=ce "
              "+optimizer=standard +running/audio=default model.audio.pre_encoder.in_channels=3 "
              "model.audio.pre_encoder.stride=[16,24] optimizer.warmup=False running.audio.norms=[-4.93839311,5.75751113] "
              "model.image.encoder.layers=1 running.audio.max_len=256 running.audio.num_mel_bins=64 running.batch_size=8 "
              "running.synthetic_steps=2 running.epochs=1 running.peep_rate=1 running.frame_emb=synthetic running.save_rate=2").split()
        res = subprocess.run([sys.executable, os.path.join(root, "train.py")] + ov, capture_output=True, text=True, timeout=800,
                             env=env, cwd=root)
        assert res.returncode == 0, res.stderr[-3000:]
        log = (tmp_path / name / "train_0.out").read_text()
        losses = [float(m.group(1)) for m in re.finditer(r" loss ([0-9.]+) ", log)]
        ck = torch.load(tmp_path / name / "00000002.pth", weights_only=False)
        return log, losses, ck

    log1, loss1, ck1 = run(1, "one")
    log2, loss2, ck2 = run(2, "two")
    assert "World size: 1; rank: 0" in log1 and "World size: 2; rank: 0" in log2 and (tmp_path / "two" / "train_1.out").exists()
    assert "per-process batch 4 x 2 replica(s) = global batch 8" in log2 and "global batch 8" in log1
    assert len(loss1) == 2 and loss1 == loss2, (loss1, loss2)                 # logged with three decimals
    for k, v in ck1["model"][1].items():
        err = float((v.float() - ck2["model"][1][k].float()).abs().max())
        assert err <= 1e-6 + 2e-4 * float(v.float().abs().max()), (k, err)


@pytest.mark.timeout(900)
def test_train_entry_two_ranks_share_the_gpu(tmp_path):
    """`train.py` with the VA launch script's overrides under torch.distributed.run, two ranks on this box's one GPU (gloo):
    group set-up from the environment, per-rank synthetic batches, the replica exchange steps inside Monitor.epoch, rank-0 logging
    and checkpoint; both ranks must end with the same weights (rank 1 writes a second checkpoint for the comparison)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VIPANT_DIST_BACKEND="gloo")
    ov = (f"+running=bimodal worker=CVALP mode=ddp eval=False verbose=False alias_root={tmp_path} model_name=t num_gpus=2 "
          "+model/image=vit_val +model/audio=vit_val +model/text=dummy +model/loss=ce +optimizer=standard +running/audio=default "
          "model.audio.pre_encoder.in_channels=3 model.audio.pre_encoder.stride=[16,24] model.image.encoder.layers=1 "
          "running.audio.max_len=256 running.audio.num_mel_bins=64 running.batch_size=4 running.synthetic_steps=2 "
          "running.epochs=1 running.peep_rate=1 running.frame_emb=synthetic running.save_rate=2").split()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "train.py")] + ov
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=800, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    log = (tmp_path / "t" / "train_0.out").read_text()
    assert "World size: 2; rank: 0" in log and "samples/s" in log and "Saving the checkpoint" in log
    assert (tmp_path / "t" / "train_1.out").exists()
    ck = torch.load(tmp_path / "t" / "00000002.pth", weights_only=False)
    assert len(ck["model"]) == 4 and "encoder.resblocks.0.attn.in_proj_weight" in ck["model"][1]
    lines = [ln for ln in log.splitlines() if "samples/s" in ln]
    assert len(lines) == 2 and "step 2" in lines[-1]
