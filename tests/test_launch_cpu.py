"""`python bench.py --gpus N` / `python train.py num_gpus=N mode=dp` start their own N replicas (vipant_amd/launch.py): the parent
stays free of torch and the GPU, relays output and exit code, and refuses -- instead of quietly measuring fewer GPUs -- when RCCL
cannot have one GPU per replica.  The dp chunking of the loader batch (the reference's data_parallel scatter,
cvap/model/cvalp.py:41-61) is checked under gloo, world 2, on CPU."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLEAN = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


@pytest.mark.timeout(300)
@pytest.mark.parametrize("entry", ["bench", "train"])
def test_more_gpus_than_visible_is_refused(entry):
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    if entry == "bench":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"]
    else:
        cmd = [sys.executable, os.path.join(ROOT, "train.py"), "+running=bimodal", "worker=CVALP", "num_gpus=2", "mode=dp", "eval=False",
               "+model/image=vit_val", "+model/audio=vit_val", "+model/text=dummy", "+model/loss=ce", "+optimizer=standard",
               "+running/audio=default"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=dict(CLEAN, VIPANT_DIST_BACKEND="nccl"), cwd=ROOT)
    assert res.returncode != 0
    assert "asked for 2 GPUs but" in res.stderr and "visible" in res.stderr, res.stderr[-1000:]
    assert '{"metric"' not in res.stdout


@pytest.mark.timeout(120)
def test_parent_of_the_replicas_never_imports_torch():
    """The launching parent must not initialise the GPU: it does not even import torch (nor the HIP library's binding)."""
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from vipant_amd import launch
        seen = []
        launch.replicas = lambda script, argv, n, extra_env=None: seen.append((script, list(argv), n)) or 0
        import bench
        sys.argv = ["bench.py", "--gpus", "4", "--steps", "3"]
        try:
            bench.main()
        except SystemExit as e:
            assert e.code == 0
        import train
        train.train(["+running=bimodal", "worker=CVALP", "num_gpus=8", "mode=dp", "eval=False", "+model/image=vit_val",
                     "+model/audio=vit_val", "+model/text=dummy", "+model/loss=ce", "+optimizer=standard", "+running/audio=default"])
        assert [s[2] for s in seen] == [4, 8] and seen[0][1] == ["--gpus", "4", "--steps", "3"], seen
        assert seen[0][0].endswith("bench.py") and seen[1][0].endswith("train.py") and "num_gpus=8" in seen[1][1]
        bad = [m for m in ("torch", "vipant_amd._ffi", "vipant_amd.ops") if m in sys.modules]
        assert not bad, bad
        print("PARENT_CLEAN")
    """)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=100, env=CLEAN, cwd=ROOT)
    assert res.returncode == 0 and "PARENT_CLEAN" in res.stdout, res.stderr[-2000:]


@pytest.mark.timeout(300)
def test_replicas_relay_output_and_exit_code(tmp_path):
    script = tmp_path / "probe.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        import torch, torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.ones(1) * (dist.get_rank() + 1)
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print("SUM", int(t), "WORLD", dist.get_world_size(), "MARK", os.environ.get("VIPANT_LAUNCHED_REPLICAS"), flush=True)
        dist.destroy_process_group()
        sys.exit(int(sys.argv[1]))
    """))
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); from vipant_amd import launch; "
            f"sys.exit(launch.replicas({str(script)!r}, [sys.argv[1]], 2))")
    env = dict(CLEAN, VIPANT_DIST_BACKEND="gloo")
    ok = subprocess.run([sys.executable, "-c", code, "0"], capture_output=True, text=True, timeout=140, env=env)
    assert ok.returncode == 0 and "SUM 3 WORLD 2 MARK 2" in ok.stdout, ok.stderr[-2000:]
    bad = subprocess.run([sys.executable, "-c", code, "3"], capture_output=True, text=True, timeout=140, env=env)
    assert bad.returncode != 0 and "SUM 3 WORLD 2" in bad.stdout


def _chunks(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        import train as entry
        from vipant_amd.config import compose
        from vipant_amd.monitor import SyntheticLoader
        cfg = compose("+running=bimodal worker=CVALP num_gpus=2 mode=dp eval=False +model/image=vit_val +model/audio=vit_val "
                      "+model/text=dummy +model/loss=ce +optimizer=standard +running/audio=default running.audio.max_len=64 "
                      "running.audio.num_mel_bins=32 running.batch_size=6 running.frame_emb=synthetic".split())
        entry.dp_chunk(cfg, world)
        assert cfg.running.batch_size == 3 and cfg.optimizer.batch_size == 3 and cfg.running.negatives == "global"
        batches = list(SyntheticLoader(cfg, 2, False, rank))
        torch.save([(b[0], b[1], b[4]) for b in batches], f"{out}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp_replicas_take_contiguous_chunks_of_one_loader_batch(tmp_path):
    from vipant_amd import launch
    from vipant_amd.config import compose
    from vipant_amd.monitor import SyntheticLoader
    out = str(tmp_path / "chunks")
    mp.spawn(_chunks, args=(2, launch.free_port(), out), nprocs=2, join=True)
    cfg = compose("+running=bimodal worker=CVALP num_gpus=1 mode=dp eval=False +model/image=vit_val +model/audio=vit_val "
                  "+model/text=dummy +model/loss=ce +optimizer=standard +running/audio=default running.audio.max_len=64 "
                  "running.audio.num_mel_bins=32 running.batch_size=6 running.frame_emb=synthetic".split())
    whole = list(SyntheticLoader(cfg, 2, False, 0))
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    for step in range(2):
        for k in (0, 1):
            assert torch.equal(torch.cat([r0[step][k], r1[step][k]]), whole[step][k])
        assert r0[step][2] + r1[step][2] == whole[step][4]
    import train as entry
    with pytest.raises(ValueError, match="does not split evenly"):
        cfg.running.batch_size = 7
        entry.dp_chunk(cfg, 2)
