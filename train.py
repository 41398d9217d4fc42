"""Entry point with the reference's command line (train.py + bash/run_bimodal_{va,at}.sh):

    python train.py +running=bimodal worker=CVALP mode=dp +model/image=vit_val +model/audio=vit_val \
        +model/text=dummy +model/loss=ce +optimizer=standard +running/audio=default key=value ...

`mode=dp`  : the reference's dp step (train.py:68-71, cvap/model/cvalp.py:41-61): ONE loader yields `running.batch_size` samples,
             the towers run on `num_gpus` GPUs with the batch split in contiguous chunks, the loss scores the whole batch.  Here
             data parallelism is one process per GPU, so `num_gpus=N mode=dp` starts N replicas by itself (a child
             `torch.distributed.run`, vipant_amd/launch.py): each takes its `batch_size / N` chunk of the loader's batch, the
             features are all-gathered (global negatives), the gradients all-reduced over RCCL.  `num_gpus=1`: one process.
`mode=ddp` : one replica per GPU, `running.batch_size` PER replica.  Launch with torchrun (RANK / LOCAL_RANK / WORLD_SIZE /
             MASTER_* in the env); backend "nccl" is RCCL over xGMI on ROCm.
"""
import os
import sys

from vipant_amd import launch
from vipant_amd.config import compose, to_yaml

# torch, the HIP library and the trainer are imported by train() AFTER the decision whether this process is a replica or the parent
# that starts the replicas: that parent must never touch the GPU
torch = dist = monitors = seed_all_rng = setup_logger = None


def main(cfg, rank, device, manager):
    cfg.rank = rank
    seed_all_rng(cfg.seed)      # same seed on every rank, as the reference (train.py:40)
    output_dir = f"{cfg.alias_root}/{cfg.model_name}"
    logger = setup_logger(output_dir=output_dir, rank=rank, output=output_dir)
    if cfg.verbose or not cfg.eval:
        logger.info(f"\n\n{to_yaml(cfg)}")
    if cfg.blockprint:
        sys.stdout = open(os.devnull, "w")
    logger.info("World size: {}; rank: {}".format(dist.get_world_size() if dist.is_initialized() else 1, rank))
    monitor_cls = getattr(monitors, manager) if isinstance(manager, str) else manager
    monitor_cls(cfg, logger.info, device).learn()


def dp_chunk(cfg, world):
    """`mode=dp` under `world` replicas: `running.batch_size` is the loader's (global) batch, as in the reference; a replica's towers
    take a contiguous chunk of it (data_parallel's scatter, cvalp.py:41-56) and the loss scores all of it."""
    B = int(cfg.running.batch_size)
    if B % world:
        raise ValueError(f"mode=dp: running.batch_size={B} does not split evenly over num_gpus={world} replicas")
    cfg.running.batch_size = B // world
    cfg.optimizer.batch_size = B // world        # the LARS schedule multiplies by the replica count again (module/lars.py)
    cfg.running.dp_chunk = True                  # the synthetic loader: chunk `rank` of ONE global batch, not a batch per rank
    cfg.running.negatives = "global"


def train(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    cfg = compose(argv)
    ngpu = int(cfg.get("num_gpus", 1) or 1)
    if cfg.mode == "dp" and ngpu > 1 and not launch.under_launcher():
        # the reference's dp launch (run_bimodal_va.sh:23 `num_gpus=$ngpu mode=dp`): start the replicas, relay their exit code
        rc = launch.replicas(os.path.abspath(__file__), argv, ngpu)
        if rc:
            sys.exit(rc)
        return
    global torch, dist, monitors, seed_all_rng, setup_logger
    import torch
    import torch.distributed as dist
    from vipant_amd import monitor as monitors
    from vipant_amd.util import seed_all_rng, setup_logger
    if cfg.mode == "dp" and launch.under_launcher() and int(os.environ["WORLD_SIZE"]) > 1:
        dp_chunk(cfg, int(os.environ["WORLD_SIZE"]))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1 or cfg.mode == "ddp" and "RANK" in os.environ:
        backend = os.environ.get("VIPANT_DIST_BACKEND", "nccl")     # "nccl" = RCCL over xGMI; "gloo" only for shared-GPU tests
        local_rank, ndev = int(os.environ.get("LOCAL_RANK", 0)), max(torch.cuda.device_count(), 1)
        if backend == "nccl" and local_rank >= ndev:                # RCCL needs one GPU per rank: fail here, not inside its init
            raise RuntimeError(f"LOCAL_RANK {local_rank} but only {ndev} visible GPU(s): RCCL takes one GPU per rank")
        local_rank %= ndev                                          # shared-GPU gloo tests: ranks wrap around the devices
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        try:
            main(cfg, dist.get_rank(), torch.device("cuda", local_rank), cfg.monitor)
        finally:
            dist.destroy_process_group()
    else:
        torch.cuda.set_device(0)
        main(cfg, 0, torch.device("cuda", 0), cfg.monitor)


if __name__ == "__main__":
    train()
