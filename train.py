"""Entry point with the reference's command line (train.py + bash/run_bimodal_{va,at}.sh):

    python train.py +running=bimodal worker=CVALP mode=dp +model/image=vit_val +model/audio=vit_val \
        +model/text=dummy +model/loss=ce +optimizer=standard +running/audio=default key=value ...

`mode=dp`  : one process, one MI355X (the reference's dp mode would replicate over `num_gpus` with
             torch.nn.parallel.data_parallel; here data parallelism is always one process per GPU).
`mode=ddp` : one replica per GPU.  Launch with torchrun (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env);
             backend "nccl" is RCCL over xGMI on ROCm.
"""
import os
import sys

import torch
import torch.distributed as dist

from vipant_amd import monitor as monitors
from vipant_amd.config import compose, to_yaml
from vipant_amd.util import seed_all_rng, setup_logger


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


def train(argv=None):
    cfg = compose(sys.argv[1:] if argv is None else argv)
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
