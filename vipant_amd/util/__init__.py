"""Seeding / logging helpers with the reference's signatures (cvap/util/__init__.py:8-62)."""
import logging
import os
import random

import numpy
import torch
import torch.distributed as dist


def seed_all_rng(seed):
    random.seed(seed)
    numpy.random.seed(seed)
    torch.manual_seed(seed)


def setup_logger(output_dir=None, name="cvap", rank=0, output=None):
    """Rank-0 console handler + one `train_{rank}.out` file per rank (cvap/util/__init__.py:13-37)."""
    logger = logging.getLogger(name)
    logger.setLevel(logging.INFO)
    logger.propagate = False
    logger.handlers.clear()
    formatter = logging.Formatter("%(asctime)s - %(levelname)s - %(message)s")
    if rank == 0:
        console = logging.StreamHandler()
        console.setLevel(logging.INFO)
        console.setFormatter(formatter)
        logger.addHandler(console)
    if output_dir is not None:
        if os.path.exists(output_dir):
            logger.info(f"Warning: the folder {output_dir} exists.")
        elif rank == 0:
            logger.info(f"Creating {output_dir}")
            os.makedirs(output_dir, exist_ok=True)
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        if output is not None:
            handler = logging.FileHandler(os.path.join(output_dir, f"train_{rank}.out"), "w")
            handler.setLevel(logging.INFO)
            handler.setFormatter(formatter)
            logger.addHandler(handler)
    return logger


def numel(model: torch.nn.Module, trainable: bool = False):
    parameters = [p for p in model.parameters() if p.requires_grad or not trainable]
    return sum(p.numel() for p in {p.data_ptr(): p for p in parameters}.values())


def detect_nan(x):
    return torch.isnan(x).any(), torch.isinf(x).any()


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.sum = self.count = self.avg = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
