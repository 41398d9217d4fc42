"""Seeding and logging for the trainer entry (the reference's helpers of the same names: cvap/util/__init__.py:8-37)."""
import logging
import os
import random

import numpy
import torch
import torch.distributed as dist

__all__ = ["seed_all_rng", "setup_logger"]


def seed_all_rng(seed):
    """Python, NumPy and torch generators, same seed on every rank (train.py:40 of the reference)."""
    for seeder in (random.seed, numpy.random.seed, torch.manual_seed):
        seeder(seed)


def setup_logger(output_dir=None, name="cvap", rank=0, output=None):
    """Logger `name`: console output on rank 0 only, plus `<output_dir>/train_<rank>.out` on every rank when `output` is
    given.  Rank 0 creates the directory; the other ranks wait for it at a barrier."""
    log = logging.getLogger(name)
    log.setLevel(logging.INFO)
    log.propagate = False
    for old in list(log.handlers):
        log.removeHandler(old)
    fmt = logging.Formatter("%(asctime)s - %(levelname)s - %(message)s")

    def attach(handler):
        handler.setLevel(logging.INFO)
        handler.setFormatter(fmt)
        log.addHandler(handler)

    if rank == 0:
        attach(logging.StreamHandler())
    if output_dir is None:
        return log
    if os.path.isdir(output_dir):
        log.info(f"Warning: the folder {output_dir} exists.")
    elif rank == 0:
        log.info(f"Creating {output_dir}")
        os.makedirs(output_dir, exist_ok=True)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if output is not None:
        attach(logging.FileHandler(os.path.join(output_dir, f"train_{rank}.out"), mode="w"))
    return log
