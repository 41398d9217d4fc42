"""Start N data-parallel replicas of an entry script (bench.py, train.py) from a plain `python script.py ...` command.

The reference's `mode=dp` feeds `num_gpus` GPUs from ONE process (torch.nn.parallel.data_parallel, cvap/model/cvalp.py:41-61,
train.py:68-71); here data parallelism is one process per GPU, so a command that names N GPUs and was not started by a launcher
starts its own: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>
script.py <same args>` as a CHILD process.  The parent never touches the GPU -- this module imports neither torch nor the HIP
library, the GPU count comes from a throw-away child -- and never exec()s: it waits, relays the children's output (inherited
stdout / stderr: rank 0 prints) and returns their exit code.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys

ENV_MARK = "VIPANT_LAUNCHED_REPLICAS"       # set for the children: "I was started by launch.replicas, N of me exist"


def under_launcher() -> bool:
    """Was this process started by a launcher (torchrun / torch.distributed.run / launch.replicas)?"""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus() -> int:
    """GPUs visible to a fresh process of this environment (HIP_/ROCR_/CUDA_VISIBLE_DEVICES honoured), counted in a throw-away
    child so that the calling process stays free of any GPU state."""
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        raise RuntimeError("could not count the visible GPUs: " + (r.stderr or r.stdout)[-500:])


def replicas(script: str, argv, n: int, extra_env=None) -> int:
    """Run `script argv` as n replicas on this node and return their exit code.  With the RCCL backend (default) every replica
    needs its own GPU: fewer than n visible GPUs is an error, never a silent fall-back to fewer replicas.
    `VIPANT_DIST_BACKEND=gloo` (tests on a one-GPU box) lets the replicas share devices round-robin."""
    backend = os.environ.get("VIPANT_DIST_BACKEND", "nccl")
    if backend == "nccl":
        have = visible_gpus()
        if have < n:
            sys.stderr.write(f"[vipant_amd.launch] {os.path.basename(script)} was asked for {n} GPUs but {have} "
                             f"{'is' if have == 1 else 'are'} visible on this node: RCCL takes one GPU per replica; not "
                             "falling back to fewer replicas\n")
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script, *argv]
    env = dict(os.environ, **{ENV_MARK: str(n)}, **(extra_env or {}))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode
