"""Build libvipant_hip.so (gfx950) and, optionally, nothing else: `python -m vipant_amd.build`.

hipcc cross-compiles without a GPU; the .so stays in-tree (vipant_amd/lib/) so it travels with the repo
snapshot to the GPU box.  Rebuilds only when a source is newer than the library.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libvipant_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "vipant_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [HIPCC, *FLAGS, "-o", LIB, *sources()]
    if verbose:
        print("[vipant_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
