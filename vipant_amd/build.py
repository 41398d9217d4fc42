"""Build libvipant_hip.so (gfx950): `python -m vipant_amd.build [--force]`.

hipcc cross-compiles without a GPU; the .so stays in-tree (vipant_amd/lib/) so it travels with the repo
snapshot to the GPU box.  Every .hip file is compiled to its own object (in parallel, only when it or a header is
newer than the object) and the objects are linked into the library.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libvipant_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# per-file additions.  gemm_nt.hip: the e4m3 K-loop has no dependency between the MFMAs of its two barrier intervals, and LLVM's
# machine sinker moves all 32 of a K-tile below both barriers (into the loop latch), which undoes the ping-pong schedule and
# spills 25 registers: 1.18 -> 1.69 PFLOP/s at K = 1024 with the pass off; the bf16 kernels of the file compile to the same code.
EXTRA_CFLAGS = {"gemm_nt.hip": ["-mllvm", "-disable-machine-sink"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "vipant_hip.h")]


def _obj(src: str) -> str:
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _older(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def stale() -> bool:
    return _older(LIB, sources() + _headers())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    todo = [s for s in sources() if force or _older(_obj(s), [s] + hdrs)]

    def compile_one(src):
        cmd = [HIPCC, *CFLAGS, *EXTRA_CFLAGS.get(os.path.basename(src), []), "-c", src, "-o", _obj(src)]
        if verbose:
            print("[vipant_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as pool:
        list(pool.map(compile_one, todo))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *[_obj(s) for s in sources()]]
    if verbose:
        print("[vipant_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
