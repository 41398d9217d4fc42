"""Build libvipant_hip.so (gfx950): `python -m vipant_amd.build [--force]`.

hipcc cross-compiles without a GPU; the .so stays in-tree (vipant_amd/lib/) so it travels with the repo
snapshot to the GPU box.  Every .hip file is compiled to its own object (in parallel, only when it or a header is
newer than the object) and the objects are linked into the library.
"""
from __future__ import annotations

import glob
import json
import os
import re
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
# attention.hip: the single-pass backward runs one wave per SIMD with 512 registers; for such kernels LLVM selects every MFMA with an
# AGPR destination, and each S / dP score then costs a v_accvgpr_read before the VALU can touch it (180 of 400 VALU instructions per
# step, and the kernel is VALU-issue bound).  With the VGPR form forced, accumulators land where the arithmetic needs them and the
# resident operand fragments take the AGPRs (pinned there by "+a" constraints in the source); the two-waves-per-SIMD kernels of the
# file were VGPR-form already.
# gemm_nt.hip, second option: the ticket walk draws a tile with ONE atomic add per workgroup whose result is needed a tile later; the
# atomic optimizer rewrites a uniform atomic as "first active lane adds, s_waitcnt vmcnt(0), readfirstlane" -- a full wait (every
# LDS-DMA piece in flight included) at the place where the draw is issued.
EXTRA_CFLAGS = {"gemm_nt.hip": ["-mllvm", "-disable-machine-sink", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"], "attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-Wno-inline-asm"]}


# files whose device assembly is kept beside the object (lib/obj/<name>-hip-amdgcn-amd-amdhsa-gfx950.s): kernels with hand-counted
# `s_waitcnt vmcnt(n)` or inline-asm loads, whose correctness depends on what the compiler put between two instructions
# (tests/test_abi_cpu.py reads it)
KEEP_ISA = ("attention.hip", "gemm_nt.hip")


def isa_path(name: str) -> str:
    return os.path.join(OBJ, name[:-4] + "-hip-amdgcn-amd-amdhsa-gfx950.s")


_REMARK_ECHO = re.compile(r"^\s*(\d+ \||\|)")      # the source-line echo clang prints under each remark


def _resource_usage(stderr: str):
    """{kernel: {field: int}} from clang's -Rpass-analysis=kernel-resource-usage remarks (registers, spills, scratch, LDS)."""
    out, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|[A-Za-z \[\]/]+):\s*(\S+)", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = out.setdefault(val, {})
        elif cur is not None and val.lstrip("-").isdigit():
            cur[key] = int(val)
    return out


def _resource_usage_from_isa(path: str):
    """The same table from the `.amdgpu_metadata` note of a kept device assembly: under `-save-temps` clang emits no resource-usage
    remarks (the backend runs as a step of its own), but the note carries every kernel's register, spill and scratch counts."""
    out, cur = {}, None
    keys = {".vgpr_count": "VGPRs", ".agpr_count": "AGPRs", ".vgpr_spill_count": "VGPRs Spill", ".sgpr_spill_count": "SGPRs Spill",
            ".private_segment_fixed_size": "ScratchSize [bytes/lane]", ".group_segment_fixed_size": "LDS Size [bytes/block]"}
    text = open(path).read()
    start = text.find(".amdgpu_metadata")
    if start < 0:
        return out
    for entry in text[start:].split("\n  - ")[1:]:                    # one list item per kernel ('.args' items are nested deeper)
        rec, name = {}, None
        for line in entry.splitlines():
            m = re.match(r"\s*(\.[a-z_]+):\s*(\S+)\s*$", line)
            if not m:
                continue
            if m.group(1) == ".name" and line.startswith("    .name"):
                name = m.group(2)
            elif m.group(1) in keys and m.group(2).isdigit():
                rec[keys[m.group(1)]] = int(m.group(2))
        if name and "VGPRs" in rec:
            out[name] = rec
    return out


def resource_usage():
    """Per-kernel resource usage of the last build: {source file: {kernel: {field: int}}}."""
    res = {}
    for s in sources():
        p = _obj(s)[:-2] + ".resources.json"
        if os.path.exists(p):
            with open(p) as f:
                res[os.path.basename(s)] = json.load(f)
    return res


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
    return _older(LIB, sources() + _headers()) or any(not os.path.exists(isa_path(n)) for n in KEEP_ISA)


def _probe_flags():
    """`-mllvm -amdgpu-mfma-vgpr-form` is an internal LLVM option: on a toolchain without it every attention file would fail with
    'Unknown command line argument'.  Probe once; without the option the files still build (the 512-register kernels then get
    AGPR-destination MFMAs and run slower; the no-spill test still guards them)."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "p.hip")
        with open(src, "w") as f:
            f.write("#include <hip/hip_runtime.h>\n__global__ void k() {}\n")
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "--cuda-device-only", "-mllvm", "-amdgpu-mfma-vgpr-form", "-c", src, "-o",
                            os.path.join(d, "p.o")], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    if r.returncode != 0 and "nknown command line argument" in (r.stderr or ""):
        print("[vipant_amd.build] WARNING: this hipcc does not know -mllvm -amdgpu-mfma-vgpr-form; building the attention kernels "
              "without it (slower single-pass backward)", flush=True)
        for k, v in EXTRA_CFLAGS.items():
            EXTRA_CFLAGS[k] = [a for i, a in enumerate(v) if a != "-amdgpu-mfma-vgpr-form" and not (a == "-mllvm" and i + 1 < len(v) and v[i + 1] == "-amdgpu-mfma-vgpr-form")]


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    _probe_flags()
    hdrs = _headers()
    todo = [s for s in sources() if force or _older(_obj(s), [s] + hdrs) or not os.path.exists(_obj(s)[:-2] + ".resources.json")
            or (os.path.basename(s) in KEEP_ISA and not os.path.exists(isa_path(os.path.basename(s))))]

    def compile_one(src):
        base = os.path.basename(src)
        # KEEP_ISA files: `-save-temps=obj` leaves the device assembly of THIS compile beside the object (isa_path) -- the text the ISA
        # guards check is the text that was assembled into the shipped code, not the output of a second compile
        keep = ["-save-temps=obj"] if base in KEEP_ISA else []
        cmd = [HIPCC, *CFLAGS, *EXTRA_CFLAGS.get(base, []), *keep, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", _obj(src)]
        if verbose:
            print("[vipant_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        usage = _resource_usage(r.stderr)
        if base in KEEP_ISA and r.returncode == 0 and not usage and os.path.exists(isa_path(base)):
            usage = _resource_usage_from_isa(isa_path(base))
        other = "\n".join(l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l and not _REMARK_ECHO.match(l))
        if other.strip():
            sys.stderr.write(other + "\n")
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        with open(_obj(src)[:-2] + ".resources.json", "w") as f:
            json.dump(usage, f, indent=0)
        if base in KEEP_ISA:
            if not os.path.exists(isa_path(base)):
                raise RuntimeError(f"{base}: -save-temps=obj left no device assembly at {isa_path(base)}")
            for junk in glob.glob(os.path.join(OBJ, base[:-4] + "-h*")) + glob.glob(os.path.join(OBJ, base + "-h*")):
                if junk != isa_path(base):      # the other intermediates (preprocessed source, bitcode, host side, fat binary)
                    os.remove(junk)
            from . import isa_guard
            try:
                isa_guard.CHECKS[base](isa_path(base))
            except AssertionError as e:
                os.remove(_obj(src))            # no object, no library: a build whose hand-counted waits do not hold must not ship
                raise RuntimeError(f"ISA guard of {base} failed on this toolchain ({e}); VIPANT_GEMM_VARIANT=4194304 selects the static "
                                   "tile walk of the NT kernels, the attention guard has no fallback") from e
        for name, u in usage.items():
            if u.get("VGPRs Spill", 0) or u.get("ScratchSize [bytes/lane]", 0):
                print(f"[vipant_amd.build] WARNING {os.path.basename(src)}: {name} spills "
                      f"({u.get('VGPRs Spill', 0)} VGPRs, {u.get('ScratchSize [bytes/lane]', 0)} B/lane of scratch)", flush=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as pool:
        list(pool.map(compile_one, todo))
    ver = subprocess.run([HIPCC, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    with open(os.path.join(OBJ, "toolchain.json"), "w") as f:          # which compiler the ISA guards were checked against
        json.dump({"hipcc": HIPCC, "version": ver.strip().splitlines()[:3]}, f, indent=0)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *[_obj(s) for s in sources()]]
    if verbose:
        print("[vipant_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


PROBES_DIR = os.path.join(HERE, "..", "tools", "probes")
PROBES_LIB = os.path.join(PROBES_DIR, "libvipant_probes.so")
PROBE_SOURCES = ("comm_shadow.hip",)


def build_probes(force: bool = False, verbose: bool = True) -> str:
    """tools/probes/libvipant_probes.so: measurement probes used by tools/ and by the ticket-walk tests (a kernel that holds CUs on a
    second stream).  Deliberately a separate library: nothing of it is linked into, or reachable from, libvipant_hip.so."""
    srcs = [os.path.join(PROBES_DIR, n) for n in PROBE_SOURCES]
    if force or _older(PROBES_LIB, srcs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", PROBES_LIB, *srcs]
        if verbose:
            print("[vipant_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return PROBES_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_probes(force="--force" in sys.argv)
    print(LIB)
