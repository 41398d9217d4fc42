"""Schedule choice per launch shape of the step, re-measured on the final build: every NT launch shape of the audio tower (M = 161 792)
and of the image tower (M = 25 600) under the schedules the library can be switched to (VIPANT_GEMM_VARIANT, read per call), alternating in
one process; results are checked bit for bit against the default.  python tools/schedule_sweep.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 11


def rb(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)


def one(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


PLAIN = {"default": 0, "DEEP": 131072, "k-step": 262144}
GELU = {"default (DEEP grouped)": 0, "round-4 choice": 8388608, "k-step grouped": 2048, "k-step plain": 4096 | 262144}
for M, tag in ((512 * 316, "audio"), (512 * 50, "image")):
    for N, K, name in ((2304, 768, "qkv"), (768, 768, "out_proj"), (768, 3072, "c_proj / dh2"), (768, 2304, "dh1"), (3072, 768, "c_fc + QuickGELU"),
                       (3072, 768, "QuickGELU'")):
        if tag == "image" and name in ("dh1", "QuickGELU'"):
            continue
        x = rb(M, K); w = rb(N, K, scale=0.03); c = torch.empty(M, N, dtype=torch.bfloat16, device=dev); bias = torch.randn(N, device=dev)
        if name == "c_fc + QuickGELU":
            aux = torch.empty(M, N, dtype=torch.uint8, device=dev)
            fn = lambda: ops.gemm_nt(x, w, c, bias=bias, aux=aux, epi=ops.EPI_QUICKGELU_D8)
            variants = GELU
        elif name == "QuickGELU'":
            aux = torch.randint(0, 256, (M, N), dtype=torch.uint8, device=dev)
            fn = lambda: ops.gemm_nt(x, w, c, aux=aux, epi=ops.EPI_DQUICKGELU_D8)
            variants = GELU
        else:
            fn = lambda: ops.gemm_nt(x, w, c, bias=bias if name in ("qkv", "out_proj", "c_proj / dh2") else None, epi=ops.EPI_BF16)
            variants = PLAIN
        t = {k: [] for k in variants}
        ref = None
        for k, v in variants.items():
            os.environ["VIPANT_GEMM_VARIANT"] = str(v)
            fn(); fn()
            got = c.clone()
            if ref is None:
                ref = got
            else:
                assert torch.equal(ref, got), (tag, name, k)
        for _ in range(rounds):
            for k, v in variants.items():
                os.environ["VIPANT_GEMM_VARIANT"] = str(v)
                t[k].append(one(fn))
        print(f"{tag} {name:18s} N={N:5d} K={K:5d}: " + "   ".join(f"{k} {sorted(a)[len(a) // 2]:7.1f}" for k, a in t.items()), flush=True)
        del x, w, c
