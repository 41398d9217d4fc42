"""The bf16 weight-gradient contraction against the vendor library on the same arrays (torch.matmul -> hipBLASLt / rocBLAS; bf16 output
there, fp32 here), interleaved in one process: is there a tile to learn from?  python tools/tn_vs_library.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vipant_amd import _ffi, ops
_ffi.call("vipant_device_check")
DEV = "cuda:0"


def timed(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


M = 161792
for P, Q in ((768, 3072), (3072, 768), (2304, 768), (768, 768)):
    a = torch.randn(M, P, device=DEV).to(torch.bfloat16); b = torch.randn(M, Q, device=DEV).to(torch.bfloat16)
    c = torch.empty(P, Q, device=DEV)
    at = a.t()
    res = []
    for r in range(3):
        res.append((timed(lambda: ops.gemm_tn(a, b, c)), timed(lambda: torch.matmul(at, b))))
    ours, lib = min(x[0] for x in res), min(x[1] for x in res)
    fl = 2.0 * M * P * Q
    print(f"[{M}, {P}]^T x [{M}, {Q}]: ours {ours:7.1f} us ({fl / ours / 1e6:5.0f} TFLOP/s)   library {lib:7.1f} us ({fl / lib / 1e6:5.0f} TFLOP/s)", flush=True)
