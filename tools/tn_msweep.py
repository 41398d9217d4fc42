"""bf16 / e4m3 weight-gradient kernels: time per K-tile and workgroup as a function of the token count (is the loop bound by where its
operands come from -- L2 / the 256 MB cache / HBM -- or by itself?).  python tools/tn_msweep.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vipant_amd import _ffi, ops
_ffi.call("vipant_device_check")
DEV = "cuda:0"
P, Q = 768, 3072


def timed(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for M in (8192, 16384, 32768, 65536, 161792, 323584):
    a = torch.randn(M, P, device=DEV).to(torch.bfloat16); b = torch.randn(M, Q, device=DEV).to(torch.bfloat16)
    c = torch.empty(P, Q, device=DEV)
    t = timed(lambda: ops.gemm_tn(a, b, c))
    tiles = 3 * 12; splits = min(256 // tiles, (M + 63) // 64); per = -(-((M + 63) // 64) // splits)
    qa, sa = ops.quant_e4m3_mx32(a); qb, sb = ops.quant_e4m3_mx32(b)
    t8 = timed(lambda: ops.gemm_tn_e4m3(qa, sa, qb, sb, c))
    per8 = -(-((M + 127) // 128) // splits)
    print(f"M={M:7d} operands {M * (P + Q) * 2 / 1e6:7.0f} MB: bf16 {t:7.1f} us, {per} K-tiles per workgroup -> {t / per:5.2f} us per K-tile ({2.0 * M * P * Q / t / 1e6:6.0f} TFLOP/s);  "
          f"e4m3 {t8:7.1f} us, {per8} K-tiles -> {t8 / per8:5.2f} us per K-tile", flush=True)
