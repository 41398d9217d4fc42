"""Weight-gradient contraction, bf16 (vipant_gemm_tn) against e4m3 (vipant_gemm_tn_e4m3), at the shapes of BASELINE.json configs[4]'s
tower (audio ViT-L: D = 1024, 1024 clips x 316 tokens) and of the headline tower (D = 768, 512 clips), plus the two quantiser passes.
    python tools/tn8_bench.py [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vipant_amd import _ffi, ops  # noqa: E402

_ffi.call("vipant_device_check")
DEV = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timed(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for M, shapes in ((1024 * 316, [(1024, 4096), (4096, 1024), (1024, 1024), (3072, 1024)]), (512 * 316, [(768, 3072), (3072, 768), (768, 768), (2304, 768)])):
    widest = max(max(s) for s in shapes)
    src = torch.randn(M, widest, device=DEV).to(torch.bfloat16)
    for P, Q in shapes:
        a, b = src[:, :P].contiguous(), src[:, :Q].flip(0).contiguous()
        c = torch.empty(P, Q, dtype=torch.float32, device=DEV)
        t_bf = timed(lambda: ops.gemm_tn(a, b, c))
        qa, sa = ops.quant_e4m3_mx32(a)
        qb, sb = ops.quant_e4m3_mx32(b)
        t_q = timed(lambda: ops.quant_e4m3_mx32(b, qb, sb))
        qr, sr = ops.quant_e4m3_mx(b)
        t_u = timed(lambda: ops.mx_uniform32(qr, sr))          # after the first call every row carries its block's scale: read-only cost
        t_8 = timed(lambda: ops.gemm_tn_e4m3(qa, sa, qb, sb, c))
        csum = torch.empty(P, device=DEV)
        t_8c = timed(lambda: ops.gemm_tn_e4m3(qa, sa, qb, sb, c, a_colsum=csum))
        fl = 2.0 * M * P * Q
        print(f"M={M} P={P} Q={Q}: bf16 {t_bf * 1e3:8.1f} us ({fl / t_bf / 1e9:7.1f} TFLOP/s)   e4m3 {t_8 * 1e3:8.1f} us ({fl / t_8 / 1e9:7.1f} TFLOP/s; with column sums {t_8c * 1e3:8.1f} us)   "
              f"quant_mx32 of [M, {Q}] {t_q * 1e3:7.1f} us ({3.0 * M * Q / t_q / 1e6:6.0f} GB/s)   uniform32 (no rewrite) {t_u * 1e3:7.1f} us", flush=True)
        del a, b, c, qa, qb, qr
