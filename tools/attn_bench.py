"""Attention kernels alone at the VA-step shape (b=512, S=316, H=12) and at BASELINE configs[4]'s (b=1024, H=16), with and without the
e4m3 emission of their outputs (q8): python tools/attn_bench.py [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def timeit(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for b, S, H in ((512, 316, 12), (1024, 316, 16)):
    D = H * 64
    M = b * S
    qkv = torch.randn(M, 3 * D, device=dev).to(torch.bfloat16)
    dout = torch.randn(M, D, device=dev).to(torch.bfloat16)
    out, lse = ops.mha_fwd(qkv, b, S, H, False)
    q8o = (torch.empty(M, D, dtype=torch.uint8, device=dev), torch.empty(ops.query("vipant_mx_scale_bytes", M, D), dtype=torch.uint8, device=dev))
    q8g = (torch.empty(M, 3 * D, dtype=torch.uint8, device=dev), torch.empty(ops.query("vipant_mx_scale_bytes", M, 3 * D), dtype=torch.uint8, device=dev))
    fl = 4.0 * b * H * S * S * 64
    for rnd in range(2):
        t0 = timeit(lambda: ops.mha_fwd(qkv, b, S, H, False))
        t1 = timeit(lambda: ops.mha_fwd(qkv, b, S, H, False, q8=q8o))
        t2 = timeit(lambda: ops.quant_e4m3_mx(out, *q8o))
        print(f"b={b} H={H} fwd {t0:7.1f} us ({fl / t0 / 1e6:5.0f} TFLOP/s)   with q8 {t1:7.1f}   stand-alone pass {t2:6.1f}")
        t0 = timeit(lambda: ops.mha_bwd(qkv, out, dout, lse, b, S, H, False))
        t1 = timeit(lambda: ops.mha_bwd(qkv, out, dout, lse, b, S, H, False, q8=q8g))
        dqkv = ops.mha_bwd(qkv, out, dout, lse, b, S, H, False)
        t2 = timeit(lambda: ops.quant_e4m3_mx(dqkv, *q8g))
        print(f"b={b} H={H} bwd {t0:7.1f} us ({2.5 * fl / t0 / 1e6:5.0f} TFLOP/s)   with q8 (incl. the dQ pass) {t1:7.1f}   stand-alone pass {t2:6.1f}")
        del dqkv
