"""What the vendor library reaches on the step's contraction shapes (M = 512 * 316 tokens), as a yardstick beside
tools/gemm_bench.py: plain bf16 GEMMs through torch (hipBLASLt / rocBLAS), no epilogue, HIP-event timing.
Measurement only -- nothing in the product path calls a library GEMM.  Usage: python tools/lib_gemm_ref.py [rounds]"""
import sys

import torch

dev = "cuda:0"
M = 512 * 316
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def rb(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


for name, N, K in (("nt qkv  ", 2304, 768), ("nt out  ", 768, 768), ("nt fc   ", 3072, 768), ("nt proj ", 768, 3072), ("nt dh1  ", 768, 2304)):
    x, w = rb(M, K), rb(N, K)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    med, best = timeit(lambda: torch.mm(x, w.t(), out=out))
    print(f"lib {name} N={N:5d} K={K:5d}: median {med * 1e3:8.1f} us  best {best * 1e3:8.1f} us  {2 * M * N * K / med / 1e9:7.1f} TFLOP/s")
    wt = w.t().contiguous()          # NN layout (weights pre-transposed), in case the library prefers it
    med, best = timeit(lambda: torch.mm(x, wt, out=out))
    print(f"lib {name} (NN)            : median {med * 1e3:8.1f} us  best {best * 1e3:8.1f} us  {2 * M * N * K / med / 1e9:7.1f} TFLOP/s")
for name, P, Q in (("tn dWo   ", 768, 768), ("tn dWfc  ", 3072, 768), ("tn dWproj", 768, 3072), ("tn dWqkv ", 2304, 768)):
    a, b = rb(M, P), rb(M, Q)
    out = torch.empty(P, Q, dtype=torch.bfloat16, device=dev)
    med, best = timeit(lambda: torch.mm(a.t(), b, out=out))
    print(f"lib {name} P={P:5d} Q={Q:5d}: median {med * 1e3:8.1f} us  best {best * 1e3:8.1f} us  {2 * M * P * Q / med / 1e9:7.1f} TFLOP/s")
