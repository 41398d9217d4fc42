"""Micro-benchmark of the contraction kernels on the shapes of the VA step (b=512, S=316): interleaved rounds in one
process, HIP-event timing, random data.  Usage: python tools/gemm_bench.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
M = 512 * 316
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def timeit(fn, n=rounds):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def rb(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)


cases = []
x768, x3072, x2304 = rb(M, 768), rb(M, 3072), rb(M, 2304)
w_qkv, w_o, w_fc, w_pr = rb(2304, 768, scale=0.03), rb(768, 768, scale=0.03), rb(3072, 768, scale=0.03), rb(768, 3072, scale=0.02)
bias3072, bias768, bias2304 = torch.randn(3072, device=dev), torch.randn(768, device=dev), torch.randn(2304, device=dev)
o2304 = torch.empty(M, 2304, dtype=torch.bfloat16, device=dev)
o768 = torch.empty(M, 768, dtype=torch.bfloat16, device=dev)
o3072, u3072 = torch.empty(M, 3072, dtype=torch.bfloat16, device=dev), torch.empty(M, 3072, dtype=torch.bfloat16, device=dev)
r768, r768b = torch.randn(M, 768, device=dev), torch.empty(M, 768, device=dev)
g768, g3072a, g3072b, g2304 = (torch.empty(768, 768, device=dev), torch.empty(3072, 768, device=dev),
                               torch.empty(768, 3072, device=dev), torch.empty(2304, 768, device=dev))

cases.append(("nt qkv  bf16   N=2304 K=768 ", 2 * M * 2304 * 768, lambda: ops.gemm_nt(x768, w_qkv, o2304, bias=bias2304, epi=ops.EPI_BF16)))
cases.append(("nt out  res32  N=768  K=768 ", 2 * M * 768 * 768, lambda: ops.gemm_nt(x768, w_o, r768b, bias=bias768, aux=r768, epi=ops.EPI_RESIDUAL_F32)))
cases.append(("nt fc   gelu   N=3072 K=768 ", 2 * M * 3072 * 768, lambda: ops.gemm_nt(x768, w_fc, o3072, bias=bias3072, aux=u3072, epi=ops.EPI_QUICKGELU)))
cases.append(("nt proj res32  N=768  K=3072", 2 * M * 768 * 3072, lambda: ops.gemm_nt(x3072, w_pr, r768b, bias=bias768, aux=r768, epi=ops.EPI_RESIDUAL_F32)))
cases.append(("nt dgelu       N=3072 K=768 ", 2 * M * 3072 * 768, lambda: ops.gemm_nt(x768, w_fc, o3072, aux=u3072, epi=ops.EPI_DQUICKGELU)))
c3072 = torch.empty(M, 3072, dtype=torch.uint8, device=dev)
cases.append(("nt fc   gelu8  N=3072 K=768 ", 2 * M * 3072 * 768, lambda: ops.gemm_nt(x768, w_fc, o3072, bias=bias3072, aux=c3072, epi=ops.EPI_QUICKGELU_D8)))
cases.append(("nt dgelu8      N=3072 K=768 ", 2 * M * 3072 * 768, lambda: ops.gemm_nt(x768, w_fc, o3072, aux=c3072, epi=ops.EPI_DQUICKGELU_D8)))
cases.append(("nt dh2  bf16   N=768  K=3072", 2 * M * 768 * 3072, lambda: ops.gemm_nt(x3072, w_pr, o768, epi=ops.EPI_BF16)))
cases.append(("nt dh1  bf16   N=768  K=2304", 2 * M * 768 * 2304, lambda: ops.gemm_nt(x2304, rb(768, 2304, scale=0.02), o768, epi=ops.EPI_BF16)))
w_o2 = rb(768, 768, scale=0.03)
cases.append(("nt out  bf16   N=768  K=768 ", 2 * M * 768 * 768, lambda: ops.gemm_nt(x768, w_o2, o768, bias=bias768, epi=ops.EPI_BF16)))
cases.append(("tn dWo         P=768  Q=768 ", 2 * M * 768 * 768, lambda: ops.gemm_tn(x768, o768, g768)))
cases.append(("tn dWfc        P=3072 Q=768 ", 2 * M * 3072 * 768, lambda: ops.gemm_tn(x3072, x768, g3072a)))
cases.append(("tn dWproj      P=768  Q=3072", 2 * M * 768 * 3072, lambda: ops.gemm_tn(x768, x3072, g3072b)))
cases.append(("tn dWqkv       P=2304 Q=768 ", 2 * M * 2304 * 768, lambda: ops.gemm_tn(x2304, x768, g2304)))

print(f"variant={os.environ.get('VIPANT_GEMM_VARIANT', '0')}")
for name, flops, fn in cases:
    med, best = timeit(fn)
    print(f"{name}: median {med * 1e3:8.1f} us  best {best * 1e3:8.1f} us  {flops / med / 1e9:7.1f} TFLOP/s")

# prologue + epilogue share: same launch with K = 64 (one K-tile) against K = 768
for name, K, fn in (("nt fc gelu K=64 ", 64, lambda: ops.gemm_nt(x768[:, :64], w_fc[:, :64], o3072, bias=bias3072, aux=u3072, epi=ops.EPI_QUICKGELU)),
                    ("nt qkv bf16 K=64", 64, lambda: ops.gemm_nt(x768[:, :64], w_qkv[:, :64], o2304, bias=bias2304, epi=ops.EPI_BF16)),
                    ("nt out res K=64 ", 64, lambda: ops.gemm_nt(x768[:, :64], w_o[:, :64], r768b, bias=bias768, aux=r768, epi=ops.EPI_RESIDUAL_F32))):
    med, best = timeit(fn)
    print(f"{name}: median {med * 1e3:8.1f} us  best {best * 1e3:8.1f} us")
