"""TN contraction with and without the fused column sums (bias gradient) at the dW_fc / dW_qkv shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops
dev = "cuda:0"; M = 512 * 316
rb = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
def timeit(fn, n=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2] * 1e3
for P, Q in ((3072, 768), (2304, 768), (768, 768), (768, 3072)):
    a, b = rb(M, P), rb(M, Q)
    c = torch.empty(P, Q, device=dev); cs = torch.empty(P, device=dev)
    t0 = timeit(lambda: ops.gemm_tn(a, b, c)); t1 = timeit(lambda: ops.gemm_tn(a, b, c, a_colsum=cs))
    print(f"P={P} Q={Q}: plain {t0:7.1f} us   with column sums {t1:7.1f} us  (+{(t1 / t0 - 1) * 100:.1f} %)")
