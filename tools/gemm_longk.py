import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops
dev = "cuda:0"
def rb(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for M, N, K in ((43520, 768, 12288), (43520, 768, 3072), (43520, 768, 768), (65536, 1024, 4096)):
    a, b, c = rb(M, K), rb(N, K, scale=0.02), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.gemm_nt(a, b, c, epi=ops.EPI_BF16))
    print(f"variant={os.environ.get('VIPANT_GEMM_VARIANT','0'):>2} M={M} N={N} K={K}: {t*1e3:8.1f} us {2*M*N*K/t/1e9:7.1f} TF/s")
