import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops
dev = "cuda:0"; M = 512 * 316
def rb(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
x3072, w_pr, o768 = rb(M, 3072), rb(768, 3072, scale=0.02), torch.empty(M, 768, dtype=torch.bfloat16, device=dev)
x768, w_qkv, o2304 = rb(M, 768), rb(2304, 768, scale=0.03), torch.empty(M, 2304, dtype=torch.bfloat16, device=dev)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
t = timeit(lambda: ops.gemm_nt(x3072, w_pr, o768, epi=ops.EPI_BF16))
print(f"variant={os.environ.get('VIPANT_GEMM_VARIANT','0'):>2} dh2 K=3072 N=768 : {t*1e3:8.1f} us {2*M*768*3072/t/1e9:7.1f} TF/s")
t = timeit(lambda: ops.gemm_nt(x768, w_qkv, o2304, epi=ops.EPI_BF16))
print(f"variant={os.environ.get('VIPANT_GEMM_VARIANT','0'):>2} qkv K=768 N=2304 : {t*1e3:8.1f} us {2*M*768*2304/t/1e9:7.1f} TF/s")
