"""e4m3 contraction vs the bf16 one at the shapes of the ViT-L block (configs[4]) and of the ViT-B block: HIP events, random data."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd import ops
dev = "cuda:0"


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for name, M, N, K in (("ViT-L qkv", 512 * 316, 3072, 1024), ("ViT-L c_fc", 512 * 316, 4096, 1024), ("ViT-L c_proj", 512 * 316, 1024, 4096),
                      ("ViT-B qkv", 512 * 316, 2304, 768), ("ViT-B c_proj", 512 * 316, 768, 3072)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev); bias = torch.randn(N, device=dev)
    qa, sa = ops.quant_e4m3_mx(a); qw, sw = ops.quant_e4m3(w)
    t16 = t(lambda: ops.gemm_nt(a, w, c, bias=bias))
    t8 = t(lambda: ops.gemm_nt_e4m3(qa, sa, qw, sw, c, bias=bias))
    tq = t(lambda: ops.quant_e4m3_mx(a, qa, sa))
    fl = 2.0 * M * N * K
    print("%-13s M=%d N=%d K=%d: bf16 %7.1f us (%.2f PF/s)  e4m3 %7.1f us (%.2f PF/s)  quantise A %6.1f us (%.2f TB/s)" %
          (name, M, N, K, t16, fl / t16 / 1e9, t8, fl / t8 / 1e9, tq, 3.0 * M * K / tq / 1e6))
