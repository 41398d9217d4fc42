import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd import ops
dev="cuda:0"
cap=int(os.environ.get("VIPANT_GEMM_GRID","256"))
M=512*316*cap//256//256*256
def t(fn,n=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for K in (128, 768):
    x=(torch.randn(M,K,device=dev)).to(torch.bfloat16); w=(torch.randn(3072,K,device=dev)*0.03).to(torch.bfloat16)
    o=torch.empty(M,3072,dtype=torch.bfloat16,device=dev); c=torch.empty(M,3072,dtype=torch.uint8,device=dev); b=torch.randn(3072,device=dev)
    us=t(lambda: ops.gemm_nt(x,w,o,bias=b,aux=c,epi=ops.EPI_QUICKGELU_D8))
    us2=t(lambda: ops.gemm_nt(x,w,o,bias=b,epi=ops.EPI_BF16))
    print(f"grid {cap} M {M} K {K}: gelu8 {us:.1f} us -> {M*3072*3/us/1e3/cap:.1f} GB/s per WG written; bf16 {us2:.1f} us -> {M*3072*2/us2/1e3/cap:.1f} GB/s per WG")
