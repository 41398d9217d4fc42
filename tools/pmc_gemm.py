import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd import ops
dev="cuda:0"; M=512*316
x=(torch.randn(M,768,device=dev)).to(torch.bfloat16); w=(torch.randn(2304,768,device=dev)*0.03).to(torch.bfloat16)
o=torch.empty(M,2304,dtype=torch.bfloat16,device=dev)
x2=(torch.randn(M,3072,device=dev)).to(torch.bfloat16); w2=(torch.randn(768,3072,device=dev)*0.02).to(torch.bfloat16); o2=torch.empty(M,768,dtype=torch.bfloat16,device=dev)
for _ in range(3):
    ops.gemm_nt(x,w,o,epi=ops.EPI_BF16)      # qkv shape: K=768 N=2304
    ops.gemm_nt(x2,w2,o2,epi=ops.EPI_BF16)   # dh2 shape: K=3072 N=768
torch.cuda.synchronize()
