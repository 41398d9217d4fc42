import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd import ops
dev="cuda:0"; M=512*316; D=768
x=torch.randn(M,D,device=dev); y=torch.randn(M,D,device=dev).to(torch.bfloat16); g=torch.ones(D,device=dev); b=torch.zeros(D,device=dev)
def t(fn,n=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
print("ln_fwd + add:", round(t(lambda: ops.layernorm_fwd(x,g,b,add=y,want_sum=True)),1), "us  (1.49 GB)")
print("ln_fwd plain:", round(t(lambda: ops.layernorm_fwd(x,g,b)),1), "us  (0.75 GB)")
