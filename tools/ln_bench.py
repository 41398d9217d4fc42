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
dy = torch.randn(M, D, device=dev).to(torch.bfloat16)
_, _, mean, rstd = ops.layernorm_fwd(x, g, b)
dxb = torch.randn(M, D, device=dev).to(torch.bfloat16)
dg, db, cs = torch.empty(D, device=dev), torch.empty(D, device=dev), torch.empty(D, device=dev)
ws = ops.scratch("ln_bwd", ops.query("vipant_layernorm_bwd_workspace_bytes", M, D), x.device)
st = torch.cuda.current_stream().cuda_stream
def bwd_bf16():
    ops.call("vipant_layernorm_bwd", dy.data_ptr(), 2, x.data_ptr(), D, mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), dxb.data_ptr(), None, D,
             dxb.data_ptr(), dg.data_ptr(), db.data_ptr(), cs.data_ptr(), 0, M, D, ws.data_ptr(), ws.numel(), st)
print("ln_bwd bf16 stream:", round(t(bwd_bf16), 1), "us  (1.24 GB)")
x16 = x.to(torch.float16)
_, _, mean16, rstd16 = ops.layernorm_fwd(x16, g, b)
def bwd_f16():
    ops.call("vipant_layernorm_bwd", dy.data_ptr(), 2 | 4, x16.data_ptr(), D, mean16.data_ptr(), rstd16.data_ptr(), g.data_ptr(), dxb.data_ptr(), None, D,
             dxb.data_ptr(), dg.data_ptr(), db.data_ptr(), cs.data_ptr(), 0, M, D, ws.data_ptr(), ws.numel(), st)
print("ln_bwd bf16 gradient stream, fp16 rows:", round(t(bwd_f16), 1), "us  (0.99 GB)")
print("ln_fwd + add, fp16 stream:", round(t(lambda: ops.layernorm_fwd(x16,g,b,add=y,want_sum=True,sum_f16=True)),1), "us  (0.99 GB)")
