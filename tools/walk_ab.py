"""Static-stride vs ticket walk of the persistent kernels (NT contractions, attention backward; VIPANT_GEMM_VARIANT bit 22), per launch shape of the VA step,
alternating in one process: python tools/walk_ab.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
M = 512 * 316
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 15


def rb(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)


x768, x3072, x2304 = rb(M, 768), rb(M, 3072), rb(M, 2304)
w_qkv, w_o, w_fc, w_pr, w_q2 = rb(2304, 768, scale=0.03), rb(768, 768, scale=0.03), rb(3072, 768, scale=0.03), rb(768, 3072, scale=0.02), rb(768, 2304, scale=0.02)
b3072, b768, b2304 = torch.randn(3072, device=dev), torch.randn(768, device=dev), torch.randn(2304, device=dev)
o2304 = torch.empty(M, 2304, dtype=torch.bfloat16, device=dev)
o768 = torch.empty(M, 768, dtype=torch.bfloat16, device=dev)
o3072 = torch.empty(M, 3072, dtype=torch.bfloat16, device=dev)
c3072 = torch.empty(M, 3072, dtype=torch.uint8, device=dev)
cases = [
    ("qkv    N=2304 K=768 ", lambda: ops.gemm_nt(x768, w_qkv, o2304, bias=b2304, epi=ops.EPI_BF16)),
    ("c_fc   N=3072 K=768 ", lambda: ops.gemm_nt(x768, w_fc, o3072, bias=b3072, aux=c3072, epi=ops.EPI_QUICKGELU_D8)),
    ("dgelu8 N=3072 K=768 ", lambda: ops.gemm_nt(x768, w_fc, o3072, aux=c3072, epi=ops.EPI_DQUICKGELU_D8)),
    ("dh2    N=768  K=3072", lambda: ops.gemm_nt(x3072, w_pr, o768, epi=ops.EPI_BF16)),
    ("dh1    N=768  K=2304", lambda: ops.gemm_nt(x2304, w_q2, o768, epi=ops.EPI_BF16)),
    ("out    N=768  K=768 ", lambda: ops.gemm_nt(x768, w_o, o768, bias=b768, epi=ops.EPI_BF16)),
]
# the persistent attention backward (6144 problems on 256 workgroups), same switch
qkv = rb(M, 2304)
att, lse = ops.mha_fwd(qkv, 512, 316, 12, False)
datt = rb(M, 768)
cases.append(("mha_bwd b=512 S=316 ", lambda: ops.mha_bwd(qkv, att, datt, lse, 512, 316, 12, False)))


def one(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for name, fn in cases:
    t = {"0": [], "4194304": []}
    for v in t:
        os.environ["VIPANT_GEMM_VARIANT"] = v
        fn(); fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for v in t:
            os.environ["VIPANT_GEMM_VARIANT"] = v
            t[v].append(one(fn))
    med = {v: sorted(x)[len(x) // 2] for v, x in t.items()}
    print(f"{name}: ticket {med['0']:7.1f} us (best {min(t['0']):7.1f})   static {med['4194304']:7.1f} us (best {min(t['4194304']):7.1f})   "
          f"delta {med['0'] - med['4194304']:+6.1f} us", flush=True)
