"""Would the weight-gradient (TN) launches of a block's backward gain from a stream of their own?  One block's backward at the VA-step
shape as its kernel sequence (NT chain, attention backward, LayerNorm backwards; TN launches), six blocks back to back: all on one
stream in the step's order, against the TN launches on a second stream that waits for each operand's producer.  Upper bound of what
restructuring the block operators could buy (profiles/r5 notes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
b, S, H, D = 512, 316, 12, 768
M = b * S


def rb(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)


dy, g, h2, h1, o = rb(M, D), rb(M, 4 * D), rb(M, D), rb(M, D), rb(M, D)
w_pr_t, w_fc_t, w_o_t, w_qkv_t = rb(4 * D, D, scale=0.02), rb(D, 4 * D, scale=0.02), rb(D, D, scale=0.03), rb(D, 3 * D, scale=0.02)
code = torch.randint(0, 256, (M, 4 * D), dtype=torch.uint8, device=dev)
du, dh, dx, da = torch.empty(M, 4 * D, dtype=torch.bfloat16, device=dev), rb(M, D), rb(M, D), torch.empty(M, D, dtype=torch.bfloat16, device=dev)
qkv = rb(M, 3 * D)
att, lse = ops.mha_fwd(qkv, b, S, H, False)
x16 = torch.randn(M, D, device=dev).to(torch.float16)
mean, rstd, gamma = torch.zeros(M, device=dev), torch.ones(M, device=dev), torch.ones(D, device=dev)
dgm, dbt, dcs = (torch.empty(D, device=dev) for _ in range(3))
dW_pr, dW_fc, dW_o, dW_qkv = (torch.empty(s, device=dev) for s in ((D, 4 * D), (4 * D, D), (D, D), (3 * D, D)))
db_fc, db_qkv = torch.empty(4 * D, device=dev), torch.empty(3 * D, device=dev)
side = torch.cuda.Stream()


def block(split):
    main = torch.cuda.current_stream()
    ev = []

    def tn(fn):
        if not split:
            fn(); return
        e = torch.cuda.Event(); e.record(main)
        with torch.cuda.stream(side):
            side.wait_event(e)
            fn()

    ops.gemm_nt(dy, w_pr_t, du, aux=code, epi=ops.EPI_DQUICKGELU_D8)                  # du
    ops.gemm_nt(du, w_fc_t, dh, epi=ops.EPI_BF16)                                      # dh2
    tn(lambda: ops.gemm_tn(dy, g, dW_pr, ws_name="tn_side" if split else "gemm_tn"))
    tn(lambda: ops.gemm_tn(du, h2, dW_fc, a_colsum=db_fc, ws_name="tn_side" if split else "gemm_tn"))
    ops.layernorm_bwd(dh, x16, mean, rstd, gamma, dres=dx, dx_bf16=dx, dgamma=dgm, dbeta=dbt, dx_colsum=dcs)
    ops.gemm_nt(dx, w_o_t, da, epi=ops.EPI_BF16)                                       # d(attention output)
    tn(lambda: ops.gemm_tn(dx, o, dW_o, ws_name="tn_side" if split else "gemm_tn"))
    dqkv = ops.mha_bwd(qkv, att, da, lse, b, S, H, False)
    ops.gemm_nt(dqkv, w_qkv_t, dh, epi=ops.EPI_BF16)                                   # dh1
    tn(lambda: ops.gemm_tn(dqkv, h1, dW_qkv, a_colsum=db_qkv, ws_name="tn_side" if split else "gemm_tn"))
    if split:
        dqkv.record_stream(side)
    ops.layernorm_bwd(dh, x16, mean, rstd, gamma, dres=dx, dx_bf16=dx, dgamma=dgm, dbeta=dbt, dx_colsum=dcs)


def run(split, nblk=6):
    for _ in range(nblk):
        block(split)
    if split:
        torch.cuda.current_stream().wait_stream(side)


for split in (False, True):
    run(split); run(split)
torch.cuda.synchronize()
res = {False: [], True: []}
for _ in range(7):
    for split in (False, True):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(split); e1.record(); torch.cuda.synchronize()
        res[split].append(e0.elapsed_time(e1) / 6)
for split in (False, True):
    v = sorted(res[split])
    print(("TN launches on a second stream" if split else "one stream                   "), f"{v[len(v) // 2]:.3f} ms per block backward (best {v[0]:.3f})")
