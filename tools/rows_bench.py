"""Micro-benchmark of the one-query-per-(item, head) attention kernels of the last block (csrc/readout_rows.hip) at the step's
shape (b = 512, S = 316, H = 12): HIP-event timing, medians.  Usage: python tools/rows_bench.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd import ops  # noqa: E402

dev = "cuda:0"
b, S, H = 512, 316, 12
D = H * 64
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 15
qkv = (torch.randn(b * S, 3 * D, device=dev) * 1.5).to(torch.bfloat16)
q = qkv[::S, :D].contiguous()
out = torch.empty(b, D, dtype=torch.bfloat16, device=dev)
probs = torch.empty(b, H, S, dtype=torch.float32, device=dev)
do = torch.randn(b, D, device=dev).to(torch.bfloat16)
dq = torch.empty_like(out)
dqkv = torch.empty_like(qkv)
st = torch.cuda.current_stream().cuda_stream


def fwd():
    ops.call("vipant_mha_rows_fwd", q.data_ptr(), qkv.data_ptr(), None, out.data_ptr(), probs.data_ptr(), b, S, H, 0, st)


def bwd():
    ops.call("vipant_mha_rows_bwd", q.data_ptr(), qkv.data_ptr(), None, probs.data_ptr(), do.data_ptr(), dq.data_ptr(), dqkv.data_ptr(),
             b, S, H, 0, st)


for name, fn, nbytes in (("fwd", fwd, 2 * b * S * D * 2), ("bwd", bwd, 4 * b * S * D * 2)):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"mha_rows_{name}: median {med * 1e3:.1f} us  best {ts[0] * 1e3:.1f} us  {nbytes / med / 1e6:.0f} GB/s of K|V (+dK|dV) bytes")

# the same attention with the K / V projection folded into the query side (csrc/readout_ctx.hip): one pass over h1
h1 = torch.randn(b * S, D, device=dev).to(torch.bfloat16)
PAIR = int(os.environ.get("ROWS_PAIR", "1"))     # 1: qk / contexts / dqk as bf16 pairs (the step's form, round 5); 0: single planes
qk = (torch.randn(2, b * H, D, device=dev) * torch.tensor([0.35, 0.35 / 256], device=dev).view(2, 1, 1)).to(torch.bfloat16)
hctx = torch.empty(2, b * H, D, dtype=torch.bfloat16, device=dev)
dctx = torch.randn(b * H, D, device=dev).to(torch.bfloat16)
dh1 = torch.empty_like(h1)
dqk = torch.empty_like(qk)


def cfwd():
    ops.call("vipant_rows_ctx_fwd", qk.data_ptr(), h1.data_ptr(), None, hctx.data_ptr(), probs.data_ptr(), b, S, H, 0, PAIR, st)


def cbwd():
    ops.call("vipant_rows_ctx_bwd", qk.data_ptr(), dctx.data_ptr(), hctx.data_ptr(), h1.data_ptr(), None, probs.data_ptr(), dh1.data_ptr(),
             dqk.data_ptr(), None, b, S, H, 0, PAIR, st)


for name, fn, nbytes in (("fwd", cfwd, b * S * D * 2), ("bwd", cbwd, 2 * b * S * D * 2)):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"rows_ctx_{name}: median {med * 1e3:.1f} us  best {ts[0] * 1e3:.1f} us  {nbytes / med / 1e6:.0f} GB/s of h1 (+dh1) bytes")
