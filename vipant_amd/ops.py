"""Tensor-level wrappers over the C ABI plus the autograd nodes of the contrastive step.

PyTorch is plumbing here: it owns device memory, streams and the autograd tape; every numeric
operation below is a call into libvipant_hip.so (hand-written gfx950 kernels).  Nothing in this file
computes on the CPU or through ATen kernels, except zero-fills / views used as allocation helpers.

Activation layout: token-major [M, D] with M = batch * tokens, row m = b * S + s (batch-first).
The residual stream and its gradient are fp32; every MFMA operand is bf16 (fp32 accumulate).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _ffi
from ._ffi import (EPI_BF16, EPI_DQUICKGELU, EPI_DQUICKGELU_D8, EPI_F32, EPI_QUICKGELU, EPI_QUICKGELU_D8,  # noqa: F401
                   EPI_RESIDUAL_F32, EPI_SCALE_F32,
                   call, query)

BF16, F32, F16, I64 = torch.bfloat16, torch.float32, torch.float16, torch.int64

# order of the 12 per-layer parameters handed to BackboneFn (reference names, cvap/module/val.py:496-507)
BLOCK_PARAM_NAMES = (
    "ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias",
    "attn.out_proj.weight", "attn.out_proj.bias", "ln_2.weight", "ln_2.bias",
    "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias",
)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _ffi.VipantError(f"{name}: expected a device tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise _ffi.VipantError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if t.stride(-1) != 1 and t.numel() > 1:
        raise _ffi.VipantError(f"{name}: last dimension must be contiguous")


def no_tape(params):
    """Parameters as the autograd nodes below should see them: under torch.no_grad() (Monitor.infer, cfg.eval) nothing
    will be differentiated, but `ctx.needs_input_grad` still reports `requires_grad` -- and `forward` itself always runs
    with grad mode off, so it cannot tell.  Detached parameters make the nodes take their forward-only path (one set of
    temporaries for all layers, cached bf16 weights, nothing saved)."""
    return params if torch.is_grad_enabled() else tuple(p.detach() for p in params)


_scratch: Dict[Tuple[str, int], torch.Tensor] = {}

# bench.py hooks a probe here to bracket selected kernel launches with HIP events on the launch stream
# (kernels are launched on torch's current stream, so torch.cuda.Event records on the right queue).
KERNEL_PROBE: Dict[str, dict] = {}


def scratch(name: str, nbytes: int, device) -> torch.Tensor:
    """Named, grow-only byte workspace (256-B aligned by the caching allocator)."""
    key = (name, device.index if device.index is not None else torch.cuda.current_device())
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _scratch[key] = buf
    return buf


# --------------------------------------------------------------------------------------- raw ops
def gemm_nt(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, *, bias=None, aux=None, epi: int = EPI_BF16,
            alpha: float = 1.0, few_rows: bool = False):
    """c[M,N] = a[M,K] @ b[N,K]^T with the epilogue `epi` (see include/vipant_hip.h).  `few_rows`: the rows are one per item of a
    batch (VIPANT_EPI_FEW_ROWS: small tiles, K split over a workgroup's waves)."""
    _need(a, BF16, "gemm_nt.a"); _need(b, BF16, "gemm_nt.b")
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and tuple(c.shape) == (M, N), (a.shape, b.shape, c.shape)
    if aux is not None:
        assert aux.stride(0) == c.stride(0)
    probe = KERNEL_PROBE.get("gemm_nt")
    if probe is not None and probe["match"](epi, M, N, K):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    else:
        probe = None
    call("vipant_gemm_nt", a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0),
         _ptr(bias), _ptr(aux), float(alpha), M, N, K, epi | (_ffi.EPI_FEW_ROWS if few_rows else 0), _stream())
    if probe is not None:
        e1.record()
        probe["events"].append((e0, e1))
    return c


def quant_e4m3(x: torch.Tensor, q: torch.Tensor = None, scale: torch.Tensor = None):
    """bf16 rows [M, K] -> (e4m3 bytes [M, K], E8M0 row-scale bytes [M]); see vipant_quant_e4m3_rows."""
    _need(x, BF16, "quant_e4m3.x")
    M, K = x.shape
    if q is None:
        q = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty((M,), dtype=torch.uint8, device=x.device)
    call("vipant_quant_e4m3_rows", x.data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), scale.data_ptr(), M, K, _stream())
    return q, scale


def quant_e4m3_mx(x: torch.Tensor, q: torch.Tensor = None, scale: torch.Tensor = None):
    """bf16 rows [M, K] -> (e4m3 bytes [M, K], E8M0 BLOCK scales, one per 32 elements of a row, in the library's MX layout:
    vipant_quant_e4m3_mx) -- the form an activation operand of gemm_nt_e4m3 takes."""
    _need(x, BF16, "quant_e4m3_mx.x")
    M, K = x.shape
    if q is None:
        q = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty((query("vipant_mx_scale_bytes", M, K),), dtype=torch.uint8, device=x.device)
    call("vipant_quant_e4m3_mx", x.data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), scale.data_ptr(), M, K, _stream())
    return q, scale


def quant_e4m3_mx32(x: torch.Tensor, q: torch.Tensor = None, scale: torch.Tensor = None):
    """bf16 rows [M, K] -> (e4m3 bytes, block scales in the MX layout) with every scale shared by an aligned block of 32 rows x 32
    columns (vipant_quant_e4m3_mx32): an operand of gemm_tn_e4m3, and as valid an activation operand of gemm_nt_e4m3 as any."""
    _need(x, BF16, "quant_e4m3_mx32.x")
    M, K = x.shape
    if q is None:
        q = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty((query("vipant_mx_scale_bytes", M, K),), dtype=torch.uint8, device=x.device)
    call("vipant_quant_e4m3_mx32", x.data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), scale.data_ptr(), M, K, _stream())
    return q, scale


def mx_uniform32(q: torch.Tensor, scale: torch.Tensor):
    """In place: e4m3 bytes [M, K] with row-wise block scales -> block-uniform scales (vipant_mx_uniform32)."""
    assert q.dtype == torch.uint8 and scale.dtype == torch.uint8 and q.dim() == 2
    M, K = q.shape
    assert scale.numel() >= query("vipant_mx_scale_bytes", M, K)
    call("vipant_mx_uniform32", q.data_ptr(), q.stride(0), scale.data_ptr(), M, K, _stream())
    return q, scale


def gemm_tn_e4m3(a, sa, b, sb, c: torch.Tensor, accumulate: bool = False, a_colsum=None, ws_name: str = "gemm_tn"):
    """c[P, Q] (fp32) (+)= dequant(a, sa)^T @ dequant(b, sb): a [M, P], b [M, Q] e4m3 bytes with block-uniform scales;
    a_colsum (fp32 [P], optional) (+)= column sums of dequant(a, sa)."""
    assert a.dtype == torch.uint8 and b.dtype == torch.uint8 and sa.dtype == torch.uint8 and sb.dtype == torch.uint8
    _need(c, F32, "gemm_tn_e4m3.c")
    M, P = a.shape
    Q = b.shape[1]
    assert b.shape[0] == M and tuple(c.shape) == (P, Q), (a.shape, b.shape, c.shape)
    ws = scratch(ws_name, query("vipant_gemm_tn_e4m3_workspace_bytes", M, P, Q), a.device)
    call("vipant_gemm_tn_e4m3", a.data_ptr(), a.stride(0), sa.data_ptr(), b.data_ptr(), b.stride(0), sb.data_ptr(), c.data_ptr(),
         c.stride(0), M, P, Q, int(accumulate), _ptr(a_colsum), ws.data_ptr(), ws.numel(), _stream())
    return c


def mx_scale_index(M: int, K: int, device) -> torch.Tensor:
    """int64 [M, K // 32]: where the MX layout keeps the scale byte of (row, 32-element block) -- for tests and tools."""
    m = torch.arange(M, device=device).view(M, 1)
    kb = torch.arange(K // 32, device=device).view(1, K // 32)
    return (((m >> 7) * (K // 128) + (kb >> 2)) * 16 + (m & 15)) * 32 + (kb & 3) * 8 + ((m >> 4) & 7)


def gemm_nt_e4m3(a, sa, b, sb, c: Optional[torch.Tensor], *, bias=None, aux=None, epi: int = EPI_BF16, emit=None):
    """c[M,N] (bf16) = dequant(a, sa) @ dequant(b, sb)^T with the epilogue `epi`: e4m3 operands; sa: a's block scales (quant_e4m3_mx),
    sb: b's row scales (quant_e4m3).  emit = (bytes [M, N], block scales): the epilogue also leaves the e4m3 form of its result there;
    c (and aux) may then be None (QuickGELU epilogue)."""
    assert a.dtype == torch.uint8 and b.dtype == torch.uint8 and sa.dtype == torch.uint8 and sb.dtype == torch.uint8
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and sa.numel() >= query("vipant_mx_scale_bytes", M, K) and sb.numel() == N, (a.shape, b.shape, sa.shape)
    assert c is None or tuple(c.shape) == (M, N)
    if aux is not None and c is not None:
        assert aux.stride(0) == c.stride(0)
    call("vipant_gemm_nt_e4m3", a.data_ptr(), a.stride(0), sa.data_ptr(), b.data_ptr(), b.stride(0), sb.data_ptr(), _ptr(c),
         c.stride(0) if c is not None else N, _ptr(bias), _ptr(aux), emit[0].data_ptr() if emit else None,
         emit[1].data_ptr() if emit else None, M, N, K, epi, _stream())
    return c


def gemm_tn(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, accumulate: bool = False, a_colsum=None,
            ws_name: str = "gemm_tn"):
    """c[P,Q] (+)= a[M,P]^T @ b[M,Q] (fp32 out); a_colsum (fp32 [P], optional) (+)= column sums of a.
    `ws_name`: the split-K workspace is shared by all calls of one name -- calls issued on different streams need their own."""
    _need(a, BF16, "gemm_tn.a"); _need(b, BF16, "gemm_tn.b"); _need(c, F32, "gemm_tn.c")
    M, P = a.shape
    Q = b.shape[1]
    assert b.shape[0] == M and tuple(c.shape) == (P, Q), (a.shape, b.shape, c.shape)
    nbytes = query("vipant_gemm_tn_workspace_bytes", M, P, Q)
    ws = scratch(ws_name, nbytes, a.device)
    call("vipant_gemm_tn", a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0), M, P, Q,
         int(accumulate), _ptr(a_colsum), ws.data_ptr(), ws.numel(), _stream())
    return c


def colsum(x: torch.Tensor, out: torch.Tensor, accumulate: bool = False):
    _need(x, BF16, "colsum.x"); _need(out, F32, "colsum.out")
    M, N = x.shape
    ws = scratch("colsum", query("vipant_colsum_workspace_bytes", M, N), x.device)
    call("vipant_colsum_bf16", x.data_ptr(), x.stride(0), out.data_ptr(), M, N, int(accumulate), ws.data_ptr(),
         ws.numel(), _stream())
    return out


def layernorm_fwd(x: torch.Tensor, gamma, beta, *, want_bf16=True, want_f32=False, rows: Optional[int] = None,
                  ldx: Optional[int] = None, add: Optional[torch.Tensor] = None, want_sum=False, sum_f16=False):
    """x fp32 | fp16 [M, D] (or `rows` rows with stride `ldx`) -> (y_bf16 | None, y_f32 | None, mean, rstd[, x + add]).
    `add` (bf16 [M, D]) fuses the block's residual add in front of the norm; want_sum returns the new stream (fp16 with sum_f16)."""
    _need(x, x.dtype if x.dtype == F16 else F32, "layernorm_fwd.x")
    D = x.shape[-1]
    M = rows if rows is not None else x.shape[0]
    ldx = ldx if ldx is not None else x.stride(0)
    y = torch.empty((M, D), dtype=BF16, device=x.device) if want_bf16 else None
    y32 = torch.empty((M, D), dtype=F32, device=x.device) if want_f32 else None
    mean = torch.empty((M,), dtype=F32, device=x.device)
    rstd = torch.empty((M,), dtype=F32, device=x.device)
    xsum = torch.empty((M, D), dtype=F16 if sum_f16 else F32, device=x.device) if (add is not None and want_sum) else None
    flags = (_ffi.STREAM_IN_F16 if x.dtype == F16 else 0) | (_ffi.STREAM_OUT_F16 if sum_f16 else 0)
    call("vipant_layernorm_fwd_e4m3", x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), _ptr(y), _ptr(y32),
         mean.data_ptr(), rstd.data_ptr(), M, D, _ptr(add), _ptr(xsum), None, None, flags, _stream())
    if add is not None:
        return y, y32, mean, rstd, xsum
    return y, y32, mean, rstd


def residual_add(x: torch.Tensor, add: torch.Tensor) -> torch.Tensor:
    """fp32 out = x (fp32 or fp16 stream) + add (bf16)."""
    out = torch.empty(x.shape, dtype=F32, device=x.device)
    call("vipant_residual_add", x.data_ptr(), add.data_ptr(), out.data_ptr(), x.numel(),
         _ffi.STREAM_IN_F16 if x.dtype == F16 else 0, _stream())
    return out


def gather_rows(x: torch.Tensor, idx: Optional[torch.Tensor], batch: int, S: int) -> torch.Tensor:
    """x [batch * S, D] (any 2- or 4-byte dtype) -> its rows i * S + idx[i] (idx None: the first row of every item), [batch, D]."""
    assert x.dim() == 2 and x.is_contiguous() and x.shape[0] == batch * S
    out = torch.empty((batch, x.shape[1]), dtype=x.dtype, device=x.device)
    call("vipant_gather_rows_bytes", x.data_ptr(), _ptr(idx), out.data_ptr(), batch, S, x.shape[1] * x.element_size(), _stream())
    return out


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, mean, rstd, gamma, *, dres=None, dx=None, lddx=None,
                  dx_bf16=None, dgamma, dbeta, dx_colsum=None, accumulate=False, rows=None, ldx=None):
    D = x.shape[-1]
    M = rows if rows is not None else x.shape[0]
    ldx = ldx if ldx is not None else x.stride(0)
    lddx = lddx if lddx is not None else (dx.stride(0) if dx is not None else D)
    ws = scratch("ln_bwd", query("vipant_layernorm_bwd_workspace_bytes", M, D), x.device)
    call("vipant_layernorm_bwd", dy.data_ptr(), int(dy.dtype == F32) | (_ffi.LN_X_F16 if x.dtype == F16 else 0), x.data_ptr(), ldx, mean.data_ptr(),
         rstd.data_ptr(), gamma.data_ptr(), _ptr(dres), _ptr(dx), lddx, _ptr(dx_bf16), dgamma.data_ptr(),
         dbeta.data_ptr(), _ptr(dx_colsum), int(accumulate), M, D, ws.data_ptr(), ws.numel(), _stream())


def mha_fwd(qkv: torch.Tensor, batch: int, S: int, H: int, causal: bool, q8=None):
    """q8 = (bytes, block scales): also leave the e4m3 form of `out` there (vipant_quant_e4m3_mx of it, rows of H * 64 bytes)."""
    _need(qkv, BF16, "mha_fwd.qkv")
    out = torch.empty((batch * S, H * 64), dtype=BF16, device=qkv.device)
    lse = torch.empty((batch, H, S), dtype=F32, device=qkv.device)
    if q8 is not None:
        call("vipant_mha_fwd_e4m3", qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), q8[0].data_ptr(), q8[1].data_ptr(), batch, S, H,
             int(causal), _stream())
    else:
        call("vipant_mha_fwd", qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), batch, S, H, int(causal), _stream())
    return out, lse


def mha_bwd(qkv, out, dout, lse, batch: int, S: int, H: int, causal: bool, q8=None):
    """q8 = (bytes, block scales): also leave the e4m3 form of `dqkv` there (rows of 3 * H * 64 bytes)."""
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    if q8 is not None:
        call("vipant_mha_bwd_e4m3", qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(),
             dqkv.data_ptr(), q8[0].data_ptr(), q8[1].data_ptr(), batch, S, H, int(causal), _stream())
    else:
        call("vipant_mha_bwd", qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(),
             dqkv.data_ptr(), batch, S, H, int(causal), _stream())
    return dqkv


def cast_bf16(w: torch.Tensor, transpose: bool = False):
    """fp32 matrix -> (bf16 copy, bf16 transposed copy | None)."""
    _need(w, F32, "cast_bf16.w")
    w2 = w.reshape(w.shape[0], -1) if w.dim() != 2 else w
    assert w2.is_contiguous()
    R, Cc = w2.shape
    dst = torch.empty((R, Cc), dtype=BF16, device=w.device)
    dst_t = torch.empty((Cc, R), dtype=BF16, device=w.device) if transpose else None
    call("vipant_cast_bf16", w2.data_ptr(), dst.data_ptr(), _ptr(dst_t), R, Cc, _stream())
    return dst, dst_t


def cast_bf16_flat(x: torch.Tensor) -> torch.Tensor:
    _need(x, F32, "cast_bf16_flat.x")
    assert x.is_contiguous() and x.numel() % 4 == 0
    dst = torch.empty(x.shape, dtype=BF16, device=x.device)
    call("vipant_cast_bf16", x.data_ptr(), dst.data_ptr(), None, 1, x.numel(), _stream())
    return dst


class _WeightCaster:
    """bf16 copies (W and W^T) of a list of fp32 weight matrices, refreshed by ONE launch (vipant_cast_bf16_multi).  The output
    buffers and the device pointer tables are built once per parameter set and reused every step."""

    def __init__(self, weights: Sequence[torch.Tensor]):
        dev = weights[0].device
        self.key = tuple(w.data_ptr() for w in weights)
        self.keep = [w.untyped_storage() for w in weights]       # the addresses stay ours while this entry lives
        shapes = [(w.shape[0], w.numel() // w.shape[0]) for w in weights]
        self.wb = [torch.empty(s, dtype=BF16, device=dev) for s in shapes]
        self.wt = [torch.empty((s[1], s[0]), dtype=BF16, device=dev) for s in shapes]
        starts = [0]
        for r, c in shapes:
            starts.append(starts[-1] + ((r + 63) // 64) * ((c + 63) // 64))
        self.total = starts[-1]
        self.src = torch.tensor([w.data_ptr() for w in weights], dtype=I64, device=dev)
        self.dst = torch.tensor([t.data_ptr() for t in self.wb], dtype=I64, device=dev)
        self.dst_t = torch.tensor([t.data_ptr() for t in self.wt], dtype=I64, device=dev)
        self.R = torch.tensor([s[0] for s in shapes], dtype=torch.int32, device=dev)
        self.C = torch.tensor([s[1] for s in shapes], dtype=torch.int32, device=dev)
        self.starts = torch.tensor(starts, dtype=torch.int32, device=dev)
        self.n = len(weights)

    def refresh(self):
        call("vipant_cast_bf16_multi", self.src.data_ptr(), self.dst.data_ptr(), self.dst_t.data_ptr(), self.R.data_ptr(),
             self.C.data_ptr(), self.starts.data_ptr(), self.n, self.total, _stream())
        return self.wb, self.wt

    def refresh_e4m3(self):
        """Row-quantised e4m3 copies of the bf16 W and W^T just refreshed: [(bytes, row scales)] per matrix, both orientations
        (the forward contractions read W rows, the input-gradient ones W^T rows)."""
        if not hasattr(self, "wq"):
            dev = self.wb[0].device
            self.wq = [(torch.empty(t.shape, dtype=torch.uint8, device=dev), torch.empty((t.shape[0],), dtype=torch.uint8, device=dev))
                       for t in self.wb]
            self.wtq = [(torch.empty(t.shape, dtype=torch.uint8, device=dev), torch.empty((t.shape[0],), dtype=torch.uint8, device=dev))
                        for t in self.wt]
        for src, (q, sc) in zip(self.wb + self.wt, self.wq + self.wtq):
            quant_e4m3(src, q, sc)
        return self.wq, self.wtq


_casters: Dict[tuple, _WeightCaster] = {}


def cast_weights(weights: Sequence[torch.Tensor], e4m3: bool = False):
    """(list of bf16 W, list of bf16 W^T) for the trainable matrices of a tower, one launch; with `e4m3` also their row-quantised
    e4m3 forms (lists of (bytes, row scales))."""
    key = tuple(w.data_ptr() for w in weights)
    c = _casters.get(key)
    if c is None:
        while len(_casters) >= 16:
            _casters.pop(next(iter(_casters)))
        c = _casters[key] = _WeightCaster([w.detach() for w in weights])
    wb, wt = c.refresh()
    if e4m3:
        return (wb, wt) + tuple(c.refresh_e4m3())
    return wb, wt


FP8_TN = os.environ.get("VIPANT_FP8_TN", "1") != "0"      # e4m3 towers: the weight-gradient contractions on e4m3 operands too (0: bf16, the round-5 form)


def fp8_plan(w=None, w2=None, act=None, dyq=None, emit=None, tn=False, keep=None, keep2=None):
    """struct vipant_fp8_plan for one fused-operator call: w / w2 = (bytes, row scales) of the operator's weights, act = (scratch
    bytes [M, 4D], its block scales), emit = the same pair for the e4m3 form an MLP operator's first contraction leaves for its
    second, dyq = (bytes [M, D], block scales): the quantised stream gradient that travels between the backward operators;
    tn: the operator's weight-gradient contractions run on e4m3 operands; keep / keep2: block-uniform e4m3 forms of its kept forward
    activations."""
    from ._ffi import Fp8Plan
    return Fp8Plan(w[0].data_ptr(), w[1].data_ptr(), w2[0].data_ptr() if w2 else None, w2[1].data_ptr() if w2 else None,
                   act[0].data_ptr(), act[1].data_ptr(), emit[0].data_ptr() if emit else None, emit[1].data_ptr() if emit else None,
                   dyq[0].data_ptr() if dyq else None, dyq[1].data_ptr() if dyq else None, int(bool(tn)),
                   keep[0].data_ptr() if keep else None, keep[1].data_ptr() if keep else None,
                   keep2[0].data_ptr() if keep2 else None, keep2[1].data_ptr() if keep2 else None)


def fp8_scratch(M: int, D: int, device):
    """(bytes [M, 4D], block scales) for an e4m3 activation operand of up to 4 D columns."""
    return (torch.empty((M, 4 * D), dtype=torch.uint8, device=device),
            torch.empty((query("vipant_mx_scale_bytes", M, 4 * D),), dtype=torch.uint8, device=device))


_frozen_cache: Dict[tuple, tuple] = {}


def cached_bf16(w: torch.Tensor, transpose_only: bool = False):
    """bf16 copy of a frozen weight, cached until the tensor is modified (frozen towers, eval).

    Keyed by address + version + shape; every entry also holds the source's storage, so the address cannot be handed
    to another tensor while the entry is alive (a freed tower's weights would otherwise alias a later model's)."""
    key = (w.data_ptr(), w._version, tuple(w.shape), transpose_only)
    hit = _frozen_cache.get(key)
    if hit is None:
        # an updated weight (the optimizer bumps its version): drop the copies made from its earlier values
        for stale in [k for k in _frozen_cache if k[0] == key[0] and k[2:] == key[2:] and k[1] != key[1]]:
            _frozen_cache.pop(stale)
        while len(_frozen_cache) >= 1024:
            _frozen_cache.pop(next(iter(_frozen_cache)))
        hit = (cast_bf16(w.detach(), transpose=transpose_only), w.untyped_storage())
        _frozen_cache[key] = hit
    return hit[0][1] if transpose_only else hit[0][0]


# ---------------------------------------------------------------------------------- node-to-node gradient hand-off
# The three autograd nodes of a tower (patch embedding -> transformer stack -> read-out) exchange the residual stream as fp32
# tensors, and autograd wants each node's input gradient as a dense fp32 tensor of that shape: for the read-out that is a
# [M, D] matrix of zeros with `batch` non-zero rows (a 497 MB fill + a cast back to bf16 at cfg2), for the stack a bf16 -> fp32
# cast that the patch embedding's LayerNorm backward immediately undoes.  When a node finds its neighbour in the autograd graph,
# it hands the compact form over directly (an attribute on the neighbour's ctx) and returns a zero-stride placeholder of the
# right shape; the receiver uses the hand-off only if what autograd delivers IS that placeholder -- the very allocation the sender
# made, checked by address, not by shape (another consumer of the same tensor makes autograd sum real gradients into a new tensor,
# and the dense path takes over, with the hand-off added in).  Guards (round 4): a tensor that retains its gradient or
# carries hooks at forward time is never handed around (the hook would see zeros); a hand-off the receiver never consumed -- a
# partial backward over the sender only -- is dropped by the sender's next backward instead of being added a second time; the
# sender keeps its link to the receiver across backward calls (retain_graph).  Hooks registered on the stream AFTER the forward
# cannot be seen from here: they observe the placeholder (zeros); VIPANT_NODE_HANDOFF=0 gives dense gradients everywhere.
_VIEW_NODES = ("ViewBackward0", "ReshapeAliasBackward0", "UnsafeViewBackward0", "AliasBackward0", "ViewBackward1")


NODE_HANDOFF = os.environ.get("VIPANT_NODE_HANDOFF", "1") != "0"     # 0: dense fp32 gradients between the nodes (tests, A/B)


def _producer(t: torch.Tensor, kind: str):
    """The custom node of `kind` that produced `t` (looking through view / reshape nodes), or None."""
    if not NODE_HANDOFF:
        return None
    if getattr(t, "retains_grad", False) or getattr(t, "_backward_hooks", None):
        return None                                  # somebody watches this tensor's gradient: keep it dense
    node = t.grad_fn
    for _ in range(4):
        if node is None:
            return None
        if getattr(node, "_vipant_kind", None) == kind:
            return node
        if type(node).__name__ not in _VIEW_NODES or not node.next_functions:
            return None
        node = node.next_functions[0][0]
    return None


_consts: Dict[Tuple[str, int], torch.Tensor] = {}


def _const(name: str, value: float, device) -> torch.Tensor:
    """A 0-dim fp32 constant per device, created once (every `torch.zeros(())` / `ones_like(loss)` inside the step is a fill launch
    on a dependent chain)."""
    key = (name, device.index if device.index is not None else torch.cuda.current_device())
    t = _consts.get(key)
    if t is None:
        t = _consts[key] = torch.full((), value, dtype=F32, device=device)
    return t


def unit_grad(device) -> torch.Tensor:
    """d loss / d loss = 1 as a cached tensor: `loss.backward(gradient=ops.unit_grad(dev))` saves autograd's `ones_like` fill, and
    InfoNCEFn.backward recognises it (by address) and skips the two multiplications by 1."""
    return _const("one", 1.0, torch.device(device))


def _placeholder(shape, device) -> torch.Tensor:
    # ONE zero scalar per device serves every hand-off: the receiver only asks "is this the 4-byte allocation seen through zero
    # strides?", and a gradient autograd has summed from several consumers never is
    return _const("zero", 0.0, torch.device(device)).expand(*shape)


def _is_placeholder(t: torch.Tensor, ph: torch.Tensor) -> bool:
    """`t` is the placeholder `ph` a neighbour returned to autograd: the same 4-byte allocation seen through zero strides (autograd
    may re-wrap the tensor on the way; it cannot give another tensor this address while `ph` is alive)."""
    return ph is not None and t.dim() == 2 and t.stride() == (0, 0) and t.data_ptr() == ph.data_ptr()


# ---------------------------------------------------------------------------------- patch embedding
class PatchEmbedFn(torch.autograd.Function):
    """ViTPreEncoder.forward (cvap/module/val.py:228-259): patch conv as im2col + MFMA contraction, cls token,
    positional table, ln_pre -- vipant_patch_embed_ln_{fwd,bwd}.  x fp32 [b, C, T, F] -> residual stream fp32 [b*S, D]."""

    @staticmethod
    def forward(ctx, x, conv_w, cls, pos, ln_w, ln_b, stride):
        _need(x, F32, "patch_embed.x")
        x = x.contiguous()
        b, Cx, T, Fq = x.shape
        D, Cw, ph, pw = conv_w.shape
        sh, sw = int(stride[0]), int(stride[1])
        nrow, ncol = (T - ph) // sh + 1, (Fq - pw) // sw + 1
        P, S = nrow * ncol, nrow * ncol + 1
        assert pos.shape[0] >= S, f"positional table has {pos.shape[0]} rows, need {S}"
        # cvap/module/val.py:236-244: non-RGB input with a multi-channel kernel -> channel-mean kernel
        mean_ch = Cx != 3 and Cx != Cw
        assert mean_ch or Cx == Cw, f"input has {Cx} channels, kernel {Cw}"
        kcols = (1 if mean_ch else Cw) * ph * pw
        dev = x.device
        w_eff = torch.empty((D, kcols), dtype=BF16, device=dev)
        patches = torch.empty((b * P, kcols), dtype=BF16, device=dev)
        tok = torch.empty((b * S, D), dtype=F32, device=dev)
        out = torch.empty((b * S, D), dtype=F32, device=dev)
        mean = torch.empty((b * S,), dtype=F32, device=dev)
        rstd = torch.empty((b * S,), dtype=F32, device=dev)
        call("vipant_patch_embed_ln_fwd", x.data_ptr(), conv_w.detach().contiguous().data_ptr(),
             cls.detach().contiguous().data_ptr(), pos.detach().contiguous().data_ptr(), ln_w.detach().data_ptr(),
             ln_b.detach().data_ptr(), w_eff.data_ptr(), patches.data_ptr(), None, tok.data_ptr(), out.data_ptr(),
             mean.data_ptr(), rstd.data_ptr(), b, Cx, T, Fq, D, Cw, ph, pw, sh, sw, int(mean_ch), _stream())
        ctx.save_for_backward(patches, tok, mean, rstd, ln_w)
        ctx.meta = (b, P, D, Cw, ph * pw, mean_ch, tuple(conv_w.shape), tuple(pos.shape))
        ctx._vipant_kind = "patch"
        ctx.stream_grad = None          # the stack's bf16 stream gradient, handed over by BackboneFn.backward
        return out

    @staticmethod
    def backward(ctx, dout):
        patches, tok, mean, rstd, ln_w = ctx.saved_tensors
        b, P, D, Cw, khw, mean_ch, conv_shape, pos_shape = ctx.meta
        dev = dout.device
        handed, ctx.stream_grad = ctx.stream_grad, None
        dout_bf16 = handed is not None and _is_placeholder(dout, handed[1])
        if dout_bf16:
            dout = handed[0]
        else:
            dout = dout.contiguous()
            if handed is not None:      # another consumer of the stream contributed a real gradient: add the stack's to it
                dout = dout + handed[0].to(F32)
        kcols = patches.shape[1]
        dtok = torch.empty_like(tok)
        dpatch = torch.empty((b * P, D), dtype=BF16, device=dev)
        dw_eff = torch.empty((D, kcols), dtype=F32, device=dev)
        dconv = torch.empty(conv_shape, dtype=F32, device=dev) if mean_ch else dw_eff.view(conv_shape)
        dcls = torch.empty((D,), dtype=F32, device=dev)
        dpos = torch.empty(pos_shape, dtype=F32, device=dev)      # rows 0 .. P are written by the kernel
        if pos_shape[0] > P + 1:
            dpos[P + 1:].zero_()                                  # a table longer than the sequence: the unused rows' gradient
        dlnw = torch.empty((D,), dtype=F32, device=dev)
        dlnb = torch.empty((D,), dtype=F32, device=dev)
        ws = scratch("patch_embed_bwd", query("vipant_patch_embed_ln_bwd_workspace_bytes", b, P, D, kcols), dev)
        call("vipant_patch_embed_ln_bwd", dout.data_ptr(), int(dout_bf16), tok.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
             ln_w.detach().data_ptr(), patches.data_ptr(), dtok.data_ptr(), dpatch.data_ptr(), dw_eff.data_ptr(),
             dconv.data_ptr(), dcls.data_ptr(), dpos.data_ptr(), dlnw.data_ptr(), dlnb.data_ptr(), b, P, D, Cw, khw,
             int(mean_ch), ws.data_ptr(), ws.numel(), _stream())
        return None, dconv, dcls, dpos, dlnw, dlnb, None


# ---------------------------------------------------------------------------------- transformer
class _LayerGrads:
    """fp32 gradients of one block, carved from one flat buffer so a replica group can all-reduce a whole
    layer with a single RCCL call as soon as its backward is done."""

    def __init__(self, shapes: Sequence[torch.Size], device):
        sizes = [int(torch.Size(s).numel()) for s in shapes]
        pad = [(n + 3) // 4 * 4 for n in sizes]            # keep every view 16-byte aligned
        self.flat = torch.empty((sum(pad),), dtype=F32, device=device)
        self.views, off = [], 0
        for s, n, p in zip(shapes, sizes, pad):
            self.views.append(self.flat[off:off + n].view(s))
            off += p


# precision of the residual-stream GRADIENT: `running.grad_stream` (bf16 | fp32; monitor.py sets it), default from the environment
GRAD_STREAM_F32 = os.environ.get("VIPANT_GRAD_STREAM", "bf16") == "fp32"
# the last block's one-query attention with the key / value projection folded into the query side (csrc/readout_ctx.hip);
# 0: project every token to K and V and attend over those (csrc/readout_rows.hip, the round-3 form)
LAST_BLOCK_CTX = os.environ.get("VIPANT_LAST_BLOCK_CTX", "1") != "0"
# 0: the e4m3 forms of the attention output and of dqkv by the stand-alone pass instead of the attention kernels' epilogues (A/B)
ATTN_EMIT = os.environ.get("VIPANT_ATTN_EMIT", "1") != "0"


def heads_to_wide(rows: torch.Tensor, w_t_cols: torch.Tensor, out: torch.Tensor, H: int) -> torch.Tensor:
    """out[(i, h), :] = rows[i, 64 h : 64 h + 64] . W_h, all heads in one launch (vipant_gemm_nt_heads): rows bf16 [n, D]; w_t_cols:
    the column block [:, c0 : c0 + D] of a TRANSPOSED weight matrix (bf16 [D, ld], a view: W_h^T is its columns 64 h ..); out bf16
    [n * H, D], or [2, n * H, D]: the result as a bf16 pair (hi plane, lo plane)."""
    n, D = rows.shape
    pair = out.dim() == 3
    assert D == 64 * H and tuple(out.shape[-2:]) == (n * H, D) and w_t_cols.shape == (D, D) and w_t_cols.stride(1) == 1 and out.is_contiguous()
    call("vipant_gemm_nt_heads", rows.data_ptr(), D, 64, 0, w_t_cols.data_ptr(), w_t_cols.stride(0), 64, out.data_ptr(), H * D, D,
         n * H * D if pair else 0, None, 0, n, D, 64, H, _stream())
    return out


def wide_to_heads(wide: torch.Tensor, w_rows: torch.Tensor, H: int, bias=None) -> torch.Tensor:
    """out[i, 64 h : 64 h + 64] = wide[(i, h), :] . W_h^T (+ bias of those columns), all heads in one launch: wide bf16 [n * H, D]
    (or a bf16 pair [2, n * H, D]: both planes enter the product); w_rows: the row block [r0 : r0 + D] of a weight matrix (bf16
    [D, D], contiguous: W_h is its rows 64 h ..); out bf16 [n, D]."""
    pair = wide.dim() == 3
    nH, D = wide.shape[-2:]
    n = nH // H
    assert D == 64 * H and nH % H == 0 and wide.is_contiguous() and w_rows.shape == (D, D) and w_rows.is_contiguous()
    out = torch.empty((n, D), dtype=BF16, device=wide.device)
    call("vipant_gemm_nt_heads", wide.data_ptr(), H * D, D, nH * D if pair else 0, w_rows.data_ptr(), D, 64 * D, out.data_ptr(), D, 64, 0,
         _ptr(bias), 64, n, 64, D, H, _stream())
    return out


def head_expand(rows: torch.Tensor, H: int) -> torch.Tensor:
    """rows bf16 [n, 64 H] -> [n * H, 64 H]: row (i, h) keeps head h's 64 columns of row i, zeros elsewhere."""
    _need(rows, BF16, "head_expand.rows")
    n, D = rows.shape
    assert D == 64 * H and rows.is_contiguous()
    out = torch.empty((n * H, D), dtype=BF16, device=rows.device)
    call("vipant_head_expand", rows.data_ptr(), out.data_ptr(), n, H, _stream())
    return out


def head_extract(full: torch.Tensor, H: int, bias=None) -> torch.Tensor:
    """full bf16 | fp32 [n * H, 64 H] -> bf16 [n, 64 H]: head h's 64 columns of row (i, h) (+ bias)."""
    nH, D = full.shape
    assert D == 64 * H and nH % H == 0 and full.is_contiguous() and full.dtype in (BF16, F32)
    out = torch.empty((nH // H, D), dtype=BF16, device=full.device)
    call("vipant_head_extract", full.data_ptr(), int(full.dtype == F32), _ptr(bias), out.data_ptr(), nH // H, H, _stream())
    return out


class BackboneFn(torch.autograd.Function):
    """TransformerBackbone.forward = L x ResidualAttentionBlock (cvap/module/val.py:493-522) on the fp32
    residual stream x [batch*S, D].  One autograd node for the whole stack: the layer loop, the saved
    activations and the (fp32, bf16) gradient-stream pair between layers are managed here, not by autograd.
    Per block four entry points of the fused operator set each way (include/vipant_hip.h): vipant_ln_qkv_*, vipant_mha_*,
    vipant_gemm_bias_residual_*, vipant_ln_mlp_quickgelu_*.

    `rows` (None | "first" | int64 [batch]): the caller will read ONE row per item of the output -- the class token
    (ViTPostEncoder, cvap/module/val.py:288-289) or the end-of-text token (GPTPostEncoder, val.py:143-145) -- and the node returns
    just those rows, [batch, D].  The last block is then evaluated on them alone (csrc/readout_rows.hip): ln_1 and the key / value
    projection see every token, the query projection, the attention (one query per item and head), out_proj, ln_2 and the MLP see
    `batch` rows, forward and backward.  Exact: the other rows of the last block's output are never read and carry no gradient, so
    every feature and every parameter gradient is what the full block gives (tests/test_model_gpu.py, against the reference's
    vectors and against rows=None)."""

    @staticmethod
    def forward(ctx, x, batch, S, causal, grad_sync, recompute_mlp, fp8, stream_f16, rows, *params):
        _need(x, F32, "backbone.x")
        x_in = x
        M, D = x.shape
        assert M == batch * S and len(params) % 12 == 0
        L, H = len(params) // 12, D // 64
        prune = rows is not None and L > 0
        ridx = rows if isinstance(rows, torch.Tensor) else None
        if ridx is not None:
            _need(ridx, I64, "backbone.rows")
            assert ridx.numel() == batch and ridx.is_contiguous()
        train = any(ctx.needs_input_grad)
        dev = x.device
        st = _stream()
        saved: List[torch.Tensor] = []
        wts: List[Tuple[torch.Tensor, ...]] = []
        kept_q: List = []           # per full block: (bytes, block scales) of ln_2's output (e4m3 towers with recompute_mlp)
        x = x.contiguous()

        def new(cols, dtype=BF16):
            return torch.empty((M, cols), dtype=dtype, device=dev)
        # `running.stream_dtype: fp16`: inside the stack the residual stream is kept in the reference's own autocast precision
        # (clip/model.py:157-160) -- every LayerNorm pass reads and writes 2 instead of 4 bytes per element of it, and so do the
        # saved copies the backward reads; statistics stay fp32, the norm is taken on the unrounded sum, and the tensors that
        # cross the autograd boundary (the stack's input and output) stay fp32
        SDT = F16 if stream_f16 else F32
        out16 = _ffi.STREAM_OUT_F16 if stream_f16 else 0

        def sflags(t):
            return (_ffi.STREAM_IN_F16 if t.dtype == F16 else 0) | out16
        # Per block (cvap/module/val.py:519-522):  x1 = x + attn(ln_1(x));  x2 = x1 + mlp(ln_2(x1)).
        # The branch outputs y1, y2 leave their contraction as bf16 and the residual add is fused into the NEXT
        # LayerNorm pass (fp32 stream in, fp32 stream + bf16 normalised activations out), so every contraction has a
        # plain single-output bf16 epilogue and the fp32 stream is only touched by the streaming LN kernels.
        keep_mlp = train and not recompute_mlp
        if not train:   # frozen tower: one set of temporaries for all layers
            h1, h2, qkv, y1, y2 = new(D), new(D), new(3 * D), new(D), new(D)
            mean = torch.empty((M,), dtype=F32, device=dev); rstd = torch.empty((M,), dtype=F32, device=dev)
        fp8 = bool(fp8)
        keep_q = fp8 and train and FP8_TN and D % 128 == 0      # the forward keeps e4m3 forms for the e4m3 weight-gradient contractions
        kept_s: List = []           # per full block: the block scales of the kept e4m3 forms
        # (e4m3 towers that do not keep the MLP activations need neither: the c_fc epilogue leaves g's e4m3 form for c_proj and nothing else)
        if not keep_mlp and not fp8:        # u: 8-bit codes of QuickGELU'(pre-activation) -- all the backward needs of it
            u, g = new(4 * D, torch.uint8), new(4 * D)
        y_prev = None
        ctx_alg = prune and LAST_BLOCK_CTX and H in (8, 12, 16) and S <= 1024
        # e4m3 operands in the NT contractions (configs[4]): a property of the tower, whether or not this call records a backward --
        # the no-grad feature pass of `running.micro_batch` and evaluation must see the forward the training pass differentiates
        if train or fp8:        # bf16 copies (W and W^T) of the 4 L weight matrices: one launch per step
            mats = [params[12 * l + i] for l in range(L) for i in (2, 4, 8, 10)]
            if fp8:
                wb_all, wt_all, wq_all, wtq_all = cast_weights(mats, e4m3=True)
                act, emit = fp8_scratch(M, D, dev), fp8_scratch(M, D, dev)
            else:
                wb_all, wt_all = cast_weights(mats)
        for l in range(L):
            ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wpr, bpr = (p.detach() for p in params[12 * l:12 * l + 12])
            if fp8:
                q_qkv, q_o, q_fc, q_pr = wq_all[4 * l:4 * l + 4]
            if train:
                wqkv_b, wo_b, wfc_b, wpr_b = wb_all[4 * l:4 * l + 4]
                wqkv_t, wo_t, wfc_t, wpr_t = wt_all[4 * l:4 * l + 4]
                wts.append((wqkv_t, wo_t, wfc_t, wpr_t, wfc_b if recompute_mlp else None) +
                           ((wq_all[4 * l:4 * l + 4], wtq_all[4 * l:4 * l + 4]) if fp8 else (None, None)))
                if prune and l == L - 1:         # the last block keeps its per-token activations for the read-out rows only
                    h1, qkv = new(D), (None if ctx_alg else new(3 * D))
                    mean1, rstd1 = (torch.empty((M,), dtype=F32, device=dev) for _ in range(2))
                else:
                    qkv, y1, y2 = new(3 * D), new(D), new(D)
                    mean1, rstd1, mean2, rstd2 = (torch.empty((M,), dtype=F32, device=dev) for _ in range(4))
                    if keep_q:
                        # e4m3 weight gradients: of the two LayerNorm outputs and of g the backward reads the e4m3 forms alone (block-
                        # uniform scales: static ones from the LayerNorm passes, the c_fc epilogue's for g) -- 1 instead of 2 bytes per
                        # element kept, and the bf16 tensors are never written; the attention output's e4m3 form (what out_proj reads)
                        # is kept beside the bf16 one the attention backward needs
                        h1 = h2 = None
                        oq = (new(D, torch.uint8), torch.empty((query("vipant_mx_scale_bytes", M, D),), dtype=torch.uint8, device=dev)) if (H % 2 == 0 and ATTN_EMIT) else None
                        h1q, h2q = (new(D, torch.uint8), torch.empty((query("vipant_mx_scale_bytes", M, D),), dtype=torch.uint8, device=dev)), (new(D, torch.uint8), torch.empty((query("vipant_mx_scale_bytes", M, D),), dtype=torch.uint8, device=dev))
                    else:
                        h1, h2 = new(D), new(D)
                    if keep_mlp and keep_q:
                        u, g = new(4 * D, torch.uint8), None
                        gq = (new(4 * D, torch.uint8), torch.empty((query("vipant_mx_scale_bytes", M, 4 * D),), dtype=torch.uint8, device=dev))
                    elif keep_mlp:
                        u, g = new(4 * D, torch.uint8), new(4 * D)
            else:
                wqkv_b, wo_b, wfc_b, wpr_b = wb_all[4 * l:4 * l + 4] if fp8 else (cached_bf16(w) for w in (wqkv, wo, wfc, wpr))
                mean1 = mean2 = mean; rstd1 = rstd2 = rstd
            if prune and l == L - 1:
                # the last block on the read-out rows: bf16 contractions whatever `fp8` says -- all of them launches on `batch` or
                # `batch * H` rows.  (Only the round-3 form, VIPANT_LAST_BLOCK_CTX=0, still has a per-token launch here, the K / V
                # projection, N = 2 D: it stays bf16 too, a documented departure from `running.fp8_gemm` for 1 of 97 contraction
                # launches.)  `running.last_block_rows=False` restores the e4m3 last block; both corners are under
                # test_end_to_end_golden_e4m3
                def newr(cols, dtype=BF16):
                    return torch.empty((batch, cols), dtype=dtype, device=dev)
                xs = new(D, SDT) if y_prev is not None else None
                call("vipant_layernorm_fwd_e4m3", x.data_ptr(), D, ln1w.data_ptr(), ln1b.data_ptr(), h1.data_ptr(), None,
                     mean1.data_ptr(), rstd1.data_ptr(), M, D, _ptr(y_prev), _ptr(xs), None, None, sflags(x), st)
                if xs is not None:
                    x = xs
                h1_r = gather_rows(h1, ridx, batch, S)
                q_r = gemm_nt(h1_r, wqkv_b[:D], newr(D), bias=bqkv[:D], epi=EPI_BF16, few_rows=True)     # Q of the read-out rows
                probs = torch.empty((batch, H, S), dtype=F32, device=dev)
                if ctx_alg:
                    # one query per (item, head): the key projection moves to the query (qk_h = W_k,h^T q_h, the key bias drops out
                    # of the softmax), the value projection behind the weighted sum (o_h = W_v,h sum_j p_j h1_j + b_v,h): two
                    # per-head contractions on `batch` rows and one pass over h1 instead of K, V of every token
                    wk_t = (wqkv_t if train else cached_bf16(wqkv, transpose_only=True))[:, D:2 * D]
                    # qk and the heads' contexts travel as bf16 PAIRS (hi + lo planes, round 5): [0]: qk, [1]: contexts
                    qkv = torch.empty((2, 2, batch * H, D), dtype=BF16, device=dev)
                    heads_to_wide(q_r, wk_t, qkv[0], H)
                    call("vipant_rows_ctx_fwd", qkv[0].data_ptr(), h1.data_ptr(), _ptr(ridx), qkv[1].data_ptr(), probs.data_ptr(),
                         batch, S, H, int(causal), 1, st)
                    o_r = wide_to_heads(qkv[1], wqkv_b[2 * D:], H, bias=bqkv[2 * D:])
                else:
                    if qkv is None:
                        qkv = new(3 * D)
                    gemm_nt(h1, wqkv_b[D:], qkv[:, D:], bias=bqkv[D:], epi=EPI_BF16)         # K, V of every token
                    o_r = newr(D)
                    call("vipant_mha_rows_fwd", q_r.data_ptr(), qkv.data_ptr(), _ptr(ridx), o_r.data_ptr(), probs.data_ptr(), batch, S,
                         H, int(causal), st)
                y1_r = gemm_nt(o_r, wo_b, newr(D), bias=bo, epi=EPI_BF16, few_rows=True)
                x_r = gather_rows(x, ridx, batch, S)
                # the read-out rows' stream stays fp32 from here, and c_proj adds it in its epilogue: the rows the features are
                # read from skip the two roundings (fp16 stream, bf16 branch output) the full block would give them
                x1_r, h2_r, u_r, g_r = newr(D, F32), newr(D), newr(4 * D, torch.uint8), newr(4 * D)
                mean2_r, rstd2_r = (torch.empty((batch,), dtype=F32, device=dev) for _ in range(2))
                call("vipant_layernorm_fwd_e4m3", x_r.data_ptr(), D, ln2w.data_ptr(), ln2b.data_ptr(), h2_r.data_ptr(), None,
                     mean2_r.data_ptr(), rstd2_r.data_ptr(), batch, D, y1_r.data_ptr(), x1_r.data_ptr(), None, None,
                     _ffi.STREAM_IN_F16 if x_r.dtype == F16 else 0, st)
                gemm_nt(h2_r, wfc_b, g_r, bias=bfc, aux=u_r, epi=EPI_QUICKGELU_D8, few_rows=True)
                out_r = gemm_nt(g_r, wpr_b, newr(D, F32), bias=bpr, aux=x1_r, epi=EPI_RESIDUAL_F32, few_rows=True)
                if train:
                    saved += [x, mean1, rstd1, h1, qkv, q_r, probs, o_r, h1_r, x1_r, mean2_r, rstd2_r, h2_r, u_r, g_r]
                x, y_prev = out_r, None
                continue
            # ln_1 (+ residual add of the previous block's MLP branch: x <- x + y2_prev) + in_proj
            xs = new(D, SDT) if y_prev is not None else None
            call("vipant_ln_qkv_fwd_e4m3", x.data_ptr(), _ptr(y_prev), _ptr(xs), ln1w.data_ptr(), ln1b.data_ptr(), wqkv_b.data_ptr(),
                 bqkv.data_ptr(), _ptr(h1), mean1.data_ptr(), rstd1.data_ptr(), qkv.data_ptr(), M, D,
                 C.byref(fp8_plan(q_qkv, None, h1q if (train and keep_q) else act)) if fp8 else None, sflags(x), st)
            if xs is not None:
                x = xs
            # (e4m3: the attention kernel leaves its output's e4m3 form in the activation scratch; out_proj reads it from there)
            q8 = ((oq if (train and keep_q) else act) if (fp8 and H % 2 == 0 and ATTN_EMIT) else None)
            o, lse = mha_fwd(qkv, batch, S, H, causal, q8=q8)
            call("vipant_gemm_bias_residual_fwd_e4m3", None if q8 else o.data_ptr(), wo_b.data_ptr(), bo.data_ptr(), None, y1.data_ptr(),
                 M, D, D, C.byref(fp8_plan(q_o, None, q8 or act)) if fp8 else None, st)
            # ln_2 (+ residual add of the attention branch) + c_fc + QuickGELU + c_proj
            x1 = new(D, SDT)
            if "gemm_nt" in KERNEL_PROBE and not fp8:     # bench.py times the c_fc launch alone: the same three launches, issued one by one
                call("vipant_layernorm_fwd_e4m3", x.data_ptr(), D, ln2w.data_ptr(), ln2b.data_ptr(), h2.data_ptr(), None,
                     mean2.data_ptr(), rstd2.data_ptr(), M, D, y1.data_ptr(), x1.data_ptr(), None, None, sflags(x), st)
                gemm_nt(h2, wfc_b, g, bias=bfc, aux=u, epi=EPI_QUICKGELU_D8)
                gemm_nt(g, wpr_b, y2, bias=bpr, epi=EPI_BF16)
            else:
                only_q = fp8 and not keep_mlp        # c_proj's operand is all that is wanted of g: neither g nor the codes are written
                # `recompute_mlp` under e4m3: the LayerNorm output's e4m3 form is kept beside it (M D bytes + scales per block), so
                # that the backward's c_fc launch needs no quantisation pass either
                if not (train and keep_q):
                    h2q = (new(D, torch.uint8), torch.empty((query("vipant_mx_scale_bytes", M, D),), dtype=torch.uint8, device=dev)) if (fp8 and train and recompute_mlp) else None
                call("vipant_ln_mlp_quickgelu_fwd_e4m3", x.data_ptr(), y1.data_ptr(), x1.data_ptr(), ln2w.data_ptr(), ln2b.data_ptr(),
                     wfc_b.data_ptr(), bfc.data_ptr(), wpr_b.data_ptr(), bpr.data_ptr(), _ptr(h2), mean2.data_ptr(),
                     rstd2.data_ptr(), None if only_q else u.data_ptr(), None if (only_q or g is None) else g.data_ptr(), y2.data_ptr(), M, D,
                     C.byref(fp8_plan(q_fc, q_pr, h2q or act, emit=(gq if (keep_mlp and keep_q) else emit))) if fp8 else None, sflags(x), st)
            if train:
                # `recompute_mlp`: the two [M, 4D] MLP activations (16 of the 36 D bytes a block keeps per token) are not
                # kept; the backward re-runs the c_fc contraction (+1 of a block's 12 contractions) to get them back
                saved += [x, mean1, rstd1, h1q[0] if keep_q else h1, qkv, o, lse, x1, mean2, rstd2, h2q[0] if keep_q else h2] + \
                         ([u, gq[0] if keep_q else g] if keep_mlp else [])
                if fp8 and recompute_mlp:
                    kept_q.append(h2q)
                kept_s.append(dict(h1=h1q[1], h2=h2q[1], o=oq, **({"g": gq[1]} if keep_mlp else {})) if keep_q else {})
            x, y_prev = x1, y2
        x = residual_add(x, y_prev) if y_prev is not None else x
        if train:
            ctx.save_for_backward(*saved, *params)
            ctx.wts = wts
            ctx.kept_q = kept_q
            ctx.kept_s = kept_s
            ctx.keep_q = keep_q
            ctx.tn8 = FP8_TN        # (decided by the forward: the backward must read what that forward kept)
            ctx.meta = (batch, S, bool(causal), L, H, bool(recompute_mlp), fp8, prune)
            ctx.rows = ridx
            ctx.last_ctx = ctx_alg
            ctx.wqkv_b_last = wqkv_b if ctx_alg else None     # bf16 in_proj weight of the last block (its K rows: dq = W_k dqk)
            ctx.grad_sync = grad_sync
            ctx._vipant_kind = "stack"
            ctx.readout_grad = None                      # (idx | None, compact fp32 rows), handed over by ReadoutFn.backward
            ctx.patch_node = _producer(x_in, "patch")    # the patch embedding that produced this stack's input, if any
        return x

    @staticmethod
    def backward(ctx, dx_in):
        batch, S, causal, L, H, recompute_mlp, fp8, prune = ctx.meta
        ridx = ctx.rows
        tensors = ctx.saved_tensors
        ns = 11 if recompute_mlp else 13
        nsaved = ns * (L - 1) + 15 if prune else ns * L
        saved, params = tensors[:nsaved], tensors[nsaved:]
        dev = dx_in.device
        st = _stream()
        M, D = batch * S, dx_in.shape[1]
        # Gradient of the residual stream.  Default: bf16 only -- the tensor the next contraction reads IS the stream (LayerNorm
        # backward 10 instead of 16 B per element); the forward stream stays fp32, so the loss and the features are untouched and
        # the gradients move from ~1.3 % to ~1.6 % rel-L2 of the fp32 reference (profiles/r2_stream_precision.md, model D; the
        # reference's own GPU path keeps this stream in fp16).  VIPANT_GRAD_STREAM=fp32: fp32 master + bf16 copy, both in place.
        if ctx.wts is None:
            raise _ffi.VipantError("BackboneFn: second backward through the same forward -- the node releases its bf16 weight copies "
                                   "and saved activations' bookkeeping in its first backward (retain_graph re-entry is not supported)")
        handed, ctx.readout_grad = ctx.readout_grad, None
        top_rows = None
        if prune:
            # the gradient arrives for the read-out rows only, [batch, D]: dense, or handed over by the read-out node
            if handed is not None:
                dx_in = handed[1] if _is_placeholder(dx_in, handed[2]) else dx_in + handed[1]
            dx, dx_b = None, None
            dxr_b = cast_bf16_flat(dx_in.contiguous())
        elif handed is not None and _is_placeholder(dx_in, handed[2]) and not GRAD_STREAM_F32:
            # the read-out's gradient as compact rows: they go straight into a zeroed bf16 stream gradient
            ridx, rows, _ = handed
            dx = None
            dx_b = torch.zeros((M, D), dtype=BF16, device=dev)
            top_rows = torch.empty((rows.shape[0], D), dtype=BF16, device=dev)
            call("vipant_scatter_rows_bf16", rows.data_ptr(), _ptr(ridx), dx_b.data_ptr(), top_rows.data_ptr(), rows.shape[0], S, D, st)
        else:
            if handed is not None:                       # dense gradient from another consumer: add the read-out rows to it
                ridx, rows, _ = handed
                dx_in = dx_in.contiguous().clone()
                call("vipant_scatter_rows", rows.data_ptr(), _ptr(ridx), dx_in.data_ptr(), rows.shape[0], S, D, st)
            if GRAD_STREAM_F32:
                dx = dx_in.contiguous().clone()
                dx_b = cast_bf16_flat(dx)
            else:
                dx = None
                dx_b = cast_bf16_flat(dx_in.contiguous())
        ws = scratch("block_bwd", query("vipant_block_workspace_bytes", M, D), dev)
        act = emit = dyq = None
        if fp8:
            act, emit = fp8_scratch(M, D, dev), fp8_scratch(M, D, dev)
            if not prune:
                dyq = quant_e4m3_mx(dx_b)   # from here on every LayerNorm backward leaves the new stream gradient's e4m3 form beside it
        # scratch shared by all blocks (e4m3 weight gradients: du exists in its e4m3 form alone, in `emit`)
        du = None if (ctx.keep_q and ctx.tn8) else torch.empty((M, 4 * D), dtype=BF16, device=dev)
        dh = torch.empty((M, D), dtype=BF16, device=dev)
        do = torch.empty((M, D), dtype=BF16, device=dev)
        if recompute_mlp:
            u = torch.empty((M, 4 * D), dtype=torch.uint8, device=dev)
            g = None if ctx.keep_q else torch.empty((M, 4 * D), dtype=BF16, device=dev)
            gq_re = fp8_scratch(M, D, dev) if ctx.keep_q else None
        grads: List[Optional[torch.Tensor]] = [None] * (12 * L)
        lg = _LayerGrads([p.shape for p in params[12 * (L - 1):12 * L]], dev)
        # d c_proj.bias of the top block (lower blocks get theirs from ln_1's backward): column sums of the stream gradient, which
        # with a handed-over read-out gradient has only those rows
        colsum(dxr_b if prune else (top_rows if top_rows is not None else dx_b), lg.views[11])
        for l in reversed(range(L)):
            ln1w, _, _, _, _, _, ln2w, _, _, bfc, _, _ = (p.detach() for p in params[12 * l:12 * l + 12])
            wqkv_t, wo_t, wfc_t, wpr_t, wfc_b, wq4, wtq4 = ctx.wts[l]
            (d_ln1w, d_ln1b, d_wqkv, d_bqkv, d_wo, d_bo, d_ln2w, d_ln2b, d_wfc, d_bfc, d_wpr, d_bpr) = lg.views
            lg_below = _LayerGrads([p.shape for p in params[12 * (l - 1):12 * l]], dev) if l > 0 else None
            if prune and l == L - 1:
                # the last block on the read-out rows: the two block operators of the MLP / out_proj half on `batch` rows, the
                # one-query attention backward, then the K / V half of in_proj and ln_1 on every token
                x, mean1, rstd1, h1, qkv, q_r, probs, o_r, h1_r, x1_r, mean2_r, rstd2_r, h2_r, u_r, g_r = saved[nsaved - 15:]
                du_r = torch.empty((batch, 4 * D), dtype=BF16, device=dev)
                dh_r, do_r, dq_r = (torch.empty((batch, D), dtype=BF16, device=dev) for _ in range(3))
                call("vipant_ln_mlp_quickgelu_bwd_e4m3", dxr_b.data_ptr(), wpr_t.data_ptr(), wfc_t.data_ptr(), u_r.data_ptr(),
                     g_r.data_ptr(), h2_r.data_ptr(), x1_r.data_ptr(), mean2_r.data_ptr(), rstd2_r.data_ptr(), ln2w.data_ptr(), None,
                     dxr_b.data_ptr(), du_r.data_ptr(), dh_r.data_ptr(), d_wpr.data_ptr(), d_wfc.data_ptr(), d_bfc.data_ptr(),
                     d_ln2w.data_ptr(), d_ln2b.data_ptr(), d_bo.data_ptr(), batch, D, ws.data_ptr(), ws.numel(), None,
                     (_ffi.STREAM_IN_F16 if x1_r.dtype == F16 else 0) | _ffi.STREAM_FEW_ROWS, st)
                # out_proj backward on the rows (vipant_gemm_bias_residual_bwd's two launches, the first as a few-rows launch)
                gemm_nt(dxr_b, wo_t, do_r, epi=EPI_BF16, few_rows=True)
                gemm_tn(dxr_b, o_r, d_wo, ws_name="block_bwd")
                if ctx.last_ctx:
                    wqkv_b_last = ctx.wqkv_b_last
                    # qkv = [qk | contexts] of the forward.  dctx_h = W_v,h^T do_h; the kernel gives dh of every token and dqk;
                    # dq_h = W_k,h dqk_h; d W_v = do (x) ctx, d W_k = q (x) dqk per head (block-sparse operands), d b_k = 0
                    # (qk, the contexts and dqk are bf16 pairs; the two weight gradients take the hi planes: their own rounding, fp32
                    # sums over `batch * H` bf16 products, is the larger term)
                    qk, hctx = qkv[0], qkv[1]
                    dctx = heads_to_wide(do_r, wqkv_t[:, 2 * D:], torch.empty((batch * H, D), dtype=BF16, device=dev), H)
                    gemm_tn(head_expand(do_r, H), hctx[0], d_wqkv[2 * D:], a_colsum=d_bqkv[2 * D:], ws_name="block_bwd")
                    dqk = torch.empty((2, batch * H, D), dtype=BF16, device=dev)
                    call("vipant_rows_ctx_bwd", qk.data_ptr(), dctx.data_ptr(), hctx.data_ptr(), h1.data_ptr(), _ptr(ridx),
                         probs.data_ptr(), dh.data_ptr(), dqk.data_ptr(), d_bqkv[D:2 * D].data_ptr(), batch, S, H, int(causal), 1, st)
                    dq_r = wide_to_heads(dqk, wqkv_b_last[D:2 * D], H)
                    gemm_tn(head_expand(q_r, H), dqk[0], d_wqkv[D:2 * D], ws_name="block_bwd")
                    del dctx, dqk
                else:
                    dqkv = torch.empty((M, 3 * D), dtype=BF16, device=dev)
                    call("vipant_mha_rows_bwd", q_r.data_ptr(), qkv.data_ptr(), _ptr(ridx), probs.data_ptr(), do_r.data_ptr(),
                         dq_r.data_ptr(), dqkv.data_ptr(), batch, S, H, int(causal), st)
                    # dh = dK|dV . W_kv on every token
                    gemm_nt(dqkv[:, D:], wqkv_t[:, D:], dh, epi=EPI_BF16)
                    gemm_tn(dqkv[:, D:], h1, d_wqkv[D:], a_colsum=d_bqkv[D:], ws_name="block_bwd")
                    del dqkv
                # + dq . W_q on the read-out rows
                dhq = gemm_nt(dq_r, wqkv_t[:, :D], torch.empty((batch, D), dtype=F32, device=dev), epi=EPI_F32, few_rows=True)
                call("vipant_add_rows_bf16", dh.data_ptr(), _ptr(ridx), dhq.data_ptr(), 1, batch, S, D, st)
                gemm_tn(dq_r, h1_r, d_wqkv[:D], a_colsum=d_bqkv[:D], ws_name="block_bwd")
                # ln_1 backward on every token; the residual gradient of this block exists on the read-out rows only
                dx_b = torch.empty((M, D), dtype=BF16, device=dev)
                call("vipant_layernorm_bwd_e4m3", dh.data_ptr(), _ffi.LN_X_F16 if x.dtype == F16 else 0, x.data_ptr(), D,
                     mean1.data_ptr(), rstd1.data_ptr(), ln1w.data_ptr(), None, None, D, dx_b.data_ptr(), d_ln1w.data_ptr(),
                     d_ln1b.data_ptr(), lg_below.views[11].data_ptr() if lg_below is not None else None, 0, M, D, ws.data_ptr(),
                     ws.numel(), None, None, st)
                call("vipant_add_rows_bf16", dx_b.data_ptr(), _ptr(ridx), dxr_b.data_ptr(), 0, batch, S, D, st)
                if lg_below is not None:
                    colsum(dxr_b, lg_below.views[11], accumulate=True)
                if GRAD_STREAM_F32:
                    dx = torch.empty((M, D), dtype=F32, device=dev)
                    call("vipant_cast_f32", dx_b.data_ptr(), dx.data_ptr(), M * D, st)
                if fp8:
                    dyq = quant_e4m3_mx(dx_b)
                for i, v in enumerate(lg.views):
                    grads[12 * l + i] = v
                if ctx.grad_sync is not None:
                    ctx.grad_sync.reduce_async(lg.flat, lg.views, params[12 * l:12 * l + 12])
                lg = lg_below
                continue
            x, mean1, rstd1, h1, qkv, o, lse, x1, mean2, rstd2, h2 = saved[ns * l:ns * l + 11]
            kq = ctx.keep_q          # h1, h2 (and g) are the e4m3 forms the forward kept; their block scales are in ctx.kept_s
            if recompute_mlp:
                h2q = ctx.kept_q[l] if fp8 else None
                # (e4m3 weight gradients: the recomputation leaves g's e4m3 form -- the same epilogue, the same bytes as a forward that
                # keeps it -- and no bf16 g)
                call("vipant_mlp_quickgelu_recompute_e4m3", None if fp8 else h2.data_ptr(), wfc_b.data_ptr(), bfc.data_ptr(), u.data_ptr(),
                     None if kq else g.data_ptr(), M, D, C.byref(fp8_plan(wq4[2], None, h2q, emit=gq_re if kq else None)) if fp8 else None, st)
                if fp8:
                    ctx.kept_q[l] = None
                keep = gq_re if kq else None
            else:
                u, g = saved[ns * l + 11:ns * l + 13]
                keep = (g, ctx.kept_s[l]["g"]) if kq else None
            keep2 = (h2, ctx.kept_s[l]["h2"]) if kq else None
            keep1 = (h1, ctx.kept_s[l]["h1"]) if kq else None
            # MLP half: c_proj^T + QuickGELU', c_fc^T, both weight gradients, ln_2 backward (+ residual gradient);
            # the produced stream gradient is also d(out_proj output): its column sum is d out_proj.bias
            call("vipant_ln_mlp_quickgelu_bwd_e4m3", dx_b.data_ptr(), wpr_t.data_ptr(), wfc_t.data_ptr(), u.data_ptr(),
                 None if keep else g.data_ptr(),
                 None if keep2 else h2.data_ptr(), x1.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(), ln2w.data_ptr(), _ptr(dx), dx_b.data_ptr(),
                 _ptr(du), dh.data_ptr(), d_wpr.data_ptr(), d_wfc.data_ptr(), d_bfc.data_ptr(), d_ln2w.data_ptr(),
                 d_ln2b.data_ptr(), d_bo.data_ptr(), M, D, ws.data_ptr(), ws.numel(),
                 C.byref(fp8_plan(wtq4[3], wtq4[2], act, dyq, emit=emit, tn=ctx.tn8, keep=keep, keep2=keep2)) if fp8 else None,
                 _ffi.STREAM_IN_F16 if x1.dtype == F16 else 0, st)
            # attention half: out_proj^T, attention core, in_proj^T + ln_1 backward; the produced stream gradient is
            # d(c_proj output) of the block below: its column sum is that block's d c_proj.bias
            call("vipant_gemm_bias_residual_bwd_e4m3", dx_b.data_ptr(), wo_t.data_ptr(), o.data_ptr(), do.data_ptr(), d_wo.data_ptr(),
                 M, D, D, ws.data_ptr(), ws.numel(),
                 C.byref(fp8_plan(wtq4[1], None, act, dyq, emit=emit, tn=ctx.tn8, keep=ctx.kept_s[l]["o"] if kq else None)) if fp8 else None, st)
            q8 = act if (fp8 and H % 2 == 0 and ATTN_EMIT) else None
            dqkv = mha_bwd(qkv, o, do, lse, batch, S, H, causal, q8=q8)
            call("vipant_ln_qkv_bwd_e4m3", dqkv.data_ptr(), wqkv_t.data_ptr(), None if keep1 else h1.data_ptr(), x.data_ptr(), mean1.data_ptr(),
                 rstd1.data_ptr(), ln1w.data_ptr(), _ptr(dx), dx_b.data_ptr(), dh.data_ptr(), d_wqkv.data_ptr(),
                 d_bqkv.data_ptr(), d_ln1w.data_ptr(), d_ln1b.data_ptr(),
                 lg_below.views[11].data_ptr() if lg_below is not None else None, M, D, ws.data_ptr(), ws.numel(),
                 C.byref(fp8_plan(wtq4[0], None, act, dyq, emit=emit, tn=ctx.tn8, keep=keep1)) if fp8 else None,
                 (_ffi.STREAM_IN_F16 if x.dtype == F16 else 0) | (_ffi.STREAM_ACT_Q if q8 else 0), st)
            del dqkv
            for i, v in enumerate(lg.views):
                grads[12 * l + i] = v
            if ctx.grad_sync is not None:
                ctx.grad_sync.reduce_async(lg.flat, lg.views, params[12 * l:12 * l + 12])
            lg = lg_below
        ctx.wts = ctx.wqkv_b_last = ctx.kept_q = ctx.kept_s = None
        need = ctx.needs_input_grad
        patch = ctx.patch_node                      # kept: a second backward over a retained graph hands over again
        if dx is None and need[0]:
            if patch is not None:        # the patch embedding's LayerNorm backward takes the bf16 stream gradient as it is
                dx = _placeholder((M, D), dev)
                patch.stream_grad = (dx_b, dx)      # (whatever an earlier, never-consumed hand-off left there is replaced)
            else:
                dx = torch.empty((M, D), dtype=F32, device=dev)
                call("vipant_cast_f32", dx_b.data_ptr(), dx.data_ptr(), M * D, st)
        out_grads = [gr if need[9 + i] else None for i, gr in enumerate(grads)]
        return (dx if need[0] else None, None, None, None, None, None, None, None, None, *out_grads)


# ---------------------------------------------------------------------------------- read-out
class ReadoutFn(torch.autograd.Function):
    """ViTPostEncoder / GPTPostEncoder (cvap/module/val.py:288-289, 143-145) + the optional L2 normalisation of
    MetaHead.forward (clip_head.py:117-118): LN(x[b, idx_b]) @ proj -> [batch, E] fp32 -- vipant_cls_ln_proj_l2norm_{fwd,bwd}."""

    @staticmethod
    def forward(ctx, x, idx, batch, S, ln_w, ln_b, proj, normalized):
        _need(x, F32, "readout.x")
        x_arg = x
        x = x.contiguous()
        D, E = proj.shape
        dev = x.device
        train = any(ctx.needs_input_grad)
        rows = torch.empty((batch, D), dtype=F32, device=dev) if idx is not None else None
        y = torch.empty((batch, D), dtype=BF16, device=dev)
        mean = torch.empty((batch,), dtype=F32, device=dev)
        rstd = torch.empty((batch,), dtype=F32, device=dev)
        if train:
            proj_b, proj_t = cast_bf16(proj.detach(), True)
        else:
            proj_b, proj_t = None, cached_bf16(proj, transpose_only=True)
        feat = torch.empty((batch, E), dtype=F32, device=dev)
        out = torch.empty_like(feat) if normalized else feat
        norm = torch.empty((batch,), dtype=F32, device=dev) if normalized else None
        call("vipant_eot_ln_proj_l2norm_fwd" if idx is not None else "vipant_cls_ln_proj_l2norm_fwd", x.data_ptr(), _ptr(idx),
             ln_w.detach().data_ptr(), ln_b.detach().data_ptr(), proj_t.data_ptr(), _ptr(rows), y.data_ptr(), mean.data_ptr(),
             rstd.data_ptr(), feat.data_ptr(), out.data_ptr(), _ptr(norm), batch, S, D, E, int(bool(normalized)), _stream())
        if train:
            ctx.save_for_backward(x, idx, rows, y, mean, rstd, ln_w, proj_b, out, norm)
            ctx.meta = (batch, S, D, E, bool(normalized))
            ctx.stack_node = _producer(x_arg, "stack")       # the transformer stack whose output is read out, if any
        return out

    @staticmethod
    def backward(ctx, dout):
        x, idx, rows, y, mean, rstd, ln_w, proj_b, out, norm = ctx.saved_tensors
        batch, S, D, E, normalized = ctx.meta
        dev = dout.device
        dout = dout.contiguous()
        dfeat = torch.empty((batch, E), dtype=BF16, device=dev)
        dy = torch.empty((batch, D), dtype=BF16, device=dev)
        stack = ctx.stack_node                       # kept across backward calls (retain_graph)
        if stack is not None and stack.readout_grad is not None:
            stack.readout_grad = None                # left by a backward in which the stack's node never ran: not this call's business
        compact = stack is not None and ctx.needs_input_grad[0]
        drows = torch.empty((batch, D), dtype=F32, device=dev) if (idx is not None or compact) else None
        dproj = torch.empty((D, E), dtype=F32, device=dev)
        dlnw = torch.empty((D,), dtype=F32, device=dev)
        dlnb = torch.empty((D,), dtype=F32, device=dev)
        dx = None if compact else torch.zeros_like(x)
        ws = scratch("readout_bwd", query("vipant_cls_ln_proj_l2norm_bwd_workspace_bytes", batch, D, E), dev)
        call("vipant_cls_ln_proj_l2norm_bwd", dout.data_ptr(), out.data_ptr(), _ptr(norm), x.data_ptr(), _ptr(idx), _ptr(rows),
             y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ln_w.detach().data_ptr(), proj_b.data_ptr(), dfeat.data_ptr(),
             dy.data_ptr(), _ptr(drows), _ptr(dx), dproj.data_ptr(), dlnw.data_ptr(), dlnb.data_ptr(), batch, S, D, E,
             int(normalized), ws.data_ptr(), ws.numel(), _stream())
        if compact:                  # hand the rows to the stack's backward; autograd gets a zero-stride placeholder
            dx = _placeholder(tuple(x.shape), dev)
            stack.readout_grad = (idx, drows, dx)
        return dx, None, None, None, dlnw, dlnb, dproj, None


def embed_tokens(tokens: torch.Tensor, table: torch.Tensor, pos: torch.Tensor):
    """GPTPreEncoder.forward (cvap/module/val.py:109-122): returns (x fp32 [b*L, D], eot index i64 [b])."""
    _need(tokens, I64, "embed_tokens.tokens")
    tokens = tokens.contiguous()
    b, L = tokens.shape
    D = table.shape[1]
    assert pos.shape[0] >= L
    x = torch.empty((b * L, D), dtype=F32, device=tokens.device)
    eot = torch.empty((b,), dtype=I64, device=tokens.device)
    call("vipant_embed_gather_pos_fwd", tokens.data_ptr(), table.detach().data_ptr(), pos.detach().contiguous().data_ptr(),
         x.data_ptr(), eot.data_ptr(), b, L, D, _stream())
    return x, eot


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """x / ||x|| for pre-computed features (cvap/model/cvalp.py:45-46); no gradient path (inputs are data)."""
    _need(x, F32, "l2_normalize.x")
    x = x.contiguous()
    out = torch.empty_like(x)
    norm = torch.empty((x.shape[0],), dtype=F32, device=x.device)
    call("vipant_l2norm_fwd", x.data_ptr(), out.data_ptr(), norm.data_ptr(), x.shape[0], x.shape[1], _stream())
    return out


def retrieval_ranks(x1: torch.Tensor, x2: torch.Tensor, gold: torch.Tensor, want_top1: bool = False):
    """Rank of the gold candidates of every query without sorting (cvap/module/decoder/loss_head.py:113-118, 141-160):
    x1 [N1, E], x2 [N2, E] fp32 L2-normalised, gold int [N1] or [N1, G] -> ranks int32 of gold's shape
    (= torch.where((x1 @ x2.t()).argsort(descending=True) == gold)[1]) and, optionally, top1 int32 [N1]."""
    _need(x1, F32, "retrieval.x1"); _need(x2, F32, "retrieval.x2")
    x1, x2 = x1.contiguous(), x2.contiguous()
    if gold.device != x1.device or gold.dtype != torch.int32:
        gold = gold.to(device=x1.device, dtype=torch.int32)
    shape = tuple(gold.shape)
    gold2 = gold.reshape(x1.shape[0], -1).contiguous()
    N1, E = x1.shape
    N2, G = x2.shape[0], gold2.shape[1]
    ranks = torch.empty((N1, G), dtype=torch.int32, device=x1.device)
    top1 = torch.empty((N1,), dtype=torch.int32, device=x1.device) if want_top1 else None
    ws = scratch("retrieval", query("vipant_retrieval_workspace_bytes", N1, N2, E, G), x1.device)
    call("vipant_retrieval_ranks", x1.data_ptr(), x2.data_ptr(), gold2.data_ptr(), ranks.data_ptr(), _ptr(top1), N1, N2, E, G,
         ws.data_ptr(), ws.numel(), _stream())
    ranks = ranks.reshape(shape)
    return (ranks, top1) if want_top1 else ranks


class L2NormFn(torch.autograd.Function):
    """x / ||x|| (cvap/module/decoder/loss_head.py:272-273) with its backward, for un-normalised head outputs."""

    @staticmethod
    def forward(ctx, x):
        _need(x, F32, "l2norm.x")
        x = x.contiguous()
        out = torch.empty_like(x)
        norm = torch.empty((x.shape[0],), dtype=F32, device=x.device)
        call("vipant_l2norm_fwd", x.data_ptr(), out.data_ptr(), norm.data_ptr(), x.shape[0], x.shape[1], _stream())
        ctx.save_for_backward(out, norm)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, norm = ctx.saved_tensors
        dout = dout.contiguous()
        dx = torch.empty_like(out)
        call("vipant_l2norm_bwd", dout.data_ptr(), out.data_ptr(), norm.data_ptr(), dx.data_ptr(), None, out.shape[0],
             out.shape[1], _stream())
        return dx


# ---------------------------------------------------------------------------------- InfoNCE
class InfoNCEFn(torch.autograd.Function):
    """CELossHead.forward (cvap/module/decoder/loss_head.py:265-284) on L2-normalised fp32 features.

    x1, x2: [B, E] -- the (all-gathered) global batch.  Gradients are produced only for rows
    [row0, row0 + nrows): the slice this replica owns; other rows get zero (their owners compute them).
    `grad_scale` pre-multiplies the slice gradients (replicas whose parameter gradients are averaged set it
    to the world size so that the mean over ranks equals the full-batch gradient)."""

    @staticmethod
    def forward(ctx, x1, x2, logit_scale, scale_max, row0, nrows, grad_scale):
        _need(x1, F32, "infonce.x1"); _need(x2, F32, "infonce.x2")
        x1, x2 = x1.contiguous(), x2.contiguous()
        B, E = x1.shape
        dev = x1.device
        need = ctx.needs_input_grad
        want = need[0] or need[1] or need[2]
        # (gradients for a strip of a large batch -- one rank of an N-GPU step -- need a third of the workspace: no B x B matrices)
        ws = scratch("infonce", query("vipant_infonce_strip_workspace_bytes", B, E, int(nrows)) if want
                     else query("vipant_infonce_workspace_bytes", B, E), dev)
        loss = torch.empty((1,), dtype=F32, device=dev)
        d1 = torch.empty((nrows, E), dtype=F32, device=dev) if want else None
        d2 = torch.empty((nrows, E), dtype=F32, device=dev) if want else None
        dls = torch.empty((1,), dtype=F32, device=dev) if want else None
        ls = logit_scale.detach().reshape(1).to(F32)
        call("vipant_infonce_fwd_bwd", x1.data_ptr(), x2.data_ptr(), ls.data_ptr(), float(scale_max or 0.0),
             loss.data_ptr(), _ptr(d1), _ptr(d2), _ptr(dls), float(grad_scale), B, E, int(row0), int(nrows),
             ws.data_ptr(), ws.numel(), _stream())
        if want:
            ctx.save_for_backward(d1, d2, dls)
            ctx.meta = (B, E, int(row0), int(nrows), tuple(logit_scale.shape))
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        d1, d2, dls = ctx.saved_tensors
        B, E, row0, nrows, ls_shape = ctx.meta
        need = ctx.needs_input_grad

        unit = dloss.dim() == 0 and dloss.data_ptr() == unit_grad(dloss.device).data_ptr()      # the trainer's cached 1.0

        def expand(d):
            if row0 == 0 and nrows == B:         # one replica: the slice is the batch (one launch instead of fill + mul + copy)
                return d if unit else d * dloss
            full = torch.zeros((B, E), dtype=F32, device=d.device)
            full[row0:row0 + nrows] = d * dloss
            return full
        return (expand(d1) if need[0] else None, expand(d2) if need[1] else None,
                (dls if unit else dls * dloss).reshape(ls_shape) if need[2] else None, None, None, None, None)


# ---------------------------------------------------------------------------------- LARS
class LarsState:
    """Device-side pointer tables for vipant_lars_step (cvap/module/lars.py:43-72); built once per optimizer.

    The per-step inputs -- gradient pointers and learning rates -- reach the device through two alternating PINNED host
    buffers and stream-ordered copies: the host never waits for the stream here (a `torch.tensor(list, device=...)` per step
    is a pageable copy, i.e. a stream synchronisation in the middle of the training loop)."""

    def __init__(self, params: Sequence[torch.Tensor], adapt: Sequence[bool]):
        self.params = list(params)
        dev = self.params[0].device
        n = len(self.params)
        self.mu = [torch.zeros_like(p) for p in self.params]
        self.n = torch.tensor([p.numel() for p in self.params], dtype=I64, device=dev)
        self.adapt = torch.tensor([int(a) for a in adapt], dtype=torch.int32, device=dev)
        self.p_ptrs = torch.tensor([p.data_ptr() for p in self.params], dtype=I64, device=dev)
        self.mu_ptrs = torch.tensor([m.data_ptr() for m in self.mu], dtype=I64, device=dev)
        self.ws = torch.empty((query("vipant_lars_workspace_bytes", n),), dtype=torch.uint8, device=dev)
        # gradient pointers (n x int64) and learning rates (n x fp32) in ONE staging buffer: one host-to-device copy per step
        self.stage = torch.empty((12 * n,), dtype=torch.uint8, device=dev)
        self.g_ptrs = self.stage[:8 * n].view(I64)
        self.lr = self.stage[8 * n:].view(F32)
        self._host = [(torch.empty((12 * n,), dtype=torch.uint8).pin_memory(), None) for _ in range(2)] if dev.type == "cuda" else None
        self._turn = 0

    def step(self, grads: Sequence[torch.Tensor], lrs: Sequence[float], weight_decay: float, momentum: float, eta: float):
        for p, g in zip(self.params, grads):
            assert g.is_contiguous() and g.dtype == F32 and g.shape == p.shape
        hb, ev = self._host[self._turn]
        if ev is not None:
            ev.synchronize()             # the copy issued from this buffer two steps ago (long done)
        n = len(self.params)
        hb[:8 * n].view(I64).copy_(torch.tensor([g.data_ptr() for g in grads], dtype=I64))
        hb[8 * n:].view(F32).copy_(torch.tensor(list(lrs), dtype=F32))
        self.stage.copy_(hb, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._host[self._turn] = (hb, ev)
        self._turn ^= 1
        call("vipant_lars_step", self.p_ptrs.data_ptr(), self.g_ptrs.data_ptr(), self.mu_ptrs.data_ptr(), self.n.data_ptr(),
             self.adapt.data_ptr(), self.lr.data_ptr(), len(self.params), float(weight_decay), float(momentum), float(eta),
             self.ws.data_ptr(), self.ws.numel(), _stream())
