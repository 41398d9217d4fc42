"""Per-kernel parity on the MI355X: every C-ABI entry point against a plain fp32/fp64 PyTorch
restatement of the same op on identical (bf16-rounded) inputs.  Tolerances are written next to each check:
bf16 outputs carry 2^-9 relative rounding, fp32-accumulated contractions of K terms ~1e-3 of the output scale.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gen  # noqa: E402  (tests/golden on sys.path via conftest)
import probe_lib  # noqa: E402  (tools/ on sys.path via conftest; NOT the product library)


@pytest.fixture(scope="module")
def ops():
    from vipant_amd import _ffi, ops as O
    _ffi.call("vipant_device_check")
    return O


DEV = "cuda:0"


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float32):
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def assert_close(got, ref, rtol, atol, what=""):
    got, ref = got.double().cpu(), ref.double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} mismatches, max err {err.max():.4e} "
                             f"(ref scale {ref.abs().max():.3e}); first at {idx}: got {got[tuple(idx)]:.6f} ref {ref[tuple(idx)]:.6f}")


# ------------------------------------------------------------------------------------------ GEMM NT
@pytest.mark.parametrize("M,N,K", [(100, 64, 192), (257, 2304, 128), (129, 520, 128), (40000, 256, 192)])
def test_gemm_nt_pingpong_edges(ops, M, N, K):
    """Corners of the ping-pong kernel's K-tile stream: two / three K-tiles per tile (the stream wraps into the next tile inside
    the look-ahead), fewer rows than one wave group, a single tile, N tails, many tiles per workgroup."""
    a = rnd(M, K, seed=41, dtype=torch.bfloat16); b = rnd(N, K, seed=42, dtype=torch.bfloat16, scale=K ** -0.5)
    bias = rnd(N, seed=43)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(a, b, c, bias=bias, epi=ops.EPI_BF16)
    assert_close(c, a.float() @ b.float().t() + bias, 1e-2, 2e-2, "bf16 epilogue")
    c2 = torch.empty_like(c)
    ops.gemm_nt(a, b, c2, bias=bias, epi=ops.EPI_BF16)
    assert torch.equal(c, c2)


def test_weight_casts(ops):
    """One multi-tensor launch == the per-matrix cast (bf16 copy and transposed bf16 copy), and the bf16 -> fp32 widening."""
    ws = [rnd(768, 768, seed=1), rnd(2304, 768, seed=2), rnd(100, 36, seed=3), rnd(64, 3072, seed=4), rnd(7, 5, seed=5)]
    wb, wt = ops.cast_weights(ws)
    for w, b1, t1 in zip(ws, wb, wt):
        b0, t0 = ops.cast_bf16(w, True)
        assert torch.equal(b0, b1) and torch.equal(t0, t1), tuple(w.shape)
        assert torch.equal(t1, w.to(torch.bfloat16).t().contiguous())
    x = rnd(1000, 768, seed=6, dtype=torch.bfloat16)
    y = torch.empty(1000, 768, device=DEV)
    ops.call("vipant_cast_f32", x.data_ptr(), y.data_ptr(), x.numel(), torch.cuda.current_stream().cuda_stream)
    assert torch.equal(y, x.float())


@pytest.mark.parametrize("M,N,K", [(992, 768, 768), (256, 256, 64), (1000, 2304, 768), (516, 512, 1536), (4096, 3072, 768),
                                   (33, 768, 3072)])
def test_gemm_nt_plain(ops, M, N, K):
    a = rnd(M, K, seed=1, dtype=torch.bfloat16); b = rnd(N, K, seed=2, dtype=torch.bfloat16, scale=K ** -0.5)
    bias = rnd(N, seed=3)
    ref = a.float() @ b.float().t() + bias
    c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(a, b, c, bias=bias, epi=ops.EPI_BF16)
    assert_close(c, ref, 1e-2, 2e-2, "bf16 epilogue")
    c32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_nt(a, b, c32, bias=bias, epi=ops.EPI_F32)
    assert_close(c32, ref, 1e-4, 2e-3, "f32 epilogue")
    ops.gemm_nt(a, b, c32, epi=ops.EPI_SCALE_F32, alpha=0.37)
    assert_close(c32, 0.37 * (a.float() @ b.float().t()), 1e-4, 2e-3, "scale epilogue")


@pytest.mark.parametrize("M,N,K", [(23117, 768, 1024),     # 91 x 3 tiles = 1 round + 17; 77 rows in the last panel
                                   (70000, 768, 2304),     # 3 rounds + 54; last panel 112 rows
                                   (8000, 2304, 768),      # 32 x 9 = 1 round + 32
                                   (65536, 256, 1024)])    # exactly one round
def test_gemm_nt_short_last_round(ops, M, N, K):
    """The DEEP ping-pong schedule with a short last round of the persistent walk and a ragged last row panel: the reference
    product, nothing written past the last row, and independence of where a row falls in the walk (written for the half-tile
    split of the last round, profiles/r3_gemm_experiments.md; the split was not kept, the test is)."""
    a, b = rnd(M, K, seed=31, dtype=torch.bfloat16), rnd(N, K, seed=32, dtype=torch.bfloat16, scale=0.05)
    bias = rnd(N, seed=33)
    c = torch.full((M + 1, N), 7.0, dtype=torch.bfloat16, device=DEV)        # one guard row behind the output
    ops.gemm_nt(a, b, c[:M], bias=bias, epi=ops.EPI_BF16)
    assert bool((c[M] == 7.0).all()), "wrote past the last row"
    for lo in range(0, M, 16384):
        hi = min(M, lo + 16384)
        assert_close(c[lo:hi], a[lo:hi].float() @ b.float().t() + bias, 1e-2, 2e-2, f"rows {lo}..{hi}")
    # a row-shifted problem changes which tile and round a row falls into: the shared rows must agree bit for bit
    c2 = torch.empty(M - 256, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(a[256:], b, c2, bias=bias, epi=ops.EPI_BF16)
    assert torch.equal(c2, c[256:M])


def test_gemm_nt_identity_layout(ops):
    """A = I with an asymmetric B catches a transposed / permuted C write."""
    K = 256
    a = torch.eye(K, device=DEV, dtype=torch.bfloat16)
    b = (torch.arange(512 * K, device=DEV, dtype=torch.float32).reshape(512, K) % 251 - 125).to(torch.bfloat16)
    c = torch.empty(K, 512, dtype=torch.float32, device=DEV)
    ops.gemm_nt(a, b, c, epi=ops.EPI_F32)
    assert_close(c, b.float().t(), 0, 0, "identity")


def test_gemm_nt_fused_epilogues(ops):
    M, N, K = 1000, 3072, 768
    a = rnd(M, K, seed=1, dtype=torch.bfloat16); b = rnd(N, K, seed=2, dtype=torch.bfloat16, scale=K ** -0.5)
    bias = rnd(N, seed=3)
    pre = a.float() @ b.float().t() + bias
    u = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); g = torch.empty_like(u)
    ops.gemm_nt(a, b, g, bias=bias, aux=u, epi=ops.EPI_QUICKGELU)
    assert_close(u, pre, 1e-2, 2e-2, "quickgelu.u")
    assert_close(g, pre * torch.sigmoid(1.702 * pre), 1e-2, 2e-2, "quickgelu.g")
    # residual, in place
    M, N, K = 992, 768, 3072
    a = rnd(M, K, seed=4, dtype=torch.bfloat16); b = rnd(N, K, seed=5, dtype=torch.bfloat16, scale=K ** -0.5)
    bias = rnd(N, seed=6); res = rnd(M, N, seed=7)
    ref = a.float() @ b.float().t() + bias + res
    out = torch.empty(M, N, device=DEV)
    ops.gemm_nt(a, b, out, bias=bias, aux=res, epi=ops.EPI_RESIDUAL_F32)
    assert_close(out, ref, 1e-4, 3e-3, "residual")
    r2 = res.clone()
    ops.gemm_nt(a, b, r2, bias=bias, aux=r2, epi=ops.EPI_RESIDUAL_F32)
    assert_close(r2, ref, 1e-4, 3e-3, "residual in place")
    # dQuickGELU
    M, N, K = 700, 3072, 768
    a = rnd(M, K, seed=8, dtype=torch.bfloat16); b = rnd(N, K, seed=9, dtype=torch.bfloat16, scale=K ** -0.5)
    uu = rnd(M, N, seed=10, dtype=torch.bfloat16, scale=2.0)
    acc = a.float() @ b.float().t()
    sg = torch.sigmoid(1.702 * uu.float())
    ref = acc * (sg * (1 + 1.702 * uu.float() * (1 - sg)))
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(a, b, out, aux=uu, epi=ops.EPI_DQUICKGELU)
    assert_close(out, ref, 1e-2, 2e-2, "dquickgelu")


def qgelu_prime(u):
    sg = torch.sigmoid(1.702 * u)
    return sg * (1 + 1.702 * u * (1 - sg))


@pytest.mark.parametrize("M,N,K", [(1000, 3072, 768), (300, 264, 128), (70000, 1024, 256),
                                   (33100, 3072, 128)])     # 130 x 12 tiles: the column-grouped walk with ragged row quarters
def test_gemm_nt_quickgelu_derivative_code(ops, M, N, K):
    """The two-output epilogues with the 8-bit QuickGELU' code (what the step uses): g as before; the code decodes to
    QuickGELU'(pre-activation) within half a code step (2.4e-3) plus the bf16 rounding of the pre-activation; the backward
    epilogue multiplies by the decoded derivative."""
    a = rnd(M, K, seed=31, dtype=torch.bfloat16); b = rnd(N, K, seed=32, dtype=torch.bfloat16, scale=2.0 * K ** -0.5)
    bias = rnd(N, seed=33)
    pre = a.float() @ b.float().t() + bias
    code = torch.empty(M, N, dtype=torch.uint8, device=DEV); g = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(a, b, g, bias=bias, aux=code, epi=ops.EPI_QUICKGELU_D8)
    assert_close(g, pre * torch.sigmoid(1.702 * pre), 1e-2, 2e-2, "quickgelu.g")
    dec = code.float() / 212.5 - 0.1
    assert float((dec - qgelu_prime(pre)).abs().max()) < 2.4e-3 + 6e-3, float((dec - qgelu_prime(pre)).abs().max())
    assert float((dec - qgelu_prime(pre)).abs().mean()) < 2e-3
    acc = a.float() @ b.float().t()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(a, b, out, aux=code, epi=ops.EPI_DQUICKGELU_D8)
    assert_close(out, acc * dec, 1e-2, 2e-2, "dquickgelu from the code")
    with pytest.raises(Exception):          # K = 64 has no ping-pong kernel: the code epilogues refuse instead of falling back
        ops.gemm_nt(a[:, :64], b[:, :64], g, bias=bias, aux=code, epi=ops.EPI_QUICKGELU_D8)


@pytest.mark.parametrize("M,N,K", [(512, 768, 768), (512, 3072, 768), (512, 768, 3072), (33, 768, 3072), (1000, 264, 128), (1, 64, 64),
                                   (4096, 520, 448)])
def test_gemm_nt_few_rows(ops, M, N, K):
    """VIPANT_EPI_FEW_ROWS (the last block on its read-out rows, the read-out projection): 64 x 64 tiles, K split over the four
    waves of a workgroup.  Every epilogue the flag accepts, against fp32 torch AND against the 256 x 256 kernels on the same
    operands (same products, another summation order: bf16 outputs within one rounding step); a row's result does not depend on
    the rows that travel with it (what `running.micro_batch` relies on)."""
    a = rnd(M, K, seed=51, dtype=torch.bfloat16); b = rnd(N, K, seed=52, dtype=torch.bfloat16, scale=2.0 * K ** -0.5)
    bias = rnd(N, seed=53); res = rnd(M, N, seed=54)
    acc = a.float() @ b.float().t()
    pre = acc + bias

    def run(few):
        o = {}
        o["bf16"] = ops.gemm_nt(a, b, torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV), bias=bias, epi=ops.EPI_BF16, few_rows=few)
        o["f32"] = ops.gemm_nt(a, b, torch.full((M, N), 7.0, device=DEV), bias=bias, epi=ops.EPI_F32, few_rows=few)
        o["f32_nobias"] = ops.gemm_nt(a, b, torch.full((M, N), 7.0, device=DEV), epi=ops.EPI_F32, few_rows=few)
        o["res"] = ops.gemm_nt(a, b, torch.full((M, N), 7.0, device=DEV), bias=bias, aux=res, epi=ops.EPI_RESIDUAL_F32, few_rows=few)
        r2 = res.clone()
        o["res_inplace"] = ops.gemm_nt(a, b, r2, bias=bias, aux=r2, epi=ops.EPI_RESIDUAL_F32, few_rows=few)
        if N % 8 == 0 and K >= 128:
            code = torch.full((M, N), 7, dtype=torch.uint8, device=DEV)
            o["g"] = ops.gemm_nt(a, b, torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV), bias=bias, aux=code,
                                 epi=ops.EPI_QUICKGELU_D8, few_rows=few)
            o["code"] = code
            o["dg"] = ops.gemm_nt(a, b, torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV), aux=code, epi=ops.EPI_DQUICKGELU_D8,
                                  few_rows=few)
        return o

    few, big = run(True), run(False)
    if M > 40:          # the same rows inside a smaller launch: bit for bit
        sub = ops.gemm_nt(a[7:40], b, torch.empty((33, N), dtype=torch.bfloat16, device=DEV), bias=bias, epi=ops.EPI_BF16, few_rows=True)
        assert torch.equal(sub, few["bf16"][7:40])
    with pytest.raises(Exception):
        ops.gemm_nt(a, b, torch.empty((M, N), device=DEV), epi=ops.EPI_SCALE_F32, alpha=0.5, few_rows=True)
    assert_close(few["bf16"], pre, 1e-2, 2e-2, "bf16")
    assert_close(few["f32"], pre, 1e-4, 3e-3, "f32")
    assert_close(few["f32_nobias"], acc, 1e-4, 3e-3, "f32, no bias")
    assert_close(few["res"], pre + res, 1e-4, 3e-3, "residual")
    assert torch.equal(few["res"], few["res_inplace"])
    for k in ("f32", "f32_nobias", "res"):
        assert_close(few[k], big[k], 1e-5, 1e-5 * float(pre.abs().max()) + 1e-6, f"{k} vs the 256 x 256 kernel")
    assert_close(few["bf16"], big["bf16"], 2 ** -7, 1e-3, "bf16 vs the 256 x 256 kernel")
    if "g" in few:
        assert_close(few["g"], pre * torch.sigmoid(1.702 * pre), 1e-2, 2e-2, "quickgelu.g")
        dec = few["code"].float() / 212.5 - 0.1
        assert float((dec - qgelu_prime(pre)).abs().max()) < 2.4e-3 + 6e-3
        assert float((few["code"].int() - big["code"].int()).abs().max()) <= 2      # one code step + a flipped bf16 rounding of u
        assert_close(few["dg"], acc * dec, 1e-2, 2e-2, "dquickgelu from the code")


@pytest.mark.parametrize("M,N,K", [(4100, 2304, 768), (5000, 200, 64), (4096, 3072, 128), (6001, 776, 1024),
                                   (20000, 1000, 128), (66000, 256, 64)])   # >= 256 tiles with a short last round: half-tile tail kernel
def test_gemm_nt_short_k_large_m(ops, M, N, K):
    """Short K, many row tiles per workgroup (the persistent tile walk): ragged M / N tails and all three staged epilogues."""
    a = rnd(M, K, seed=21, dtype=torch.bfloat16); b = rnd(N, K, seed=22, dtype=torch.bfloat16, scale=K ** -0.5)
    bias = rnd(N, seed=23)
    pre = a.float() @ b.float().t() + bias
    c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(a, b, c, bias=bias, epi=ops.EPI_BF16)
    assert_close(c, pre, 1e-2, 2e-2, "bf16 epilogue")
    u = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); g = torch.empty_like(u)
    ops.gemm_nt(a, b, g, bias=bias, aux=u, epi=ops.EPI_QUICKGELU)
    assert_close(u, pre, 1e-2, 2e-2, "quickgelu.u")
    assert_close(g, pre * torch.sigmoid(1.702 * pre), 1e-2, 2e-2, "quickgelu.g")
    uu = rnd(M, N, seed=24, dtype=torch.bfloat16, scale=2.0)
    acc = a.float() @ b.float().t()
    sg = torch.sigmoid(1.702 * uu.float())
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(a, b, out, aux=uu, epi=ops.EPI_DQUICKGELU)
    assert_close(out, acc * (sg * (1 + 1.702 * uu.float() * (1 - sg))), 1e-2, 2e-2, "dquickgelu")


def test_gemm_nt_strided_rows(ops):
    """cls-row read-out: A rows taken with a large row stride."""
    S, D = 31, 768
    x = rnd(8 * S, D, seed=11, dtype=torch.bfloat16)
    a = x.view(8, S * D)[:, :D]
    b = rnd(512, D, seed=12, dtype=torch.bfloat16, scale=D ** -0.5)
    c = torch.empty(8, 512, device=DEV)
    ops.gemm_nt(a, b, c, epi=ops.EPI_F32)
    assert_close(c, a.float() @ b.float().t(), 1e-4, 2e-3, "strided A")


# ------------------------------------------------------------------------------------------ GEMM TN
@pytest.mark.parametrize("M,P,Q", [(992, 768, 768), (64, 256, 256), (100, 16, 512), (5000, 2304, 768), (4096, 768, 3072),
                                   (129, 129 // 8 * 8, 512)])
def test_gemm_tn(ops, M, P, Q):
    a = rnd(M, P, seed=1, dtype=torch.bfloat16); b = rnd(M, Q, seed=2, dtype=torch.bfloat16)
    ref = a.float().t() @ b.float()
    c = torch.empty(P, Q, device=DEV)
    ops.gemm_tn(a, b, c)
    assert_close(c, ref, 1e-4, 2e-3 * math.sqrt(M), "tn")
    c2 = torch.full((P, Q), 1.5, device=DEV)
    cs = torch.full((P,), 0.25, device=DEV)
    ops.gemm_tn(a, b, c2, accumulate=True, a_colsum=cs)
    assert_close(c2, ref + 1.5, 1e-4, 2e-3 * math.sqrt(M), "tn accumulate")
    assert_close(cs, a.float().sum(0) + 0.25, 1e-5, 1e-3 * math.sqrt(M), "tn fused column sums (accumulate)")
    cs2 = torch.empty(P, device=DEV)
    ops.gemm_tn(a, b, c, a_colsum=cs2)
    assert_close(cs2, a.float().sum(0), 1e-5, 1e-3 * math.sqrt(M), "tn fused column sums")
    assert_close(c, ref, 1e-4, 2e-3 * math.sqrt(M), "tn (with column sums)")


def test_gemm_tn_layout(ops):
    """One-hot rows: C[p, q] = B[m(p), q] catches any permutation inside the transposed reads."""
    M, P, Q = 256, 256, 256
    perm = torch.randperm(P, generator=torch.Generator().manual_seed(3))
    a = torch.zeros(M, P, dtype=torch.bfloat16, device=DEV)
    a[torch.arange(M), perm.to(DEV)] = 1.0
    b = (torch.arange(M * Q, device=DEV, dtype=torch.float32).reshape(M, Q) % 253 - 126).to(torch.bfloat16)
    c = torch.empty(P, Q, device=DEV)
    ops.gemm_tn(a, b, c)
    assert_close(c, a.float().t() @ b.float(), 0, 0, "tn one-hot")


def test_colsum(ops):
    for M, N in ((992, 768), (5000, 3072), (3, 2304)):
        x = rnd(M, N, seed=1, dtype=torch.bfloat16)
        out = torch.empty(N, device=DEV)
        ops.colsum(x, out)
        assert_close(out, x.float().sum(0), 1e-5, 1e-3 * math.sqrt(M), "colsum")


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M,D", [(992, 768), (77 * 3, 512), (5, 1024), (4100, 768)])
def test_layernorm(ops, M, D):
    x = rnd(M, D, seed=1, scale=2.0) + 0.3
    gm = rnd(D, seed=2, scale=0.1) + 1.0; bt = rnd(D, seed=3, scale=0.1)
    y, y32, mean, rstd = ops.layernorm_fwd(x, gm, bt, want_f32=True)
    xr = x.double().requires_grad_()
    gmr, btr = gm.double().requires_grad_(), bt.double().requires_grad_()
    ref = torch.nn.functional.layer_norm(xr, (D,), gmr, btr, 1e-5)
    assert_close(y32, ref, 1e-5, 1e-5, "ln fwd f32")
    assert_close(y, ref, 1e-2, 1e-2, "ln fwd bf16")
    dy = rnd(M, D, seed=4, dtype=torch.bfloat16)
    dres = rnd(M, D, seed=5)
    ref.backward(dy.double())
    dx = torch.empty(M, D, device=DEV); dxb = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    dg = torch.empty(D, device=DEV); db = torch.empty(D, device=DEV)
    dcs = torch.empty(D, device=DEV)
    ops.layernorm_bwd(dy, x, mean, rstd, gm, dres=dres, dx=dx, dx_bf16=dxb, dgamma=dg, dbeta=db, dx_colsum=dcs)
    assert_close(dcs, (xr.grad + dres.double()).sum(0), 1e-4, 1e-3 * math.sqrt(M), "ln bwd dx colsum")
    assert_close(dx, xr.grad + dres.double(), 1e-4, 1e-4, "ln bwd dx")
    assert_close(dxb, xr.grad + dres.double(), 1e-2, 1e-2, "ln bwd dx bf16")
    assert_close(dg, gmr.grad, 1e-4, 1e-3 * math.sqrt(M), "ln bwd dgamma")
    assert_close(db, btr.grad, 1e-4, 1e-3 * math.sqrt(M), "ln bwd dbeta")
    # fp32 dy, no residual, in-place capable
    dy32 = rnd(M, D, seed=6)
    xr.grad = None
    torch.nn.functional.layer_norm(xr, (D,), gmr, btr, 1e-5).backward(dy32.double())
    ops.layernorm_bwd(dy32, x, mean, rstd, gm, dx=dx, dgamma=dg, dbeta=db)
    assert_close(dx, xr.grad, 1e-4, 1e-4, "ln bwd dx (f32 dy)")


def test_layernorm_fp16_stream(ops):
    """`running.stream_dtype: fp16`: the stream read / written as fp16, statistics in fp32, the norm taken on the unrounded sum."""
    M, D = 300, 768
    x = (rnd(M, D, seed=11) * 3).to(torch.float16)
    add = rnd(M, D, seed=12).to(torch.bfloat16)
    gm, bt = rnd(D, seed=13) * 0.1 + 1.0, rnd(D, seed=14) * 0.1
    h, _, mean, rstd, xs = ops.layernorm_fwd(x, gm, bt, add=add, want_sum=True, sum_f16=True)
    assert xs.dtype == torch.float16
    s64 = x.double() + add.double()
    assert torch.equal(xs, s64.float().to(torch.float16)), "the stored stream is the fp16 rounding of the exact sum"
    ref = torch.nn.functional.layer_norm(s64, (D,), gm.double(), bt.double(), 1e-5)
    assert_close(h, ref, 1e-2, 1e-2, "ln fwd (fp16 stream)")
    assert_close(mean, s64.mean(-1), 1e-5, 1e-5, "ln mean (fp16 stream)")
    # fp16 in, fp32 out and fp32 in, fp16 out (the first / last pass of a stack)
    xs32 = ops.layernorm_fwd(x, gm, bt, add=add, want_sum=True)[4]
    assert xs32.dtype == torch.float32 and torch.equal(xs32, s64.float())
    xs16 = ops.layernorm_fwd(x.float(), gm, bt, add=add, want_sum=True, sum_f16=True)[4]
    assert torch.equal(xs16, xs)
    assert_close(ops.residual_add(x, add), s64, 1e-6, 1e-6, "residual_add (fp16 stream)")
    # backward on the saved fp16 rows
    xr = xs.double().requires_grad_()
    dy = rnd(M, D, seed=15, dtype=torch.bfloat16)
    gmr = gm.double().requires_grad_()
    torch.nn.functional.layer_norm(xr, (D,), gmr, bt.double(), 1e-5).backward(dy.double())
    _, _, mean2, rstd2 = ops.layernorm_fwd(xs, gm, bt)
    dx = torch.empty(M, D, device=DEV); dg = torch.empty(D, device=DEV); db = torch.empty(D, device=DEV)
    ops.layernorm_bwd(dy, xs, mean2, rstd2, gm, dx=dx, dgamma=dg, dbeta=db)
    assert_close(dx, xr.grad, 1e-4, 1e-4, "ln bwd dx (fp16 rows)")
    assert_close(dg, gmr.grad, 1e-4, 1e-3 * math.sqrt(M), "ln bwd dgamma (fp16 rows)")


# ------------------------------------------------------------------------------------------ attention
def ref_attention(qkv, batch, S, H, causal):
    D = H * 64
    q, k, v = qkv.double().view(batch, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    if causal:
        s = s + torch.full((S, S), float("-inf"), device=s.device, dtype=s.dtype).triu_(1)
    p = torch.softmax(s, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(batch * S, D)
    return o, torch.logsumexp(s, -1)


@pytest.mark.parametrize("batch,S,H,causal", [(2, 31, 12, False), (3, 316, 12, False), (2, 77, 8, True), (2, 50, 12, False),
                                              (1, 16, 1, False), (2, 306, 12, False), (3, 20, 8, True),
                                              # S > 384: streaming kernels (YAML-default stride [16,16] at T=1000 -> 428)
                                              (2, 428, 12, False), (1, 645, 4, False), (2, 450, 2, True), (1, 1213, 2, False)])
def test_mha(ops, batch, S, H, causal):
    D = H * 64
    qkv = rnd(batch * S, 3 * D, seed=1, dtype=torch.bfloat16, scale=1.5)
    out, lse = ops.mha_fwd(qkv, batch, S, H, causal)
    qr = qkv.double().requires_grad_()
    ref, rlse = ref_attention(qr, batch, S, H, causal)
    assert_close(out, ref, 1e-2, 1e-2, "mha fwd")
    assert_close(lse, rlse, 1e-4, 1e-3, "mha lse")
    dout = rnd(batch * S, D, seed=2, dtype=torch.bfloat16)
    ref.backward(dout.double())
    dqkv = ops.mha_bwd(qkv, out, dout, lse, batch, S, H, causal)
    scale = qr.grad.abs().max().item()
    assert_close(dqkv, qr.grad, 2e-2, 2e-2 * scale, "mha bwd")


@pytest.mark.parametrize("batch,H,S", [(60, 12, 316), (171, 3, 306), (43, 12, 300)])
def test_mha_bwd_ticket_walk_equals_static_walk(ops, monkeypatch, batch, H, S):
    """Round 5: the persistent attention backward (mha_bwd1s_kernel, one workgroup per CU) draws its third and later problems from the
    stream's ticket counter when batch * heads > 2 x CUs (720, 513 and 516 problems here: long queues, a single drawn problem, four).
    Which workgroup computes a problem must not matter: bit-identical to the static stride (VIPANT_GEMM_VARIANT bit 22), alone and with
    CUs held by the probe kernel (tools/probes/comm_shadow.hip) on a second stream, repeatedly (the counters must be back at zero for every launch, also when a
    ticket-walk NT contraction runs in between on the same stream)."""
    D = H * 64
    qkv = rnd(batch * S, 3 * D, seed=11, dtype=torch.bfloat16, scale=1.5)
    dout = rnd(batch * S, D, seed=12, dtype=torch.bfloat16)
    out, lse = ops.mha_fwd(qkv, batch, S, H, False)
    src = torch.empty(28 << 20, dtype=torch.uint8, device=DEV); dst = torch.empty_like(src)
    side = torch.cuda.Stream()
    a = rnd(40448, 768, seed=13, dtype=torch.bfloat16); w = rnd(768, 768, seed=14, dtype=torch.bfloat16, scale=0.03)
    c = torch.empty(40448, 768, dtype=torch.bfloat16, device=DEV)
    monkeypatch.setenv("VIPANT_GEMM_VARIANT", "4194304")
    ref = ops.mha_bwd(qkv, out, dout, lse, batch, S, H, False)
    assert torch.isfinite(ref.float()).all()
    monkeypatch.setenv("VIPANT_GEMM_VARIANT", "0")
    for held, us in [(0, 0.0), (64, 300.0), (0, 0.0), (24, 2000.0), (0, 0.0)]:
        if held:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                probe_lib.comm_shadow(src.data_ptr(), dst.data_ptr(), src.numel(), held, us, side.cuda_stream)
        else:
            ops.gemm_nt(a, w, c, epi=ops.EPI_BF16)
        got = ops.mha_bwd(qkv, out, dout, lse, batch, S, H, False)
        assert torch.equal(ref, got), (held, us)
    torch.cuda.synchronize()


@pytest.mark.parametrize("key,gain", [(7, 40), (200, 40), (200, 250), (7, 250), (315, 250), (170, 120)])
def test_mha_spiky_scores(ops, key, gain):
    """A dominant key per query (the softmax must not lose it to rounding / masking): keys in the first and the second half of a
    query block's key range, with a lead of 18 in the exponent (gain 40) and of more than 64 (gain 250: a kernel that exponentiates
    against a partial maximum -- the wide forward kept as text under tools/probes/ did -- has to rescale), on whole query blocks
    (query 100) and on the block two waves share (query 300)."""
    batch, S, H = 1, 316, 1
    qkv = rnd(S, 192, seed=3, dtype=torch.bfloat16, scale=0.2)
    qkv[:, 64:128][key] = qkv[:, :64][100] * gain       # key aligned with query 100
    qkv[:, :64][300] = qkv[:, :64][100]                 # ... and with query 300 (last query block)
    out, lse = ops.mha_fwd(qkv, batch, S, H, False)
    ref, rlse = ref_attention(qkv, batch, S, H, False)
    assert_close(out, ref, 1e-2, 1e-2, "mha spiky")
    assert_close(lse, rlse, 1e-4, 1e-3, "mha spiky lse")


@pytest.mark.parametrize("M,P,Q", [(4096, 4096, 512), (1000, 520, 64), (129, 256, 512)])
def test_gemm_tn_pair(ops, M, P, Q):
    """Two token-reduction contractions of one shape in one launch (the InfoNCE feature gradients): each equals the single launch
    of the same operands up to the split-K grouping, and the fp32 product."""
    a0, a1 = rnd(M, P, seed=41, dtype=torch.bfloat16), rnd(M, P, seed=42, dtype=torch.bfloat16)
    b0, b1 = rnd(M, Q, seed=43, dtype=torch.bfloat16), rnd(M, Q, seed=44, dtype=torch.bfloat16)
    c0, c1 = torch.empty(P, Q, device=DEV), torch.empty(P, Q, device=DEV)
    ws = torch.empty(ops.query("vipant_gemm_tn_pair_workspace_bytes", M, P, Q) + 256, dtype=torch.uint8, device=DEV)
    ops.call("vipant_gemm_tn_pair", a0.data_ptr(), b0.data_ptr(), c0.data_ptr(), a1.data_ptr(), b1.data_ptr(), c1.data_ptr(), P, Q, Q,
             M, P, Q, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    for a, b, c in ((a0, b0, c0), (a1, b1, c1)):
        ref = a.float().t() @ b.float()
        assert_close(c, ref, 1e-3, 2e-3 * float(ref.abs().max()), "tn pair")
        assert_close(c, ops.gemm_tn(a, b, torch.empty(P, Q, device=DEV)), 1e-5, 1e-5 * float(ref.abs().max()), "tn pair vs single")


@pytest.mark.parametrize("batch,S,H,causal,with_idx", [(3, 316, 12, False, False), (4, 77, 8, True, True), (2, 50, 12, False, True),
                                                       (5, 17, 2, True, False), (2, 645, 4, False, False)])
def test_mha_rows(ops, batch, S, H, causal, with_idx):
    """One query per (item, head) -- the last block on its read-out rows (csrc/readout_rows.hip) -- against the full attention of
    the reference op (cvap/module/val.py:511-517) evaluated in fp64 and read at those rows: output, softmax rows, dq, dK, dV."""
    D = H * 64
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    idx = torch.randint(0, S, (batch,), generator=g).to(DEV) if with_idx else None
    if with_idx and causal:
        idx[0] = S - 1; idx[1] = 0          # the whole sequence / a single key
    r = (torch.arange(batch, device=DEV) * S + (idx if idx is not None else 0))
    qkv = rnd(batch * S, 3 * D, seed=1, dtype=torch.bfloat16, scale=1.5)
    q_rows = qkv[r, :D].contiguous()
    out = torch.empty(batch, D, dtype=torch.bfloat16, device=DEV)
    probs = torch.empty(batch, H, S, dtype=torch.float32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    ops.call("vipant_mha_rows_fwd", q_rows.data_ptr(), qkv.data_ptr(), idx.data_ptr() if idx is not None else None, out.data_ptr(),
             probs.data_ptr(), batch, S, H, int(causal), st)
    qr = qkv.double().requires_grad_()
    ref, _ = ref_attention(qr, batch, S, H, causal)
    assert_close(out, ref[r], 1e-2, 1e-2, "mha_rows fwd")
    assert_close(probs.sum(-1), torch.ones(batch, H), 1e-5, 1e-5, "softmax rows sum to one")
    dout_rows = rnd(batch, D, seed=2, dtype=torch.bfloat16)
    dfull = torch.zeros(batch * S, D, dtype=torch.float64, device=DEV)
    dfull[r] = dout_rows.double()
    ref.backward(dfull)
    dq = torch.empty(batch, D, dtype=torch.bfloat16, device=DEV)
    dqkv = torch.full((batch * S, 3 * D), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.call("vipant_mha_rows_bwd", q_rows.data_ptr(), qkv.data_ptr(), idx.data_ptr() if idx is not None else None, probs.data_ptr(),
             dout_rows.data_ptr(), dq.data_ptr(), dqkv.data_ptr(), batch, S, H, int(causal), st)
    scale = qr.grad.abs().max().item()
    assert_close(dq, qr.grad[r, :D], 2e-2, 2e-2 * scale, "mha_rows dq")
    assert_close(dqkv[:, D:], qr.grad[:, D:], 2e-2, 2e-2 * scale, "mha_rows dK | dV")
    assert bool((dqkv[:, :D] == 7.0).all()), "the Q column block is not this kernel's to write"
    mask = torch.ones(batch * S, dtype=torch.bool, device=DEV); mask[r] = False
    assert float(qr.grad[:, :D][mask].abs().max()) == 0.0        # the premise: no query gradient off the read-out rows


@pytest.mark.parametrize("pair", [1, 0])
@pytest.mark.parametrize("batch,S,H,causal,with_idx", [(3, 316, 12, False, False), (4, 77, 8, True, True), (2, 50, 12, False, True),
                                                       (3, 257, 16, False, False), (2, 5, 8, True, True), (2, 645, 12, False, False),
                                                       (3, 77, 8, True, False),          # causal, no row index: row 0, one key
                                                       (1, 1024, 16, False, False)])     # the longest item the kernels take (LDS)
def test_rows_ctx(ops, batch, S, H, causal, with_idx, pair):
    """The one-query attention of the last block with the K / V projection folded into the query side (csrc/readout_ctx.hip) against
    its definition in fp64 autograd: contexts, softmax rows, dh1 of every token, dqk.  pair = 1 (the step's form, round 5): qk, the
    contexts and dqk are bf16 pairs hi + lo -- the contexts and dqk must then be good to 2^-15 of their scale, not 2^-8; pair = 0:
    single bf16 planes."""
    D = H * 64
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    idx = torch.randint(0, S, (batch,), generator=g).to(DEV) if with_idx else None
    if with_idx and causal:
        idx[0] = S - 1; idx[1] = 0          # the whole sequence / a single key
    def planes(x32):           # fp32 -> the operand as the kernels take it: [2, n, D] hi / lo planes, or one bf16 plane
        hi = x32.to(torch.bfloat16)
        return torch.stack([hi, (x32 - hi.float()).to(torch.bfloat16)]) if pair else hi

    def value(t):              # ... and back to one fp64 tensor
        return (t[0].double() + t[1].double()) if pair else t.double()

    shape = (2, batch * H, D) if pair else (batch * H, D)
    qk = planes(rnd(batch * H, D, seed=1, scale=0.35))
    h1 = rnd(batch * S, D, seed=2, dtype=torch.bfloat16)
    ctx = torch.full(shape, 7.0, dtype=torch.bfloat16, device=DEV)
    probs = torch.full((batch, H, S), 7.0, dtype=torch.float32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    ip = idx.data_ptr() if idx is not None else None
    ops.call("vipant_rows_ctx_fwd", qk.data_ptr(), h1.data_ptr(), ip, ctx.data_ptr(), probs.data_ptr(), batch, S, H, int(causal), pair, st)
    qd = value(qk).view(batch, H, D).requires_grad_()
    hd = h1.double().view(batch, S, D).requires_grad_()
    sc = torch.einsum("bhd,bsd->bhs", qd, hd) / 8.0
    if causal:
        lim = (idx if idx is not None else torch.zeros(batch, dtype=torch.long, device=DEV)).view(batch, 1, 1)
        sc = sc.masked_fill(torch.arange(S, device=DEV).view(1, 1, S) > lim, float("-inf"))
    pd = torch.softmax(sc, -1)
    ref = torch.einsum("bhs,bsd->bhd", pd, hd)
    assert_close(probs, pd, 2e-3, 1e-5, "rows_ctx softmax rows")
    # the probabilities enter the context product as bf16 (as in the reference-shaped attention kernels): 2^-9 each, averaged over the
    # keys; the pair removes the OUTPUT rounding, so its error budget is the product's alone
    assert_close(value(ctx).view(batch, H, D), ref, 3e-3 if pair else 1e-2, 3e-3 if pair else 1e-2, "rows_ctx contexts")
    dctx = rnd(batch * H, D, seed=3, dtype=torch.bfloat16)
    ref.backward(dctx.double().view(batch, H, D))
    dh1 = torch.full((batch * S, D), 7.0, dtype=torch.bfloat16, device=DEV)
    dqk = torch.full(shape, 7.0, dtype=torch.bfloat16, device=DEV)
    ops.call("vipant_rows_ctx_bwd", qk.data_ptr(), dctx.data_ptr(), ctx.data_ptr(), h1.data_ptr(), ip, probs.data_ptr(), dh1.data_ptr(),
             dqk.data_ptr(), None, batch, S, H, int(causal), pair, st)
    assert_close(dh1.view(batch, S, D), hd.grad, 2e-2, 2e-2 * hd.grad.abs().max().item(), "rows_ctx dh1")
    assert_close(value(dqk).view(batch, H, D), qd.grad, 2e-2, 2e-2 * max(qd.grad.abs().max().item(), 1e-3), "rows_ctx dqk")
    if pair:        # the lo plane is what the hi plane's rounding left: |lo| <= 2^-8 |hi| elementwise, and not identically zero
        one_key = causal and not with_idx        # softmax over one key: the context IS that bf16 row, dqk is exactly zero
        assert float((dqk[1].float().abs() - 2.0 ** -8 * dqk[0].float().abs()).max()) <= 1e-30
        assert float((ctx[1].float().abs() - 2.0 ** -8 * ctx[0].float().abs()).max()) <= 1e-30
        assert one_key or (float(dqk[1].float().abs().max()) > 0 and float(ctx[1].float().abs().max()) > 0)
    if S == 1024:
        with pytest.raises(Exception):      # one token more: refused (the caller takes the K / V form)
            ops.call("vipant_rows_ctx_fwd", qk.data_ptr(), h1.data_ptr(), ip, ctx.data_ptr(), probs.data_ptr(), batch, S + 1, H, int(causal), pair, st)
    if causal:          # rows behind the limit get exact zeros, not the fill
        lim = idx if idx is not None else torch.zeros(batch, dtype=torch.long, device=DEV)
        behind = torch.arange(S, device=DEV).view(1, S) > lim.view(batch, 1)
        assert float(dh1.view(batch, S, D)[behind].abs().max() if behind.any() else 0.0) == 0.0


@pytest.mark.parametrize("n,H", [(512, 12), (37, 8), (3, 16)])
def test_gemm_nt_heads(ops, n, H):
    """The per-head contractions of the folded last block in one launch each (vipant_gemm_nt_heads): a head's 64 columns of `n` rows
    against that head's block of a weight matrix and back -- against the block-sparse formulation they replace (head_expand +
    [n H, D] x [D, D] + head_extract) and against fp32 torch."""
    D = 64 * H
    rows = rnd(n, D, seed=1, dtype=torch.bfloat16)
    w = rnd(3 * D, D, seed=2, dtype=torch.bfloat16, scale=D ** -0.5)            # an in_proj_weight: rows D .. 2D are W_k
    wt = w.t().contiguous()                                                     # [D, 3D]
    bias = rnd(D, seed=3)
    wide = ops.heads_to_wide(rows, wt[:, D:2 * D], torch.full((n * H, D), 7.0, dtype=torch.bfloat16, device=DEV), H)
    ref = torch.einsum("nhc,hcd->nhd", rows.float().view(n, H, 64), w[D:2 * D].float().view(H, 64, D)).reshape(n * H, D)
    assert_close(wide, ref, 1e-2, 1e-2, "heads_to_wide")
    old = ops.gemm_nt(ops.head_expand(rows, H), wt[:, D:2 * D], torch.empty((n * H, D), dtype=torch.bfloat16, device=DEV))
    assert_close(wide, old, 2 ** -7, 1e-3, "heads_to_wide vs the block-sparse form")
    back = ops.wide_to_heads(wide, w[2 * D:], H, bias=bias)
    ref = torch.einsum("nhd,hcd->nhc", wide.float().view(n, H, D), w[2 * D:].float().view(H, 64, D)).reshape(n, D) + bias
    assert_close(back, ref, 1e-2, 2e-2, "wide_to_heads")
    old = ops.head_extract(ops.gemm_nt(wide, w[2 * D:], torch.empty((n * H, D), dtype=torch.bfloat16, device=DEV), bias=bias), H)
    assert_close(back, old, 2 ** -7, 2e-3, "wide_to_heads vs the block-sparse form")
    assert_close(ops.wide_to_heads(wide, w[2 * D:], H), ref - bias, 1e-2, 2e-2, "wide_to_heads, no bias")
    # round 5: the [n * H, D] side as a bf16 pair -- written as (bf16(acc), bf16(acc - hi)), read with both planes in the product
    pw = ops.heads_to_wide(rows, wt[:, D:2 * D], torch.full((2, n * H, D), 7.0, dtype=torch.bfloat16, device=DEV), H)
    assert torch.equal(pw[0], wide)
    exact = torch.einsum("nhc,hcd->nhd", rows.double().view(n, H, 64), w[D:2 * D].double().view(H, 64, D)).reshape(n * H, D)
    assert_close(pw[0].double() + pw[1].double(), exact, 2.0 ** -14, 1e-4, "heads_to_wide, pair")
    pb = ops.wide_to_heads(pw, w[2 * D:], H, bias=bias)
    ref2 = torch.einsum("nhd,hcd->nhc", (pw[0].double() + pw[1].double()).view(n, H, D), w[2 * D:].double().view(H, 64, D)).reshape(n, D) + bias.double()
    assert_close(pb, ref2, 2.0 ** -8, 2e-3, "wide_to_heads, pair in")


def test_head_expand_extract(ops):
    for n, H in ((5, 12), (3, 8), (2, 16)):
        D = 64 * H
        rows = rnd(n, D, seed=1, dtype=torch.bfloat16)
        x = ops.head_expand(rows, H).view(n, H, H, 64)
        for h in range(H):
            assert torch.equal(x[:, h, h], rows.view(n, H, 64)[:, h])
            x[:, h, h] = 0
        assert float(x.abs().max()) == 0.0
        bias = rnd(D, seed=2)
        for dt in (torch.bfloat16, torch.float32):
            full = rnd(n * H, D, seed=3).to(dt)
            diag = torch.stack([full.view(n, H, H, 64)[:, h, h] for h in range(H)], 1).reshape(n, D).float()
            assert torch.equal(ops.head_extract(full, H), diag.to(torch.bfloat16))
            assert torch.equal(ops.head_extract(full, H, bias), (diag + bias).to(torch.bfloat16))


def test_gather_and_add_rows(ops):
    batch, S, D = 5, 9, 768
    idx = torch.tensor([0, 8, 3, 3, 7], device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    r = torch.arange(batch, device=DEV) * S + idx
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        x = rnd(batch * S, D, seed=3).to(dt)
        assert torch.equal(ops.gather_rows(x, idx, batch, S), x[r])
        assert torch.equal(ops.gather_rows(x, None, batch, S), x[::S])
    x = rnd(batch * S, D, seed=4, dtype=torch.bfloat16)
    for add in (rnd(batch, D, seed=5), rnd(batch, D, seed=6, dtype=torch.bfloat16)):
        y = x.clone()
        ops.call("vipant_add_rows_bf16", y.data_ptr(), idx.data_ptr(), add.data_ptr(), int(add.dtype == torch.float32), batch, S, D, st)
        ref = x.clone(); ref[r] = (x[r].float() + add.float()).to(torch.bfloat16)
        assert torch.equal(y, ref)


# ------------------------------------------------------------------------------------------ layout helpers
def test_cast_transpose(ops):
    for R, C in ((768, 3072), (2304, 768), (100, 36), (768, 512)):
        w = rnd(R, C, seed=1)
        d, dt = ops.cast_bf16(w, True)
        assert torch.equal(d, w.to(torch.bfloat16)) and torch.equal(dt, w.to(torch.bfloat16).t().contiguous())


def test_l2norm(ops):
    x = rnd(37, 512, seed=1)
    y = ops.l2_normalize(x)
    assert_close(y, x / x.norm(dim=-1, keepdim=True), 1e-6, 1e-7, "l2norm")


# ------------------------------------------------------------------------------------------ InfoNCE
def nce_inputs(B):
    a = gen.det_randn(f"nce/a/{B}", (B, 512)); a = a / a.norm(dim=-1, keepdim=True)
    t = gen.det_randn(f"nce/t/{B}", (B, 512)); t = t / t.norm(dim=-1, keepdim=True)
    t = t + 0.5 * a; t = t / t.norm(dim=-1, keepdim=True)
    return a, t


@pytest.fixture(params=["rows", "tiles"])
def nce_path(request, monkeypatch):
    """Up to 768 clips the loss takes the row-block kernels (round 4), above that the 256 x 256 tile kernels; VIPANT_NCE_ROWS=0
    (read per call) sends every batch through the tile kernels, so the small fixtures keep covering both."""
    if request.param == "tiles":
        monkeypatch.setenv("VIPANT_NCE_ROWS", "0")
    return request.param


@pytest.mark.parametrize("B", [8, 32, 129])
@pytest.mark.parametrize("tag", ["", "_clamp", "_hot"])
def test_infonce_golden(ops, golden, nce_path, B, tag):
    """K8 boundary against vectors produced by the reference's CELossHead: loss within 1e-3 (north-star
    tolerance; observed ~1e-5), gradients within 1 % of their scale (bf16 MFMA operands in the backward)."""
    g = golden(f"infonce_B{B}{tag}")
    a, t = nce_inputs(B)
    a_, t_ = a.to(DEV).requires_grad_(), t.to(DEV).requires_grad_()
    ls = torch.tensor(float(g["logit_scale"]), device=DEV, requires_grad=True)
    loss = ops.InfoNCEFn.apply(a_, t_, ls, float(g["scale_max"]), 0, B, 1.0)
    loss.backward()
    loss = loss.detach()
    assert abs(float(loss) - float(g["loss"])) < 1e-3, (float(loss), float(g["loss"]))
    assert abs(float(loss) - float(g["loss"])) < 5e-5 * max(1.0, abs(float(g["loss"]))), "fp32-level agreement expected"
    da, dt = torch.from_numpy(g["da"]), torch.from_numpy(g["dt"])
    # softmax_row + softmax_col - 2I cancels to fp32 rounding on a saturated diagonal: absolute floor ~ eps * s
    floor = 3e-7 * math.exp(float(g["logit_scale"]))
    assert_close(a_.grad, da, 1e-2, 1e-2 * da.abs().max().item() + floor, "dx1")
    assert_close(t_.grad, dt, 1e-2, 1e-2 * dt.abs().max().item() + floor, "dx2")
    assert abs(float(ls.grad) - float(g["dls"])) < 1e-4 + 1e-3 * abs(float(g["dls"])), (float(ls.grad), float(g["dls"]))


@pytest.mark.parametrize("B,row0,nrows", [(512, 0, 512), (768, 512, 256), (769, 0, 769), (1000, 0, 1000), (1024, 256, 256), (4096, 3584, 512),
                                          (8192, 7168, 1024),                      # BASELINE.json configs[4]: 8 x 1024 clips
                                          # round 5, the strip path (B > 768, a strip of <= 768 rows: one rank of an N-GPU step; the two
                                          # cases above with 256 / 512 rows take it too): first / odd-start / maximal strips
                                          (4096, 0, 512), (2000, 1000, 500), (1025, 1, 768), (8192, 4096, 512),
                                          (432, 108, 108), (432, 324, 108), (8, 4, 4), (30, 3, 5)])      # strips off the 8-row grid
def test_infonce_large_and_sliced(ops, nce_path, B, row0, nrows):
    from oracle import ref_cpu as R
    a = rnd(B, 512, seed=1); a = a / a.norm(dim=-1, keepdim=True)
    t = rnd(B, 512, seed=2); t = t / t.norm(dim=-1, keepdim=True)
    t = t + 0.7 * a; t = t / t.norm(dim=-1, keepdim=True)
    ls = 3.1
    loss_ref, da, dt, dls = R.infonce_manual(a.cpu(), t.cpu(), ls, None)
    a_, t_ = a.clone().requires_grad_(), t.clone().requires_grad_()
    lsp = torch.tensor(ls, device=DEV, requires_grad=True)
    loss = ops.InfoNCEFn.apply(a_, t_, lsp, 0.0, row0, nrows, 2.0)
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-4, (float(loss), float(loss_ref))
    sl = slice(row0, row0 + nrows)
    assert_close(a_.grad[sl], 2.0 * da[sl], 1e-2, 2e-2 * da.abs().max().item(), "dx1 slice")
    assert_close(t_.grad[sl], 2.0 * dt[sl], 1e-2, 2e-2 * dt.abs().max().item(), "dx2 slice")
    if nrows < B:       # rows outside the strip: exact zeros
        rest = torch.cat([a_.grad[:row0], a_.grad[row0 + nrows:], t_.grad[:row0], t_.grad[row0 + nrows:]])
        assert float(rest.abs().max()) == 0.0
    assert abs(float(lsp.grad) - 2.0 * float(dls)) < 1e-3 * max(1.0, abs(2.0 * float(dls)))


# ------------------------------------------------------------------------------------------ LARS
def test_lars_golden(ops, golden):
    from oracle import ref_cpu as R
    g = golden("lars")
    ps = [gen.det_randn("lars/w0", (16, 24)), gen.det_randn("lars/w1", (4, 3, 5, 5)), gen.det_randn("lars/b0", (24,)),
          torch.ones([]) * 2.6593, torch.zeros(6, 6)]
    ps = [p.to(DEV).contiguous() for p in ps]
    st = ops.LarsState(ps, [p.ndim > 1 for p in ps])
    for step in range(12):
        lw, lb = R.adjust_learning_rate(step, epochs=3, steps_per_epoch=5, warmup_epoch=1, batch_size=64, lr_weight=0.2,
                                        lr_bias=0.0048)
        grads = [(gen.det_randn(f"lars/g{i}/{step}", tuple(p.shape)) * (0.0 if i == 4 and step < 2 else 1.0)).to(DEV)
                 for i, p in enumerate(ps)]
        st.step(grads, [lw if p.ndim > 1 else lb for p in ps], 1e-6, 0.9, 0.001)
        if step in (0, 1, 5, 11):
            for i, p in enumerate(ps):
                assert_close(p, torch.from_numpy(g[f"p{i}_{step}"]), 1e-5, 1e-6, f"lars p{i} step {step}")


def test_layernorm_fused_residual_add(ops):
    M, D = 1000, 768
    x = rnd(M, D, seed=1); y = rnd(M, D, seed=2, dtype=torch.bfloat16)
    gm = rnd(D, seed=3, scale=0.1) + 1.0; bt = rnd(D, seed=4, scale=0.1)
    h, _, mean, rstd, xs = ops.layernorm_fwd(x, gm, bt, add=y, want_sum=True)
    ref_sum = x.double() + y.double()
    assert_close(xs, ref_sum, 1e-6, 1e-6, "x + branch")
    assert_close(h, torch.nn.functional.layer_norm(ref_sum, (D,), gm.double(), bt.double(), 1e-5), 1e-2, 1e-2, "ln(x + branch)")
    assert_close(ops.residual_add(x, y), ref_sum, 1e-6, 1e-6, "residual_add")


# ------------------------------------------------------------------------------------------ retrieval evaluation
@pytest.mark.parametrize("N1,N2,G", [(40, 40, 1), (36, 180, 5), (700, 3500, 5), (1000, 513, 3), (257, 256, 1)])
def test_retrieval_ranks_match_argsort(ops, N1, N2, G):
    """ranks == position of the gold column in the reference's descending argsort (loss_head.py:113-118)."""
    g = torch.Generator().manual_seed(7)
    x1 = torch.nn.functional.normalize(torch.randn(N1, 512, generator=g), dim=-1)
    x2 = torch.nn.functional.normalize(torch.randn(N2, 512, generator=g), dim=-1)
    x2[: min(N1, N2)] += 0.2 * x1[: min(N1, N2)]                      # some structure
    x2 = torch.nn.functional.normalize(x2, dim=-1)
    gold = torch.randint(0, N2, (N1, G), generator=g)
    sim = x1.double() @ x2.double().t()
    pos = sim.argsort(descending=True).argsort()
    want = torch.gather(pos, 1, gold)
    ranks, top1 = ops.retrieval_ranks(x1.to(DEV), x2.to(DEV), gold.to(DEV), want_top1=True)
    assert ranks.shape == (N1, G) and ranks.dtype == torch.int32
    got = ranks.cpu().long()
    # similarities are fp32-grade (hi/lo bf16 split): the rank may differ from the exact one only by candidates
    # whose similarity is within 2e-6 of the gold's
    gs = torch.gather(sim, 1, gold)                                    # [N1, G]
    lo = (sim.unsqueeze(1) > (gs + 2e-6).unsqueeze(2)).sum(-1)
    hi = (sim.unsqueeze(1) > (gs - 2e-6).unsqueeze(2)).sum(-1)
    assert bool(((got >= lo) & (got <= hi)).all())
    assert (got != want).float().mean() < 2e-2      # an fp32 GEMM disagrees with the exact order about as often
    best = torch.gather(sim, 1, top1.cpu().long()[:, None])[:, 0]
    assert bool((sim.max(1)[0] - best < 2e-6).all())
    assert (top1.cpu().long() != sim.argmax(1)).float().mean() < 2e-2


def test_retrieval_report_goldens(golden):
    """LossHead.report strings against the reference's own output (tests/golden/make_golden.py, section vii)."""
    import os
    from vipant_amd import module as M
    from types import SimpleNamespace as NS
    import gen
    lh = M.build_loss_head(NS(name="CELossHead", layers=[], scaling=True, scale_max=None)).to(DEV)
    lh.eval()
    x1 = gen.det_randn("report/x1", (40, 512)); x2 = x1 + 0.9 * gen.det_randn("report/x2", (40, 512))
    x1 = x1 / x1.norm(dim=-1, keepdim=True); x2 = x2 / x2.norm(dim=-1, keepdim=True)
    x1, x2 = x1.to(DEV), x2.to(DEV)
    lh(x1[:25], x2[:25], normalized=True); lh(x1[25:], x2[25:], normalized=True)
    assert lh.report() == str(golden("report")["report"])
    g = golden("report_protocols")
    h2 = x1.cpu() + 0.45 * gen.det_randn("report/hard", (40, 512)); h2 = (h2 / h2.norm(dim=-1, keepdim=True)).to(DEV)
    lh(x1, h2, normalized=True)
    assert lh.report() == str(g["report_hard"])
    n = 36
    a = gen.det_randn("report5/a", (n, 512)); t = a.repeat_interleave(5, 0) + 9.0 * gen.det_randn("report5/t", (5 * n, 512))
    a = (a / a.norm(dim=-1, keepdim=True)).to(DEV); t = (t / t.norm(dim=-1, keepdim=True)).to(DEV)
    lh(a[:20], t[:100], normalized=True); lh(a[20:], t[100:], normalized=True)
    assert lh.report() == str(g["report_1v5"])
    names = [f"clip{i:03d}" for i in range(40)]
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "report_gold.jsonl")
    lh(x1[:25], h2[:25], normalized=True, names=names[:25]); lh(x1[25:], h2[25:], normalized=True, names=names[25:])
    assert lh.report(gold_file=gold) == str(g["report_gold"])
    lh(x1[:7], h2[:9], normalized=True)
    assert lh.report() == str(g["report_mismatch"])
    # un-normalised inputs are normalised on the way in (loss_head.py:38-41)
    lh(3.0 * x1, 0.5 * h2)
    assert lh.report() == str(g["report_hard"])


def test_zero_shot_report_golden(golden):
    """Zero-shot classification report (ClassificationHead.report with text prompts) against the reference's string."""
    from vipant_amd import module as M
    g = golden("report_protocols")
    prompts = gen.det_randn("zs/text", (50, 512)); prompts = prompts / prompts.norm(dim=-1, keepdim=True)
    lab = (torch.arange(200) * 7) % 50
    feats = prompts[lab] + 7.0 * gen.det_randn("zs/noise", (200, 512)) / 512 ** 0.5
    feats = feats / feats.norm(dim=-1, keepdim=True)
    perm = {i: (i * 3) % 50 for i in range(50)}
    assert M.zero_shot_report(feats.to(DEV), lab.to(DEV), prompts.to(DEV)) == str(g["zero_shot"])
    mapped = torch.tensor([perm[int(v)] for v in lab])
    assert M.zero_shot_report(feats.to(DEV), mapped.to(DEV), prompts.to(DEV), label_map=perm) == str(g["zero_shot_mapped"])


@pytest.mark.parametrize("B", [1, 2, 3, 7, 9, 255, 257])
def test_infonce_tiny_and_ragged_batches(ops, nce_path, B):
    """Degenerate and ragged batch sizes (B = 1 gives loss 0 and zero gradients, as the reference's cross entropy does)."""
    from oracle import ref_cpu as R
    g = torch.Generator().manual_seed(B)
    a = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1)
    t = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1)
    ls = torch.tensor(2.6593)
    a_, t_, l_ = a.to(DEV).requires_grad_(), t.to(DEV).requires_grad_(), ls.to(DEV).requires_grad_()
    loss = ops.InfoNCEFn.apply(a_, t_, l_, None, 0, B, 1.0)
    loss.backward()
    ar, tr, lr = a.clone().requires_grad_(), t.clone().requires_grad_(), ls.clone().requires_grad_()
    ref = R.ce_loss_head(ar, tr, lr, None)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-4
    scale = max(float(ar.grad.abs().max()), 1e-6)
    assert float((a_.grad.cpu() - ar.grad).abs().max()) <= 2e-2 * scale + 1e-7
    assert float((t_.grad.cpu() - tr.grad).abs().max()) <= 2e-2 * scale + 1e-7
    assert abs(float(l_.grad) - float(lr.grad)) <= 1e-3 * max(abs(float(lr.grad)), 1.0)



def test_round4_kernels_are_repeatable_under_load(ops):
    """The same race screen for the kernels of round 4 that stage through LDS-DMA or split work between waves: the folded last-block
    attention (forward, backward), the few-rows contraction (plain and per-head), the row-block InfoNCE -- sixty launches each at the
    step's shapes with a bandwidth hog on a second stream; bit-identical results every time (the InfoNCE loss is two atomic adds onto
    zero: a + b = b + a)."""
    b, S, H = 96, 316, 12
    D = 64 * H
    st = torch.cuda.current_stream().cuda_stream
    qk32 = rnd(b * H, D, seed=1, scale=0.35); h1 = rnd(b * S, D, seed=2, dtype=torch.bfloat16)
    qk = torch.stack([qk32.to(torch.bfloat16), (qk32 - qk32.to(torch.bfloat16).float()).to(torch.bfloat16)])      # a bf16 pair
    dctx = rnd(b * H, D, seed=3, dtype=torch.bfloat16)
    rows = rnd(512, D, seed=4, dtype=torch.bfloat16); w = rnd(3 * D, D, seed=5, dtype=torch.bfloat16, scale=D ** -0.5)
    wt = w.t().contiguous(); bias = rnd(D, seed=6)
    x1 = torch.nn.functional.normalize(rnd(512, 512, seed=7), dim=-1); x2 = torch.nn.functional.normalize(rnd(512, 512, seed=8) + 0.5 * x1, dim=-1)
    hog_src = torch.empty(64 << 20, dtype=torch.float32, device=DEV); hog_dst = torch.empty_like(hog_src)
    side = torch.cuda.Stream()
    ref = None
    for it in range(60):
        if it % 2:
            with torch.cuda.stream(side):
                hog_dst.copy_(hog_src)
        ctx = torch.empty(2, b * H, D, dtype=torch.bfloat16, device=DEV); probs = torch.empty(b, H, S, device=DEV)
        ops.call("vipant_rows_ctx_fwd", qk.data_ptr(), h1.data_ptr(), None, ctx.data_ptr(), probs.data_ptr(), b, S, H, 0, 1, st)
        dh1 = torch.empty(b * S, D, dtype=torch.bfloat16, device=DEV); dqk = torch.empty(2, b * H, D, dtype=torch.bfloat16, device=DEV)
        ops.call("vipant_rows_ctx_bwd", qk.data_ptr(), dctx.data_ptr(), ctx.data_ptr(), h1.data_ptr(), None, probs.data_ptr(), dh1.data_ptr(),
                 dqk.data_ptr(), None, b, S, H, 0, 1, st)
        few = ops.gemm_nt(rows, w[:D], torch.empty(512, D, dtype=torch.bfloat16, device=DEV), bias=bias, few_rows=True)
        res = ops.gemm_nt(rows, w[:D], torch.empty(512, D, device=DEV), bias=bias, aux=x1[:, :1].expand(512, D).contiguous(),
                          epi=ops.EPI_RESIDUAL_F32, few_rows=True)
        wide = ops.heads_to_wide(rows, wt[:, D:2 * D], torch.empty(2, 512 * H, D, dtype=torch.bfloat16, device=DEV), H)
        back = ops.wide_to_heads(wide, w[2 * D:], H, bias=bias)
        a_, t_ = x1.clone().requires_grad_(), x2.clone().requires_grad_()
        ls = torch.tensor(2.6593, device=DEV, requires_grad=True)
        loss = ops.InfoNCEFn.apply(a_, t_, ls, None, 0, 512, 1.0)
        loss.backward()
        cur = (ctx, probs, dh1, dqk, few, res, wide, back, loss.detach().clone(), a_.grad, t_.grad, ls.grad)
        if ref is None:
            ref = cur
        else:
            for k, (r_, c_) in enumerate(zip(ref, cur)):
                assert torch.equal(r_, c_), (it, k)
    torch.cuda.synchronize()


def test_pingpong_kernels_are_repeatable_under_load(ops):
    """Race screen for the ping-pong schedules: the same NT (bf16 and code epilogues) and TN launches forty times, interleaved with
    a bandwidth hog on a second stream that moves the LDS-DMA landing times around; every result must equal the first bit for bit
    (a read placed before its data is guaranteed to have landed shows up as a tile that comes and goes)."""
    M, K, N = 40000, 768, 3072
    a = rnd(M, K, seed=51, dtype=torch.bfloat16); b = rnd(N, K, seed=52, dtype=torch.bfloat16, scale=K ** -0.5)
    bias = rnd(N, seed=53)
    x3 = rnd(M, N, seed=54, dtype=torch.bfloat16)
    hog_src = torch.empty(64 << 20, dtype=torch.float32, device=DEV); hog_dst = torch.empty_like(hog_src)
    side = torch.cuda.Stream()
    ref = None
    for it in range(40):
        if it % 2:
            with torch.cuda.stream(side):
                hog_dst.copy_(hog_src)
        c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ops.gemm_nt(a, b, c, bias=bias, epi=ops.EPI_BF16)
        g = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); code = torch.empty(M, N, dtype=torch.uint8, device=DEV)
        ops.gemm_nt(a, b, g, bias=bias, aux=code, epi=ops.EPI_QUICKGELU_D8)
        dw = torch.empty(N, K, device=DEV)
        ops.gemm_tn(x3, a, dw)
        cur = (c, g, code, dw)
        if ref is None:
            ref = cur
        else:
            for r_, c_ in zip(ref, cur):
                assert torch.equal(r_, c_), it
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K,epi", [(40448, 3072, 768, "bf16"), (40448, 3072, 768, "gelu8"), (40448, 3072, 768, "dgelu8"),
                                       (70000, 768, 3072, "bf16"), (50001, 2304, 768, "bf16")])
def test_gemm_nt_ticket_walk_equals_static_walk(ops, monkeypatch, M, N, K, epi):
    """Round 5: the persistent NT kernels draw their tiles from per-XCD ticket queues (common.h) so that a workgroup whose CU is held
    by another stream's kernel -- the RCCL all-reduce of a gradient bucket -- costs 1/256 of a launch, not a round.  Which workgroup
    computes a tile must not matter: bit-identical to the static-stride walk (VIPANT_GEMM_VARIANT bit 22), with the chip to itself
    and with 64 CUs held for 300 us / 24 CUs for 3 ms by the probe kernel (tools/probes/comm_shadow.hip) on a second stream (late workgroups, some finding their
    queue empty); every element written each time (the ticket block must be back at zero after every launch)."""
    a = rnd(M, K, seed=61, dtype=torch.bfloat16); b = rnd(N, K, seed=62, dtype=torch.bfloat16, scale=K ** -0.5)
    bias = rnd(N, seed=63)
    code_in = torch.randint(0, 256, (M, N), dtype=torch.uint8, device=DEV)
    src = torch.empty(28 << 20, dtype=torch.uint8, device=DEV); dst = torch.empty_like(src)
    side = torch.cuda.Stream()

    def run():
        c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        if epi == "bf16":
            ops.gemm_nt(a, b, c, bias=bias, epi=ops.EPI_BF16)
            return (c,)
        if epi == "gelu8":
            code = torch.zeros(M, N, dtype=torch.uint8, device=DEV)
            ops.gemm_nt(a, b, c, bias=bias, aux=code, epi=ops.EPI_QUICKGELU_D8)
            return (c, code)
        ops.gemm_nt(a, b, c, aux=code_in, epi=ops.EPI_DQUICKGELU_D8)
        return (c,)

    monkeypatch.setenv("VIPANT_GEMM_VARIANT", "4194304")
    ref = run()
    assert all(torch.isfinite(t.float()).all() for t in ref)
    monkeypatch.setenv("VIPANT_GEMM_VARIANT", "0")
    for held, us in [(0, 0.0), (64, 300.0), (0, 0.0), (24, 3000.0), (0, 0.0)]:
        if held:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                probe_lib.comm_shadow(src.data_ptr(), dst.data_ptr(), src.numel(), held, us, side.cuda_stream)
        got = run()
        for r_, g_ in zip(ref, got):
            assert torch.equal(r_, g_), (held, us)
    torch.cuda.synchronize()
