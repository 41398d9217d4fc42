"""e4m3 operands of the block contractions (BASELINE.json configs[4], "fp8 MFMA weights").  The reference has no fp8 path to
compare with (its contractions are fp16 autocast matmuls, clip/model.py:170-187), so the bar here is exactness against an
emulation of the stated format: the quantiser bit for bit against torch's float8_e4m3fn rounding, the contraction against an
fp32 matmul of the SAME dequantised operands within bf16 output rounding, and the distance to the unquantised product reported."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from vipant_amd import _ffi, ops as O
    _ffi.call("vipant_device_check")
    return O


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    return torch.randn(*shape, generator=g, device=DEV) * scale


def emulate_quant(x_bf16):
    """per row: e = smallest exponent with amax / 2^e <= 448; q = rne(x / 2^e) in OCP e4m3."""
    x = x_bf16.float()
    amax = x.abs().amax(dim=1)
    e = torch.floor(torch.log2(amax.clamp_min(1e-38))) - 8
    e = torch.where(amax * torch.exp2(-e) > 448, e + 1, e)
    e = torch.where(amax > 0, e, torch.zeros_like(e)).clamp(-127, 127)
    q = (x * torch.exp2(-e)[:, None]).to(torch.float8_e4m3fn)
    return q, (e + 127).to(torch.uint8)


def dequant(q_u8, s_u8):
    return q_u8.view(torch.float8_e4m3fn).float() * torch.exp2(s_u8.float() - 127)[:, None]


@pytest.mark.parametrize("M,K", [(5, 256), (300, 1024), (1031, 768), (64, 4096)])
def test_quantiser_matches_the_stated_format(ops, M, K):
    rows = torch.exp2(torch.randint(-12, 12, (M, 1), device=DEV).float())                 # row magnitudes over 24 octaves
    x = (rnd(M, K, seed=M + K) * rows).to(torch.bfloat16)
    x[0].zero_()                                                                           # an all-zero row
    x[1, 3] = 448.0 * 2 ** 5                                                               # amax exactly on the format's maximum
    q, s = ops.quant_e4m3(x)
    q_ref, s_ref = emulate_quant(x)
    assert torch.equal(s, s_ref)
    assert torch.equal(q, q_ref.view(torch.uint8))
    back = dequant(q, s)
    rel = ((back - x.float()).abs().amax(dim=1) / x.float().abs().amax(dim=1).clamp_min(1e-30))
    assert float(rel.max()) <= 2 ** -4 + 1e-6                                              # 3 mantissa bits: half an ulp of the row maximum


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (300, 264, 384), (1000, 768, 1024), (4100, 2304, 768), (515, 3072, 4096)])
def test_e4m3_contraction_is_exact_on_its_operands(ops, M, N, K):
    ra = torch.exp2(torch.randint(-6, 7, (M, 1), device=DEV).float())
    rb = torch.exp2(torch.randint(-6, 7, (N, 1), device=DEV).float())
    a = (rnd(M, K, seed=1) * ra).to(torch.bfloat16)
    b = (rnd(N, K, seed=2) * rb * K ** -0.5).to(torch.bfloat16)
    bias = rnd(N, seed=3)
    qa, sa = ops.quant_e4m3(a)
    qb, sb = ops.quant_e4m3(b)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt_e4m3(qa, sa, qb, sb, c, bias=bias)
    ref = dequant(qa, sa).double() @ dequant(qb, sb).double().t() + bias.double()
    # bf16 output rounding only (2^-9 relative per element), row by row since the rows span 12 octaves
    err = ((c.double() - ref).abs() / ref.abs().amax(dim=1, keepdim=True)).max()
    assert float(err) < 2 ** -8, float(err)
    full = a.double() @ b.double().t() + bias.double()
    dist = ((c.double() - full).norm(dim=1) / full.norm(dim=1)).median()
    assert float(dist) < 6e-2, float(dist)            # e4m3 x e4m3 on gaussian rows: ~2^-4 / sqrt(3) per factor, observed ~3.7e-2


def test_e4m3_quickgelu_epilogues(ops):
    M, N, K = 1024, 3072, 768
    a = rnd(M, K, seed=5).to(torch.bfloat16)
    w = (rnd(N, K, seed=6) * K ** -0.5).to(torch.bfloat16)
    bias = rnd(N, seed=7)
    qa, sa = ops.quant_e4m3(a)
    qw, sw = ops.quant_e4m3(w)
    g = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    code = torch.empty(M, N, dtype=torch.uint8, device=DEV)
    ops.gemm_nt_e4m3(qa, sa, qw, sw, g, bias=bias, aux=code, epi=ops.EPI_QUICKGELU_D8)
    u = (dequant(qa, sa).double() @ dequant(qw, sw).double().t() + bias.double()).float()
    ub = u.to(torch.bfloat16).float()                                             # the epilogue sees the bf16-rounded pre-activation
    sg = torch.sigmoid(1.702 * ub)
    assert float(((g.float() - ub * sg).abs() / (ub * sg).abs().amax()).max()) < 2 ** -7       # one bf16 ulp of u at the top of the range
    d = sg * (1 + 1.702 * ub * (1 - sg))
    assert float((code.float() / 212.5 - 0.1 - d).abs().max()) < 5e-3
    # backward form: dg (bf16, quantised here) x W^T with the derivative code applied
    dy = rnd(M, K, seed=8).to(torch.bfloat16)
    wt = w.t().contiguous()                                                       # [K, N]: rows of the transposed weight
    qd, sd = ops.quant_e4m3(dy)
    qt, st = ops.quant_e4m3(wt)
    # du[M, N] = dy[M, K] @ w[N, K]^T needs B = w as [N, K] rows: the same operand as the forward
    du = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt_e4m3(qd, sd, qw, sw, du, aux=code, epi=ops.EPI_DQUICKGELU_D8)
    ref = (dequant(qd, sd).double() @ dequant(qw, sw).double().t()).float().to(torch.bfloat16).float() * (code.float() / 212.5 - 0.1)
    assert float(((du.float() - ref).abs() / ref.abs().amax()).max()) < 2 ** -7
