"""BASELINE.json configs[1] at FULL size on the MI355X (per-GPU batch 512, 1024 x 128 spectrograms, S = 316, M = 161 792
token rows): the shapes `bench.py` times, checked -- the CPU oracle cannot run a 512-clip batch in test time, so parity at
this size goes through (i) sampled rows / whole reductions against fp32 PyTorch on the same device arrays, and (ii) a
size-independent property of the path: samples are independent through the towers, so a sample's activations and input
gradients in the 512-clip batch must equal, bit for bit, what the same sample gives in an 8-clip batch (whose numerics
are pinned to the reference by tests/test_model_gpu.py).
"""
import math
import os
import sys
from types import SimpleNamespace as NS

import pytest
import torch

pytestmark = pytest.mark.gpu

import gen  # noqa: E402

DEV = "cuda:0"
B, S, D = 512, 316, 768
M = B * S


@pytest.fixture(scope="module")
def ops():
    from vipant_amd import _ffi, ops as O
    _ffi.call("vipant_device_check")
    return O


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float32):
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(dtype)


def rows_sample(seed=0, n=1536):
    g = torch.Generator().manual_seed(seed)
    r = torch.randint(0, M, (n,), generator=g)
    edge = torch.tensor([0, 1, 255, 256, 257, M - 257, M - 256, M - 2, M - 1])      # first / last tile, tile seams
    return torch.cat([r, edge]).to(DEV)


def max_rel(got, ref):
    return float((got.double() - ref.double()).abs().max() / ref.double().abs().max())


def test_full_size_token_contractions(ops):
    """The five NT launches of a block at M = 161 792 (incl. both QuickGELU epilogues) on sampled rows, and two of the weight-
    gradient reductions over all 161 792 tokens, against fp32 matmuls of the same bf16 arrays."""
    x = rnd(M, D, seed=1, dtype=torch.bfloat16)
    w_qkv = rnd(3 * D, D, seed=2, scale=D ** -0.5, dtype=torch.bfloat16)
    w_fc = rnd(4 * D, D, seed=3, scale=D ** -0.5, dtype=torch.bfloat16)
    w_pr = rnd(D, 4 * D, seed=4, scale=(4 * D) ** -0.5, dtype=torch.bfloat16)
    b_qkv, b_fc, b_pr = rnd(3 * D, seed=5), rnd(4 * D, seed=6), rnd(D, seed=7)
    rows = rows_sample()
    xs = x[rows].float()

    qkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(x, w_qkv, qkv, bias=b_qkv, epi=ops.EPI_BF16)
    assert max_rel(qkv[rows], xs @ w_qkv.float().t() + b_qkv) < 6e-3               # bf16 output rounding: 2^-9 of the row scale

    u = torch.empty(M, 4 * D, dtype=torch.bfloat16, device=DEV); g = torch.empty_like(u)
    ops.gemm_nt(x, w_fc, g, bias=b_fc, aux=u, epi=ops.EPI_QUICKGELU)
    pre = xs @ w_fc.float().t() + b_fc
    assert max_rel(u[rows], pre) < 6e-3
    assert max_rel(g[rows], pre * torch.sigmoid(1.702 * pre)) < 8e-3
    torch.cuda.synchronize()

    y = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(g, w_pr, y, bias=b_pr, epi=ops.EPI_BF16)                            # K = 3072
    assert max_rel(y[rows], g[rows].float() @ w_pr.float().t() + b_pr) < 6e-3

    du = torch.empty(M, 4 * D, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(x, w_pr.t().contiguous(), du, aux=u, epi=ops.EPI_DQUICKGELU)        # d(c_proj) with QuickGELU'
    us = u[rows].float()
    sg = torch.sigmoid(1.702 * us)
    assert max_rel(du[rows], (xs @ w_pr.float()) * (sg * (1 + 1.702 * us * (1 - sg)))) < 8e-3

    # weight gradients: reductions over all M tokens (split over workgroups, deterministic second pass) + fused column sums
    dW = torch.empty(4 * D, D, device=DEV); db = torch.empty(4 * D, device=DEV)
    ops.gemm_tn(du, x, dW, a_colsum=db)
    ref = torch.zeros(4 * D, D, device=DEV, dtype=torch.float64)
    refb = torch.zeros(4 * D, device=DEV, dtype=torch.float64)
    for c in range(0, M, 16384):                                                    # fp64 accumulation of fp32 chunk products
        ref += (du[c:c + 16384].float().t() @ x[c:c + 16384].float()).double()
        refb += du[c:c + 16384].float().sum(0).double()
    assert max_rel(dW, ref) < 2e-4, max_rel(dW, ref)                                # fp32 accumulation over 161 792 terms
    assert max_rel(db, refb) < 2e-4
    dW2 = torch.empty_like(dW)
    ops.gemm_tn(du, x, dW2)
    assert torch.equal(dW, dW2)                                                     # no float atomics: run-to-run identical


def test_full_batch_equals_small_batches(ops):
    """Sample independence at cfg2 size: a 2-block audio stack on all 512 clips (M = 161 792) gives, for the first and the last
    8 clips, exactly the activations and input gradients of 8-clip runs; the weight gradients of the full batch equal the sum
    over 64 chunks of 8 up to fp32 summation order."""
    import vipant_amd.module as Mod
    layers = 2
    bb = Mod.TransformerBackbone(NS(layers=layers, skip_attn_mask=True), width=D, ctx_len=None)
    w = gen.det_weights("full/768", gen.backbone_shapes(D, layers))
    bb.load_state_dict({k[len("encoder."):]: v for k, v in w.items()}, strict=True)
    bb = bb.to(DEV)
    x = rnd(B, S, D, seed=11)
    gy = rnd(B, S, D, seed=12)
    xf = x.clone().requires_grad_()
    yf = bb(xf)
    yf.backward(gy)
    full = {k: p.grad.clone() for k, p in bb.named_parameters()}
    acc = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in full.items()}
    for c in range(0, B, 8):
        for p in bb.parameters():
            p.grad = None
        xs = x[c:c + 8].clone().requires_grad_()
        ys = bb(xs)
        ys.backward(gy[c:c + 8])
        if c in (0, B - 8):
            assert torch.equal(ys, yf[c:c + 8]), c
            assert torch.equal(xs.grad, xf.grad[c:c + 8]), c
        for k, p in bb.named_parameters():
            acc[k] += p.grad.double()
    for k in full:
        assert max_rel(full[k], acc[k]) < 1e-4, (k, max_rel(full[k], acc[k]))


def test_full_size_step_starts_at_two_ln_batch():
    """bench.py's workload through the trainer: the first VA step at batch 512 on a from-scratch tower starts at
    2 ln 512 = 12.477 (uniform softmax both ways) plus the spread of the initial logits; the loss falls over the next steps."""
    from vipant_amd.config import compose
    from vipant_amd.module import adjust_learning_rate
    from vipant_amd.monitor import VAMonitor
    ov = ("+running=bimodal worker=CVALP mode=dp eval=False +model/image=vit_val +model/audio=vit_val +model/text=dummy "
          "+model/loss=ce +optimizer=standard +running/audio=default model.audio.pre_encoder.in_channels=3 "
          "model.audio.pre_encoder.stride=[16,24] model.image.encoder.layers=2 running.audio.max_len=1024 "
          "running.audio.num_mel_bins=128 running.batch_size=512 running.epochs=1000 running.save_epoch=False "
          "running.save_rate=1e9 running.peep_rate=1000000 running.synthetic_steps=4 num_gpus=1").split()
    cfg = compose(ov)
    cfg.rank = 0
    torch.manual_seed(cfg.seed)
    mon = VAMonitor(cfg, lambda *_: None, torch.device(DEV))
    g = torch.Generator().manual_seed(1213)
    images = torch.randn(B, 3, 224, 224, generator=g).to(DEV)
    audios = torch.randn(B, 1, 1024, 128, generator=g).to(DEV)
    # the step-0 loss against the ORACLE's loss head on the features the HIP towers produce for the same 512 clips (the towers
    # themselves are pinned to the reference at fixture sizes and, at this size, by the sample-independence property above)
    from oracle import ref_cpu as R
    from vipant_amd import ops
    with torch.no_grad():
        f_img = mon.model.image_head(images, normalized=True)
        f_aud = mon.model.audio_head(audios, normalized=True)
    ls0 = float(mon.model.loss_head.logit_scale.detach())
    loss_oracle = float(R.ce_loss_head(f_img.cpu(), f_aud.cpu(), torch.tensor(ls0)))
    losses = []
    for i in range(3):
        adjust_learning_rate(cfg.optimizer, mon.optimizer, mon.dataloader, i + 10)
        losses.append(float(mon.step(images, audios, None).detach()))
    assert all(math.isfinite(v) for v in losses), losses
    assert abs(losses[0] - loss_oracle) < 1e-4, (losses[0], loss_oracle)
    assert abs(losses[0] - 2 * math.log(B)) < 0.5, losses            # and it starts near the uniform-softmax value
    assert losses[-1] < losses[0], losses


def test_full_size_last_block_rows_match_full_block():
    """The benchmarked shape (audio ViT-B, 12 blocks, 1024 x 128 spectrograms -> S = 316, 512 clips) with the weights of the
    `e2e_cfg2` fixture: loss + backward with the last block on its read-out rows (`running.last_block_rows`, the default) against
    the same with the full last block -- features, loss and the gradient of every parameter.  (From-scratch weights would not do:
    their features are nearly identical across clips and the contrastive gradient is rounding noise in either mode.)"""
    import vipant_amd.module as M
    from vipant_amd import ops
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_model_gpu import audio_cfg
    runs = []
    aud = gen.det_randn("full/rows/aud", (B, 1, 1024, 128)).to(DEV)
    img = ops.l2_normalize(gen.det_randn("full/rows/img", (B, 512)).to(DEV))
    for rows in (False, True):
        head = M.build_audio_head(audio_cfg(1024, 128, 12))
        assert head.misc.positional_embedding.shape[0] == S
        head.load_state_dict(gen.det_weights("e2e/cfg2", gen.vit_head_shapes(768, 12, 512, S)), strict=True)
        lhead = M.build_loss_head(NS(name="CELossHead", layers=[], scaling=True, scale_max=None))
        head, lhead = head.to(DEV).train(), lhead.to(DEV).train()
        head.encoder.last_block_rows = rows
        feat = head(aud, normalized=True)
        loss = lhead(img, feat, None, normalized=True)
        loss.backward()
        runs.append((feat.detach().clone(), float(loss), {k: p.grad.double() for k, p in head.named_parameters()}))
        del head, lhead, feat, loss
        torch.cuda.empty_cache()
    (f0, l0, g0), (f1, l1, g1) = runs
    diffs = sorted(((float((g1[k] - g0[k]).norm() / g0[k].norm().clamp_min(1e-30)), k) for k in g0), reverse=True)
    print(f"full size, read-out rows vs full last block: feature max rel diff {max_rel(f1, f0):.2e}, loss {l0:.5f} / {l1:.5f}, "
          f"worst gradient rel-L2 diff {diffs[0][0]:.3e} ({diffs[0][1]}), median {diffs[len(diffs) // 2][0]:.3e}")
    from test_model_gpu import observe
    observe("full_size_last_block_rows_vs_full_block", loss_full=l0, loss_rows=l1, loss_abs_diff=abs(l1 - l0), feat_max_rel=max_rel(f1, f0),
            worst_grad_rel_l2=diffs[0][0], median_grad_rel_l2=diffs[len(diffs) // 2][0], batch=B, tokens=S)
    assert max_rel(f1, f0) < 1e-2, max_rel(f1, f0)
    assert abs(l1 - l0) < 1e-3, (l0, l1)
    assert diffs[0][0] < 5e-2, diffs[0]


def test_full_size_at_step():
    """BASELINE.json configs[2] shape on one GPU at full size: AT fine-tuning layout -- 512 clips, audio ViT-B (12 blocks, S = 316) +
    the frozen 12-block causal text transformer at 77 tokens with its end-of-text read-out rows + VALCELossHead(al).  The step-0 loss
    equals the ORACLE's loss head on the features the HIP towers produce (<= 1e-4), and the two towers' last blocks on their read-out
    rows (`running.last_block_rows`, the default: class row for audio, the end-of-text row of each caption for text) agree with the
    full last blocks in features, loss and every audio gradient."""
    import vipant_amd.module as M
    from oracle import ref_cpu as R
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_model_gpu import audio_cfg, text_cfg
    aud = gen.det_randn("full/at/aud", (B, 1, 1024, 128)).to(DEV)
    tok = gen.det_tokens("full/at/tok", B).to(DEV)
    assert tok.shape == (B, 77)
    runs = []
    for rows in (False, True):
        head = M.build_audio_head(audio_cfg(1024, 128, 12))
        head.load_state_dict(gen.det_weights("e2e/cfg2", gen.vit_head_shapes(768, 12, 512, S)), strict=True)
        thead = M.build_text_head(text_cfg(12))
        thead.load_state_dict(gen.det_weights("text/l12", gen.text_head_shapes(512, 12, 512)), strict=True)
        lhead = M.build_loss_head(NS(name="VALCELossHead", layers=[], scaling=True, scale_max=None, va=False, lv=False, al=True))
        head, thead, lhead = head.to(DEV).train(), thead.to(DEV).eval(), lhead.to(DEV).train()
        for q in thead.parameters():
            q.requires_grad = False
        head.encoder.last_block_rows = rows
        thead.encoder.last_block_rows = rows
        feat = head(aud, normalized=True)
        with torch.no_grad():
            tf = thead(tok, normalized=True)
        loss = lhead(None, feat, tf, normalized=True)
        loss.backward()
        ls = {"al": lhead.loss_head_al.logit_scale.detach().cpu().clone()}
        loss_oracle = float(R.valce_loss_head(None, feat.detach().cpu(), tf.cpu(), ls, va=False, lv=False, al=True))
        assert abs(float(loss) - loss_oracle) < 1e-4, (rows, float(loss), loss_oracle)
        runs.append((feat.detach().clone(), tf.clone(), float(loss), {k: p.grad.double() for k, p in head.named_parameters()}))
        del head, thead, lhead, feat, loss
        torch.cuda.empty_cache()
    (f0, t0, l0, g0), (f1, t1, l1, g1) = runs
    diffs = sorted(((float((g1[k] - g0[k]).norm() / g0[k].norm().clamp_min(1e-30)), k) for k in g0), reverse=True)
    print(f"full-size AT step, read-out rows vs full last blocks: audio feature max rel diff {max_rel(f1, f0):.2e}, text {max_rel(t1, t0):.2e}, "
          f"loss {l0:.5f} / {l1:.5f}, worst gradient rel-L2 diff {diffs[0][0]:.3e} ({diffs[0][1]}), median {diffs[len(diffs) // 2][0]:.3e}")
    assert max_rel(f1, f0) < 1e-2 and max_rel(t1, t0) < 1e-2, (max_rel(f1, f0), max_rel(t1, t0))
    assert abs(l1 - l0) < 1e-3, (l0, l1)
    assert diffs[0][0] < 5e-2, diffs[0]


def test_vit_l_depth_sample_independence(ops):
    """BASELINE.json configs[4] tower (audio ViT-L: 24 blocks, width 1024, 16 heads, S = 316) with e4m3 contractions and recomputed
    MLP activations: the outputs and input gradients of 8 clips in a 16-clip batch equal, bit for bit, those of an 8-clip run; the
    weight gradients are the sum over the chunks up to fp32 summation order.  (Round 6: the e4m3 forms carry one scale per 32
    consecutive TOKEN rows x 32 columns -- what the e4m3 weight-gradient contraction needs -- so a chunk is independent of the rest
    of the batch when its token count is a multiple of 32: 8 clips x 316 tokens = 79 blocks.  Every per-GPU batch of BASELINE.json,
    512 or 1024 clips, is such a chunk, which is what keeps N replicas equal to one process.)"""
    import vipant_amd.module as Mod
    layers, width, b16 = 24, 1024, 16
    bb = Mod.TransformerBackbone(NS(layers=layers, skip_attn_mask=True), width=width, ctx_len=None)
    w = gen.det_weights("full/1024", gen.backbone_shapes(width, layers))
    bb.load_state_dict({k[len("encoder."):]: v for k, v in w.items()}, strict=True)
    bb = bb.to(DEV)
    bb.fp8, bb.recompute_mlp = True, True
    x = rnd(b16, S, width, seed=21)
    gy = rnd(b16, S, width, seed=22)
    xf = x.clone().requires_grad_()
    yf = bb(xf)
    yf.backward(gy)
    assert torch.isfinite(yf).all()
    full = {k: p.grad.clone() for k, p in bb.named_parameters()}
    acc = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in full.items()}
    assert (8 * S) % 32 == 0
    for c in range(0, b16, 8):
        for p in bb.parameters():
            p.grad = None
        xs = x[c:c + 8].clone().requires_grad_()
        ys = bb(xs)
        ys.backward(gy[c:c + 8])
        assert torch.equal(ys, yf[c:c + 8]), c
        assert torch.equal(xs.grad, xf.grad[c:c + 8]), c
        for k, p in bb.named_parameters():
            acc[k] += p.grad.double()
    for k in full:
        assert max_rel(full[k], acc[k]) < 1e-4, (k, max_rel(full[k], acc[k]))


def test_cfg5_full_size_tower_step(ops):
    """BASELINE.json configs[4]'s tower at FULL size in the driver's own test run ("nothing driver-run has
    allocated that step"): audio ViT-L stack (24 blocks, width 1024, 16 heads), 1024 clips x 316 tokens = 323 584 token rows, e4m3
    contractions with MX block scales, recomputed MLP activations (~150 GB; without recomputation the step fits too, 235 GB:
    profiles/r5_cfg5_mx.md).  One forward + backward; the first and last eight clips' outputs and input gradients equal, bit for bit,
    those of 8-clip runs (a chunk whose token count is a multiple of 32 -- 8 x 316 -- shares no quantisation block with the rest of
    the batch, see test_vit_l_depth_sample_independence, so the 1024-clip numerics are the ones tests/test_model_gpu.py pins to
    the reference at small batches)."""
    import vipant_amd.module as Mod
    if torch.cuda.get_device_properties(0).total_memory < 200 * 2 ** 30:
        pytest.skip("needs the 288 GB of an MI355X")
    torch.cuda.empty_cache()
    layers, width, b = 24, 1024, 1024
    bb = Mod.TransformerBackbone(NS(layers=layers, skip_attn_mask=True), width=width, ctx_len=None)
    w = gen.det_weights("full/1024", gen.backbone_shapes(width, layers))
    bb.load_state_dict({k[len("encoder."):]: v for k, v in w.items()}, strict=True)
    bb = bb.to(DEV)
    bb.fp8, bb.recompute_mlp = True, True
    x = rnd(b, S, width, seed=31)
    gy = rnd(b, S, width, seed=32)
    torch.cuda.reset_peak_memory_stats()
    xf = x.clone().requires_grad_()
    yf = bb(xf)
    yf.backward(gy)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() / 1e9
    print(f"cfg5 tower step at 1024 clips: peak {peak:.1f} GB")
    assert torch.isfinite(yf).all() and torch.isfinite(xf.grad).all()
    for k, p in bb.named_parameters():
        assert torch.isfinite(p.grad).all(), k
    keep = [(c, yf[c:c + 8].clone(), xf.grad[c:c + 8].clone()) for c in (0, b - 8)]
    del yf, xf
    for p in bb.parameters():
        p.grad = None
    torch.cuda.empty_cache()
    for c, y_ref, dx_ref in keep:
        xs = x[c:c + 8].clone().requires_grad_()
        ys = bb(xs)
        ys.backward(gy[c:c + 8])
        assert torch.equal(ys, y_ref), c
        assert torch.equal(xs.grad, dx_ref), c
