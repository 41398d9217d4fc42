"""Pin the CPU oracle (oracle/ref_cpu.py) against vectors produced by the reference itself
(tests/golden/make_golden.py).  fp32 on CPU; tolerances cover only summation-order differences."""
import numpy as np
import pytest
import torch

import gen
from oracle import ref_cpu as R


def t(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=2e-4, atol=2e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"max err {err:.3e} vs ref max {ref:.3e}"


def nce_inputs(B):
    a = gen.det_randn(f"nce/a/{B}", (B, 512)); a = a / a.norm(dim=-1, keepdim=True)
    tt = gen.det_randn(f"nce/t/{B}", (B, 512)); tt = tt / tt.norm(dim=-1, keepdim=True)
    tt = tt + 0.5 * a; tt = tt / tt.norm(dim=-1, keepdim=True)
    return a, tt


@pytest.mark.parametrize("B", [8, 32, 129])
@pytest.mark.parametrize("tag", ["", "_clamp", "_hot"])
def test_infonce(golden, B, tag):
    g = golden(f"infonce_B{B}{tag}")
    a, tt = nce_inputs(B)
    close(gen.checksum(a), g["a_sum"], 1e-6, 1e-6)
    ls = torch.tensor(float(g["logit_scale"]), requires_grad=True)
    smax = float(g["scale_max"]) or None
    a_, t_ = a.clone().requires_grad_(), tt.clone().requires_grad_()
    loss = R.ce_loss_head(a_, t_, ls, smax)
    loss.backward()
    close(loss, g["loss"], 1e-5, 1e-6)
    close(a_.grad, g["da"]); close(t_.grad, g["dt"]); close(ls.grad, g["dls"], 1e-4, 5e-6)
    # closed-form gradients agree too (they are what the HIP kernel implements)
    l2, da, dt, dls = R.infonce_manual(a, tt, float(g["logit_scale"]), smax)
    close(l2, g["loss"], 1e-5, 1e-6); close(da, g["da"]); close(dt, g["dt"]); close(dls, g["dls"], 1e-4, 5e-6)


def test_valce(golden):
    g = golden("valce")
    a = gen.det_randn("valce/a", (8, 512)); tt = gen.det_randn("valce/t", (8, 512))
    a = a / a.norm(dim=-1, keepdim=True); tt = tt / tt.norm(dim=-1, keepdim=True)
    ls = torch.tensor(np.log(1 / 0.07), dtype=torch.float32)
    loss = R.valce_loss_head(None, a, tt, {"al": ls}, va=False, lv=False, al=True)
    close(loss, g["loss"], 1e-5, 1e-6)
    assert list(g["keys"]) == ["loss_head_al.logit_scale"]
    assert str(g["stats"]) == f"al {float(loss):.3f}"


@pytest.mark.parametrize("tag,D,S,b,causal", [("vit_s31", 768, 31, 2, False), ("vit_s316", 768, 316, 1, False),
                                                ("gpt_s77", 512, 77, 2, True)])
def test_block(golden, tag, D, S, b, causal):
    g = golden(f"block_{tag}")
    w = {k: v.requires_grad_() for k, v in gen.det_weights(f"block/{tag}", gen.backbone_shapes(D, 1)).items()}
    x = gen.det_randn(f"block/{tag}/x", (b, S, D)).requires_grad_()
    y = R.transformer_backbone(x, w, "encoder.", 1, D, 77 if causal else None, not causal)
    close(y, g["y"])
    y.backward(gen.det_randn(f"block/{tag}/gy", (b, S, D)))
    close(x.grad, g["dx"])
    for k, p in w.items():
        rk = k[len("encoder."):]
        close(p.grad.norm(), g[f"n_{rk}"], 2e-4, 1e-6)
        if f"g_{rk}" in g.files:
            close(p.grad, g[f"g_{rk}"])
    close(w["encoder.resblocks.0.attn.in_proj_weight"].grad[::97, ::13], g["g_in_proj_w_rows"])
    close(w["encoder.resblocks.0.mlp.c_fc.weight"].grad[::131, ::17], g["g_c_fc_w_rows"])


@pytest.mark.parametrize("tag,T,Fq,b", [("256x64", 256, 64, 3), ("1024x128", 1024, 128, 2)])
def test_pre_post(golden, tag, T, Fq, b):
    g = golden(f"prepost_{tag}")
    stride, S, pr = R.vit_position_resolution([T, Fq], 32, [16, 24])
    assert S == int(g["S"]) and tuple(pr) == tuple(g["position_resolution"])
    w = {k: v.requires_grad_() for k, v in gen.det_weights(f"prepost/{tag}", gen.vit_head_shapes(768, 1, 512, S)).items()}
    x = gen.det_randn(f"prepost/{tag}/x", (b, 1, T, Fq))
    pre = R.vit_pre_encoder(x, w, stride, w["misc.positional_embedding"], w["misc.class_embedding"])
    close(pre, g["pre"])
    pre.backward(gen.det_randn(f"prepost/{tag}/gpre", tuple(pre.shape)))
    close(w["misc.positional_embedding"].grad[::5], g["g_pos"])
    close(w["misc.class_embedding"].grad, g["g_cls"])
    gc = w["pre_encoder.conv1.weight"].grad
    close(gc.norm(), g["g_conv_norm"], 2e-4, 1e-6); close(gc[::61, :, ::5, ::7], g["g_conv_slice"])
    close(w["pre_encoder.ln.weight"].grad, g["g_ln_w"]); close(w["pre_encoder.ln.bias"].grad, g["g_ln_b"])
    h = gen.det_randn(f"prepost/{tag}/h", (b, S, 768))
    close(R.vit_post_encoder(h, w), g["post"])


def test_text_head(golden):
    g = golden("text_l2")
    w = gen.det_weights("text/l2", gen.text_head_shapes(512, 2, 512))
    assert sorted(w.keys()) == list(g["keys"])
    feat = R.text_head_forward(t(g["tokens"]), w, layers=2)
    close(feat, g["feat"])
    tok77 = gen.det_tokens("text/tok77", 4)
    assert int(tok77.sum()) == int(g["tok77_sum"])
    close(R.text_head_forward(tok77, w, layers=2), g["feat77"])


def test_image_head(golden):
    g = golden("image_l2")
    w = gen.det_weights("img/l2", gen.vit_head_shapes(768, 2, 512, 50))
    assert sorted(w.keys()) == list(g["keys"])
    x = gen.det_randn("img/l2/x", (2, 3, 224, 224))
    feat = R.vit_head_forward(x, w, width=768, layers=2, stride=[32, 32], position_resolution=(7, 7))
    close(feat, g["feat"])


@pytest.mark.parametrize("tag,L,b,T,Fq", [("L2", 2, 8, 256, 64), ("L12", 12, 32, 256, 64), ("T1000", 2, 4, 1000, 128)])
def test_end_to_end(golden, tag, L, b, T, Fq):
    g = golden(f"e2e_{tag}")
    stride, S, pr = R.vit_position_resolution([T, Fq], 32, [16, 24])
    assert S == int(g["S"]) == (31 if T == 256 else 306)
    w = {k: v.requires_grad_() for k, v in gen.det_weights(f"e2e/{tag}", gen.vit_head_shapes(768, L, 512, S)).items()}
    assert sum(v.numel() for v in w.values()) == int(g["n_params"])
    aud = gen.det_randn(f"e2e/{tag}/aud", (b, 1, T, Fq))
    img = gen.det_randn(f"e2e/{tag}/img", (b, 512))
    ls = torch.tensor(np.log(1 / 0.07), dtype=torch.float32, requires_grad=True)
    loss = R.cvalp_forward(img, aud, None, audio_sd=w, loss="ce", scales={"logit_scale": ls},
                           audio_cfg=dict(width=768, layers=L, stride=stride, position_resolution=pr))
    feat = R.vit_head_forward(aud, w, width=768, layers=L, stride=stride, position_resolution=pr)
    close(feat, g["feat"], 1e-3, 1e-5)
    close(loss, g["loss"], 1e-5, 1e-5)
    loss.backward()
    close(ls.grad, g["dls"], 1e-3, 1e-6)
    keys = list(g["keys"])
    assert keys == sorted(w.keys())
    gn = np.array([float(w[k].grad.norm()) for k in keys])
    assert np.allclose(gn, g["gnorm"], rtol=2e-3, atol=1e-7), np.abs(gn / g["gnorm"] - 1).max()
    close(w["misc.class_embedding"].grad, g["g_cls"], 2e-3, 1e-7)
    close(w["misc.positional_embedding"].grad, g["g_pos"], 2e-3, 1e-7)
    close(w["post_encoder.proj"].grad[::7, ::5], g["g_proj_slice"], 2e-3, 1e-7)
    close(w["pre_encoder.conv1.weight"].grad[::61, :, ::5, ::7], g["g_conv_slice"], 2e-3, 1e-7)
    close(w["encoder.resblocks.0.attn.in_proj_bias"].grad, g["g_b0_qkv_bias"], 2e-3, 1e-7)
    close(w[f"encoder.resblocks.{L - 1}.mlp.c_fc.bias"].grad, g["g_last_fc_bias"], 2e-3, 1e-7)


def test_lars_and_schedule(golden):
    g = golden("lars")
    ps = [gen.det_randn("lars/w0", (16, 24)), gen.det_randn("lars/w1", (4, 3, 5, 5)), gen.det_randn("lars/b0", (24,)),
          torch.ones([]) * 2.6593, torch.zeros(6, 6)]
    mus = [torch.zeros_like(p) for p in ps]
    for step in range(12):
        lw, lb = R.adjust_learning_rate(step, epochs=3, steps_per_epoch=5, warmup_epoch=1, batch_size=64,
                                        lr_weight=0.2, lr_bias=0.0048)
        assert np.allclose([lw, lb], g[f"lr_{step}"], rtol=1e-12)
        for i in range(len(ps)):
            grad = gen.det_randn(f"lars/g{i}/{step}", tuple(ps[i].shape)) * (0.0 if i == 4 and step < 2 else 1.0)
            ps[i], mus[i] = R.lars_step(ps[i], grad, mus[i], lw if ps[i].ndim > 1 else lb)
        if step in (0, 1, 5, 11):
            for i in range(len(ps)):
                close(ps[i], g[f"p{i}_{step}"], 1e-5, 1e-7)


def test_init_remap(golden):
    g = golden("init_remap")
    old = gen.det_randn("interp/pos50", (50, 64))
    close(R.interp_clip_vp_embedding(old, (15, 2)), g["pos_15x2"], 1e-6, 1e-7)
    close(R.interp_clip_vp_embedding(old, (63, 5)), g["pos_63x5"], 1e-6, 1e-7)
    close(R.interp_clip_vp_embedding(old, (7, 7)), g["pos_same"], 0, 0)


def test_report(golden):
    g = golden("report")
    x1 = gen.det_randn("report/x1", (40, 512)); x2 = x1 + 0.9 * gen.det_randn("report/x2", (40, 512))
    x1 = x1 / x1.norm(dim=-1, keepdim=True); x2 = x2 / x2.norm(dim=-1, keepdim=True)
    assert R.retrieval_report(x1, x2) == str(g["report"])


def _report_inputs():
    x1 = gen.det_randn("report/x1", (40, 512)); x1 = x1 / x1.norm(dim=-1, keepdim=True)
    h2 = x1 + 0.45 * gen.det_randn("report/hard", (40, 512)); h2 = h2 / h2.norm(dim=-1, keepdim=True)
    n = 36
    a = gen.det_randn("report5/a", (n, 512)); t = a.repeat_interleave(5, 0) + 9.0 * gen.det_randn("report5/t", (5 * n, 512))
    return x1, h2, a / a.norm(dim=-1, keepdim=True), t / t.norm(dim=-1, keepdim=True)


def test_report_protocols(golden):
    """LossHead.report: hard equal-size case, 1 clip vs 5 captions (+ retrieval_eval block), gold-file class statistics."""
    import os
    g = golden("report_protocols")
    x1, h2, a, t = _report_inputs()
    assert R.retrieval_report(x1, h2) == str(g["report_hard"])
    assert R.retrieval_report(a, t) == str(g["report_1v5"])
    names = [f"clip{i:03d}" for i in range(40)]
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "report_gold.jsonl")
    assert R.retrieval_report(x1, h2, ids=names, gold_file=gold) == str(g["report_gold"])
    assert R.retrieval_report(x1[:7], h2[:9]) == str(g["report_mismatch"])


def _zero_shot_inputs():
    prompts = gen.det_randn("zs/text", (50, 512)); prompts = prompts / prompts.norm(dim=-1, keepdim=True)
    lab = (torch.arange(200) * 7) % 50
    feats = prompts[lab] + 7.0 * gen.det_randn("zs/noise", (200, 512)) / 512 ** 0.5
    return feats / feats.norm(dim=-1, keepdim=True), lab, prompts, {i: (i * 3) % 50 for i in range(50)}


def test_zero_shot_report(golden):
    g = golden("report_protocols")
    feats, lab, prompts, perm = _zero_shot_inputs()
    assert R.zero_shot_report(feats, lab, prompts) == str(g["zero_shot"])
    mapped = torch.tensor([perm[int(v)] for v in lab])
    assert R.zero_shot_report(feats, mapped, prompts, label_map=perm) == str(g["zero_shot_mapped"])



@pytest.mark.parametrize("tag", ["va", "at"])
def test_trainer_trajectory(golden, tag):
    """Four optimisation steps (warm-up into cosine, LARS) of the reference's own heads / loss head / optimizer, VA and AT
    layouts: the oracle's forward, autograd and `lars_step` / `adjust_learning_rate` reproduce every step's learning rates,
    loss and per-tensor update norm, and the final values of the small tensors."""
    g = golden(f"traj_{tag}")
    L, b, T, Fq = 2, 8, 256, 64
    stride, S, pr = R.vit_position_resolution([T, Fq], 32, [16, 24])
    asd = {k: v.clone().requires_grad_() for k, v in gen.det_weights(f"traj/{tag}", gen.vit_head_shapes(768, L, 512, S)).items()}
    ls = torch.tensor(float(np.log(1 / 0.07)), requires_grad=True)
    lkey = "loss_head.loss_head_al.logit_scale" if tag == "at" else "loss_head.logit_scale"
    named = [(f"audio_head.{k}", v) for k, v in asd.items()] + [(lkey, ls)]
    order = {k: i for i, k in enumerate(list(g["keys"]))}
    assert set(order) == {k for k, _ in named}
    tsd = gen.det_weights("traj/text", gen.text_head_shapes(512, L, 512)) if tag == "at" else None
    mus = {k: torch.zeros_like(v) for k, v in named}
    for step in range(4):
        lw, lb = R.adjust_learning_rate(step, epochs=3, steps_per_epoch=2, warmup_epoch=1, batch_size=b, lr_weight=0.2,
                                        lr_bias=0.0048)
        close([lw, lb], g["lrs"][step], 1e-6, 1e-12)
        aud = gen.det_randn(f"traj/{tag}/aud/{step}", (b, 1, T, Fq))
        feat = R.vit_head_forward(aud, asd, width=768, layers=L, stride=stride, position_resolution=pr)
        if tag == "at":
            with torch.no_grad():
                tf = R.text_head_forward(gen.det_tokens(f"traj/tok/{step}", b), tsd, width=512, layers=L, ctx_len=77)
            loss = R.valce_loss_head(None, feat, tf, {"al": ls}, va=False, lv=False, al=True)
        else:
            loss = R.ce_loss_head(R.l2_normalize(gen.det_randn(f"traj/{tag}/img/{step}", (b, 512))), feat, ls)
        for _, v in named:
            v.grad = None
        loss.backward()
        assert abs(float(loss) - float(g["losses"][step])) < 2e-4, (step, float(loss), float(g["losses"][step]))
        with torch.no_grad():
            for k, v in named:
                p_new, mus[k] = R.lars_step(v.detach(), v.grad, mus[k], lw if v.ndim > 1 else lb)
                dn = float((p_new - v.detach()).norm())
                ref = float(g["dnorm"][step][order[k]])
                assert abs(dn - ref) <= 1e-7 + 2e-3 * ref, (step, k, dn, ref)
                v.copy_(p_new)
    for k, v in named:
        if f"final_{k}" in g.files:
            close(v.detach(), g[f"final_{k}"], 1e-4, 1e-6)


def test_frozen_towers_full_depth(golden):
    """The frozen CLIP text transformer (12 blocks, causal, EOT read-out; full-width and short batches) and ViT-B/32 image tower
    (12 blocks) against the reference's features."""
    g = golden("towers_l12")
    tsd = gen.det_weights("text/l12", gen.text_head_shapes(512, 12, 512))
    isd = gen.det_weights("img/l12", gen.vit_head_shapes(768, 12, 512, 50))
    tok = gen.det_tokens("text/l12/tok", 6)
    assert int(tok.sum()) == int(g["tok_sum"])
    with torch.no_grad():
        close(R.text_head_forward(tok, tsd, width=512, layers=12, ctx_len=77), g["text_feat"], 2e-4, 2e-6)
        close(R.text_head_forward(gen.det_tokens("text/l12/short", 5, L=40)[:, :40], tsd, width=512, layers=12, ctx_len=77),
              g["text_feat_short"], 2e-4, 2e-6)
        close(R.vit_head_forward(gen.det_randn("img/l12/x", (3, 3, 224, 224)), isd, width=768, layers=12, stride=[32, 32],
                                 position_resolution=(7, 7)), g["image_feat"], 2e-4, 2e-6)
