"""Module-level parity on the MI355X against golden vectors produced by the imported reference
(tests/golden/make_golden.py): transformer block, pre / post encoders, frozen text and image towers, and the
end-to-end VA step (features, loss, gradients).

Tolerances.  The HIP path feeds bf16 operands to MFMA (fp32 accumulate, fp32 residual stream); the golden vectors
are the reference's fp32 CPU path.  bf16 rounding (2^-9 relative per operand) gives ~0.3-1 % error on a single
contraction output and a few % on gradients that pass through 12 layers; each check below states its budget
relative to the tensor's own scale (max |ref|), never absolute-only.
"""
import math
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gen  # noqa: E402

DEV = "cuda:0"


def observe(name, **vals):
    """Observed parity numbers go to gpurun_out/parity_observed.jsonl (summarised in profiles/r2_parity_observed.md)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "parity_observed.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **{k: float(v) for k, v in vals.items()}}) + "\n")


TRAJ_CORNERS = {}        # (script, stream, rows) -> (losses of the four steps, calls of the read-out-row kernels)


@pytest.fixture(scope="module")
def M():
    import vipant_amd.module as mod
    from vipant_amd import _ffi
    _ffi.call("vipant_device_check")
    return mod


def rel_err(got, ref):
    got, ref = torch.as_tensor(got).double().cpu(), torch.as_tensor(ref).double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def rel_l2(got, ref):
    got, ref = torch.as_tensor(got).double().cpu(), torch.as_tensor(ref).double().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


def audio_cfg(T, Fq, layers, name="CLIPAudioHead"):
    return NS(name=name, width=768, embed_dim=512, resolution=[T, Fq], ctx_len=77,
              encoder=NS(name="TransformerBackbone", layers=layers, skip_attn_mask=True),
              pre_encoder=NS(name="ViTPreEncoder", patch_size=32, stride=[16, 24], in_channels=3),
              post_encoder=NS(name="ViTPostEncoder"), misc=NS(name="CLIPMisc"),
              pre_encoder_addon=NS(name="AddonEncoder"), post_encoder_addon=NS(name="AddonEncoder"))


def image_cfg(layers):
    return NS(name="CLIPImageHead", width=768, embed_dim=512, resolution=224, ctx_len=77,
              encoder=NS(name="TransformerBackbone", layers=layers, skip_attn_mask=True),
              pre_encoder=NS(name="ViTPreEncoder", patch_size=32, stride=32, in_channels=3),
              post_encoder=NS(name="ViTPostEncoder"), misc=NS(name="CLIPMisc"),
              pre_encoder_addon=NS(name="AddonEncoder"), post_encoder_addon=NS(name="AddonEncoder"))


def text_cfg(layers):
    return NS(name="CLIPTextHead", width=512, embed_dim=512, resolution=None, ctx_len=77,
              encoder=NS(name="TransformerBackbone", layers=layers, skip_attn_mask=False),
              pre_encoder=NS(name="GPTPreEncoder", vocab_size=49408),
              post_encoder=NS(name="GPTPostEncoder"), misc=NS(name="CLIPMisc"),
              pre_encoder_addon=NS(name="AddonEncoder"), post_encoder_addon=NS(name="AddonEncoder"))


@pytest.mark.parametrize("tag,D,S,b,causal", [("vit_s31", 768, 31, 2, False), ("vit_s316", 768, 316, 1, False),
                                                ("gpt_s77", 512, 77, 2, True)])
def test_block_golden(M, golden, tag, D, S, b, causal):
    g = golden(f"block_{tag}")
    bb = M.TransformerBackbone(NS(layers=1, skip_attn_mask=not causal), width=D, ctx_len=77 if causal else None)
    w = gen.det_weights(f"block/{tag}", gen.backbone_shapes(D, 1))
    bb.load_state_dict({k[len("encoder."):]: v for k, v in w.items()}, strict=True)
    bb = bb.to(DEV)
    x = gen.det_randn(f"block/{tag}/x", (b, S, D)).to(DEV).requires_grad_()
    y = bb(x)
    assert rel_err(y, g["y"]) < 1.5e-2, rel_err(y, g["y"])                 # one block, bf16 operands
    y.backward(gen.det_randn(f"block/{tag}/gy", (b, S, D)).to(DEV))
    assert rel_err(x.grad, g["dx"]) < 3e-2, rel_err(x.grad, g["dx"])
    for k, p in bb.named_parameters():
        ref_norm = float(g[f"n_{k}"])
        assert abs(float(p.grad.norm()) / ref_norm - 1) < 2e-2, (k, float(p.grad.norm()), ref_norm)
        if f"g_{k}" in g.files:
            assert rel_err(p.grad, g[f"g_{k}"]) < 4e-2, (k, rel_err(p.grad, g[f"g_{k}"]))
    sl = dict(bb.named_parameters())
    assert rel_err(sl["resblocks.0.attn.in_proj_weight"].grad[::97, ::13], g["g_in_proj_w_rows"]) < 4e-2
    assert rel_err(sl["resblocks.0.mlp.c_fc.weight"].grad[::131, ::17], g["g_c_fc_w_rows"]) < 4e-2


@pytest.mark.parametrize("D,S,b,layers", [(1024, 50, 2, 2), (768, 306, 1, 1), (256, 17, 3, 1), (768, 428, 1, 1)])
def test_backbone_other_widths_match_oracle(M, D, S, b, layers):
    """Widths beyond the two reference towers (ViT-L's 1024 = BASELINE configs[4], heads = width // 64, val.py:474) and the
    shipped-default token counts (T=1000: S=306 with the scripts' stride [16,24], S=428 with the YAML's [16,16], which takes the
    streaming attention kernels): HIP stack against the CPU restatement."""
    from oracle import ref_cpu as R
    bb = M.TransformerBackbone(NS(layers=layers, skip_attn_mask=True), width=D, ctx_len=None)
    w = gen.det_weights(f"wide/{D}", gen.backbone_shapes(D, layers))
    sd = {k[len("encoder."):]: v for k, v in w.items()}
    bb.load_state_dict(sd, strict=True)
    bb = bb.to(DEV)
    x = gen.det_randn(f"wide/{D}/x", (b, S, D))
    gy = gen.det_randn(f"wide/{D}/gy", (b, S, D))
    xr = x.clone().requires_grad_()
    sdr = {k: v.clone().requires_grad_() for k, v in sd.items()}
    yr = R.transformer_backbone(xr, sdr, "", layers, D, None, True)          # the restatement is batch-first
    yr.backward(gy)
    xg = x.to(DEV).requires_grad_()
    y = bb(xg)
    y.backward(gy.to(DEV))
    assert rel_err(y, yr.detach()) < 2e-2, rel_err(y, yr.detach())
    assert rel_err(xg.grad, xr.grad) < 4e-2, rel_err(xg.grad, xr.grad)
    for k, p in bb.named_parameters():
        assert rel_err(p.grad, sdr[k].grad) < 5e-2, (k, rel_err(p.grad, sdr[k].grad))


@pytest.mark.parametrize("b,S", [(1, 1), (1, 5), (3, 2)])
def test_backbone_degenerate_shapes(M, b, S):
    """One sample / one token: every kernel of the stack has to survive M = b * S far below a tile."""
    from oracle import ref_cpu as R
    bb = M.TransformerBackbone(NS(layers=1, skip_attn_mask=True), width=768, ctx_len=None)
    w = gen.det_weights("tiny/768", gen.backbone_shapes(768, 1))
    sd = {k[len("encoder."):]: v for k, v in w.items()}
    bb.load_state_dict(sd, strict=True)
    bb = bb.to(DEV)
    x = gen.det_randn(f"tiny/x/{b}/{S}", (b, S, 768))
    xg = x.to(DEV).requires_grad_()
    y = bb(xg)
    y.backward(torch.ones_like(y))
    xr = x.clone().requires_grad_()
    yr = R.transformer_backbone(xr, sd, "", 1, 768, None, True)
    yr.backward(torch.ones_like(yr))
    assert rel_err(y, yr.detach()) < 2e-2 and rel_err(xg.grad, xr.grad) < 4e-2


@pytest.mark.parametrize("tag,T,Fq,b", [("256x64", 256, 64, 3), ("1024x128", 1024, 128, 2)])
def test_pre_post_golden(M, golden, tag, T, Fq, b):
    g = golden(f"prepost_{tag}")
    head = M.build_audio_head(audio_cfg(T, Fq, 1))
    S = head.misc.positional_embedding.shape[0]
    assert S == int(g["S"]) and tuple(head.misc.position_resolution) == tuple(g["position_resolution"])
    head.load_state_dict(gen.det_weights(f"prepost/{tag}", gen.vit_head_shapes(768, 1, 512, S)), strict=True)
    head = head.to(DEV)
    x = gen.det_randn(f"prepost/{tag}/x", (b, 1, T, Fq)).to(DEV)
    pre = head.pre_encoder(x, positional_embedding=head.misc.pos_embedding, class_embedding=head.misc.cls_embedding)
    assert rel_err(pre, g["pre"]) < 1e-2, rel_err(pre, g["pre"])
    pre.backward(gen.det_randn(f"prepost/{tag}/gpre", tuple(pre.shape)).to(DEV))
    assert rel_err(head.misc.positional_embedding.grad[::5], g["g_pos"]) < 1e-3       # fp32 LN / reduction path
    assert rel_err(head.misc.class_embedding.grad, g["g_cls"]) < 1e-3
    assert rel_err(head.pre_encoder.ln.weight.grad, g["g_ln_w"]) < 1e-2
    assert rel_err(head.pre_encoder.ln.bias.grad, g["g_ln_b"]) < 1e-3
    gc = head.pre_encoder.conv1.weight.grad
    assert abs(float(gc.norm()) / float(g["g_conv_norm"]) - 1) < 1e-2
    assert rel_err(gc[::61, :, ::5, ::7], g["g_conv_slice"]) < 2e-2
    h = gen.det_randn(f"prepost/{tag}/h", (b, S, 768)).to(DEV)
    post = head.post_encoder(h)
    assert rel_err(post, g["post"]) < 1e-2, rel_err(post, g["post"])


def test_text_head_golden(M, golden):
    g = golden("text_l2")
    head = M.build_text_head(text_cfg(2))
    w = gen.det_weights("text/l2", gen.text_head_shapes(512, 2, 512))
    assert sorted(head.state_dict().keys()) == list(g["keys"])
    head.load_state_dict(w, strict=True)
    head = head.to(DEV).eval()
    with torch.no_grad():
        feat = head(torch.from_numpy(g["tokens"]).to(DEV), normalized=True)
        feat77 = head(gen.det_tokens("text/tok77", 4).to(DEV), normalized=True)
    assert rel_err(feat, g["feat"]) < 2e-2, rel_err(feat, g["feat"])
    assert rel_err(feat77, g["feat77"]) < 2e-2, rel_err(feat77, g["feat77"])


def test_image_head_golden(M, golden):
    g = golden("image_l2")
    head = M.build_image_head(image_cfg(2))
    assert sorted(head.state_dict().keys()) == list(g["keys"])
    head.load_state_dict(gen.det_weights("img/l2", gen.vit_head_shapes(768, 2, 512, 50)), strict=True)
    head = head.to(DEV).eval()
    with torch.no_grad():
        feat = head(gen.det_randn("img/l2/x", (2, 3, 224, 224)).to(DEV), normalized=True)
    assert rel_err(feat, g["feat"]) < 2e-2, rel_err(feat, g["feat"])


# |loss - reference| budgets, about 2x the values observed on MI355X in round 5 (profiles/r5_parity_observed.jsonl; fp32 / fp16 stream):
# L2 (b = 8, two blocks) 2.6e-3 / 3.3e-3; L12 (b = 32) 3.5e-4 / 4.7e-4; T1000 (b = 4) 1.3e-3 / 1.3e-3; cfg2 (the benchmarked shape, b = 64)
# 3.9e-4 / 2.4e-4 against the north-star budget of 1e-3 itself.  History of cfg2: round 3 4.1e-4 / 4.9e-4, round 4 4.5e-4 / 6.0e-4 (the
# folded last block rounded qk / contexts to bf16 per (item, head)), round 5 3.9e-4 / 2.4e-4 (those tensors travel as bf16 pairs).  The
# loss kernel's own boundary: tests/test_kernels_gpu.py::test_infonce_golden, observed < 5e-5 x loss.
# L2 (which change moved it from round 4's 2.1e-3?): measured in round 6 on one box, the default build reads 2.56e-3 / 3.31e-3
# and the same build with the round-3 form of the last block (VIPANT_LAST_BLOCK_CTX=0: K / V projected for every token) 2.31e-3 / 2.73e-3
# -- the folded last block (bf16 pairs for qk / contexts) accounts for 0.25e-3 / 0.6e-3, the walk order and paired draws for nothing (they
# are bit-identical to the static walk), the rest is the two bf16 blocks' noise at 8 clips.  The budget stays at 6e-3 = ~2x the larger one.
E2E_LOSS_BUDGET = {"L2": 6e-3, "L12": 1e-3, "T1000": 3e-3, "cfg2": 1e-3}


@pytest.mark.parametrize("stream", ["fp32", "fp16"])
@pytest.mark.parametrize("tag,L,b,T,Fq", [("L2", 2, 8, 256, 64), ("L12", 12, 32, 256, 64), ("T1000", 2, 4, 1000, 128),
                                          ("cfg2", 12, 64, 1024, 128)])
def test_end_to_end_golden(M, golden, tag, L, b, T, Fq, stream):
    """VA step on precomputed image embeddings at the cfg1 shape (256x64 spectrograms), at the shipped default spectrogram
    size (1000 x 128 -> S = 306) and at the BENCHMARKED shape (cfg2: 12 blocks, 1024 x 128 -> S = 316, 64 clips): features, InfoNCE
    loss, gradients against the reference's own outputs, with the residual stream inside the stack in fp32 and in fp16
    (`running.stream_dtype`)."""
    tag_obs = tag if stream == "fp32" else f"{tag}_stream16"
    g = golden(f"e2e_{tag}")
    head = M.build_audio_head(audio_cfg(T, Fq, L))
    S = head.misc.positional_embedding.shape[0]
    head.load_state_dict(gen.det_weights(f"e2e/{tag}", gen.vit_head_shapes(768, L, 512, S)), strict=True)
    assert sum(p.numel() for p in head.parameters()) == int(g["n_params"])
    lhead = M.build_loss_head(NS(name="CELossHead", layers=[], scaling=True, scale_max=None))
    head, lhead = head.to(DEV).train(), lhead.to(DEV).train()
    head.encoder.stream_f16 = stream == "fp16"
    from vipant_amd import ops
    aud = gen.det_randn(f"e2e/{tag}/aud", (b, 1, T, Fq)).to(DEV)
    img = ops.l2_normalize(gen.det_randn(f"e2e/{tag}/img", (b, 512)).to(DEV))
    feat = head(aud, normalized=True)
    loss = lhead(img, feat, None, normalized=True)
    loss.backward()
    # features: unit vectors; budget 2 % of the largest component after L bf16 layers (observed < 1 %)
    assert rel_err(feat, g["feat"]) < 2e-2, rel_err(feat, g["feat"])
    cos = torch.nn.functional.cosine_similarity(feat.detach().double().cpu(), torch.from_numpy(g["feat"]).double(), dim=-1)
    assert float(cos.min()) > 0.9995, float(cos.min())
    # loss: north-star budget 1e-3 is for the K8 boundary (tests/test_kernels_gpu.py); end to end through bf16
    # towers the stated budget is 5e-3 absolute on a loss of ~2 ln(b)
    observe(f"e2e_{tag_obs}", loss_hip=float(loss), loss_ref=float(g["loss"]), loss_abs_err=abs(float(loss) - float(g["loss"])),
            feat_rel_err=rel_err(feat, g["feat"]), min_cos=float(cos.min()),
            dls_rel_err=abs(float(lhead.logit_scale.grad) - float(g["dls"])) / max(abs(float(g["dls"])), 1e-3))
    assert abs(float(loss) - float(g["loss"])) < E2E_LOSS_BUDGET[tag], (float(loss), float(g["loss"]))
    assert abs(float(lhead.logit_scale.grad) - float(g["dls"])) < 2e-2 * max(abs(float(g["dls"])), 1e-3)
    grads = {k: p.grad for k, p in head.named_parameters()}
    keys = list(g["keys"])
    assert keys == sorted(grads.keys())
    gn = np.array([float(grads[k].norm()) for k in keys])
    ratio = gn / g["gnorm"]
    assert np.all(np.abs(ratio - 1) < 5e-2), (keys[int(np.abs(ratio - 1).argmax())], ratio.min(), ratio.max())
    observe(f"e2e_{tag_obs}_grads", gnorm_ratio_max_dev=float(np.abs(ratio - 1).max()),
            cls=rel_l2(grads["misc.class_embedding"], g["g_cls"]), pos=rel_l2(grads["misc.positional_embedding"], g["g_pos"]),
            proj=rel_l2(grads["post_encoder.proj"][::7, ::5], g["g_proj_slice"]),
            conv=rel_l2(grads["pre_encoder.conv1.weight"][::61, :, ::5, ::7], g["g_conv_slice"]),
            qkv_bias0=rel_l2(grads["encoder.resblocks.0.attn.in_proj_bias"], g["g_b0_qkv_bias"]),
            fc_bias_last=rel_l2(grads[f"encoder.resblocks.{L - 1}.mlp.c_fc.bias"], g["g_last_fc_bias"]))
    assert rel_l2(grads["misc.class_embedding"], g["g_cls"]) < 5e-2
    assert rel_l2(grads["misc.positional_embedding"], g["g_pos"]) < 5e-2
    assert rel_l2(grads["post_encoder.proj"][::7, ::5], g["g_proj_slice"]) < 5e-2
    assert rel_l2(grads["pre_encoder.conv1.weight"][::61, :, ::5, ::7], g["g_conv_slice"]) < 5e-2
    assert rel_l2(grads["encoder.resblocks.0.attn.in_proj_bias"], g["g_b0_qkv_bias"]) < 5e-2
    assert rel_l2(grads[f"encoder.resblocks.{L - 1}.mlp.c_fc.bias"], g["g_last_fc_bias"]) < 5e-2


def _audio_step(M, handoff, monkeypatch, extra_consumer=False):
    from vipant_amd import ops
    monkeypatch.setattr(ops, "NODE_HANDOFF", handoff)
    L, b, T, Fq = 2, 8, 256, 64
    head = M.build_audio_head(audio_cfg(T, Fq, L))
    S = head.misc.positional_embedding.shape[0]
    head.load_state_dict(gen.det_weights("e2e/L2", gen.vit_head_shapes(768, L, 512, S)), strict=True)
    lhead = M.build_loss_head(NS(name="CELossHead", layers=[], scaling=True, scale_max=None))
    head, lhead = head.to(DEV).train(), lhead.to(DEV).train()
    aud = gen.det_randn("e2e/L2/aud", (b, 1, T, Fq)).to(DEV)
    img = ops.l2_normalize(gen.det_randn("e2e/L2/img", (b, 512)).to(DEV))
    seen = {}
    if extra_consumer:       # a second consumer of the stack's output and of its input: autograd sums real gradients into both
        pre = head.pre_encoder(aud, positional_embedding=head.misc.pos_embedding, class_embedding=head.misc.cls_embedding)
        x = head.encoder(pre)
        feat = head.post_encoder(x)
        feat = feat / feat.norm(dim=-1, keepdim=True)
        loss = lhead(img, feat, None, normalized=True) + 1e-3 * x.float().square().mean() + 1e-3 * pre.float().square().mean()
    else:
        feat = head(aud, normalized=True)
        loss = lhead(img, feat, None, normalized=True)
        node = feat.grad_fn
        while node is not None and getattr(node, "_vipant_kind", None) != "stack":
            node = node.next_functions[0][0] if node.next_functions else None
        seen["stack"] = node
    loss.backward()
    return float(loss), {k: p.grad.clone() for k, p in head.named_parameters()}, seen


def test_node_handoff_matches_dense_gradients(M, monkeypatch):
    """The compact read-out gradient / bf16 stream gradient hand-off between the tower's three autograd nodes (vipant_amd/ops.py,
    `_producer`) against dense fp32 gradients through autograd: same values; only the top block's c_proj.bias (column sums taken
    over the compact rows instead of the dense matrix) may differ in summation order."""
    l0, g0, _ = _audio_step(M, False, monkeypatch)
    l1, g1, seen = _audio_step(M, True, monkeypatch)
    assert seen["stack"] is not None and seen["stack"].readout_grad is None          # the hand-off was consumed
    assert l0 == l1
    for k in g0:
        if k.endswith("resblocks.1.mlp.c_proj.bias"):
            assert rel_l2(g1[k], g0[k]) < 1e-6, k
        else:
            assert torch.equal(g1[k], g0[k]), (k, rel_l2(g1[k], g0[k]))


def test_node_handoff_partial_and_repeated_backward(M, monkeypatch):
    """the hand-off must survive (i) a partial backward that runs the read-out node but not the stack's --
    `torch.autograd.grad` over the read-out's parameters with retain_graph -- followed by the full backward (the rows handed over by
    the first call are NOT added a second time), (ii) a second full backward over a retained graph fails loudly (the stack's node has
    released its weight copies) instead of silently using stale state, and (iii) with the hand-off switched off a watched stream
    tensor sees the real gradient (the documented way to inspect it)."""
    from vipant_amd import ops
    monkeypatch.setattr(ops, "NODE_HANDOFF", True)
    L, b, T, Fq = 2, 8, 256, 64

    def build():
        head = M.build_audio_head(audio_cfg(T, Fq, L))
        S = head.misc.positional_embedding.shape[0]
        head.load_state_dict(gen.det_weights("e2e/L2", gen.vit_head_shapes(768, L, 512, S)), strict=True)
        lhead = M.build_loss_head(NS(name="CELossHead", layers=[], scaling=True, scale_max=None))
        return head.to(DEV).train(), lhead.to(DEV).train()

    aud = gen.det_randn("e2e/L2/aud", (b, 1, T, Fq)).to(DEV)
    img = ops.l2_normalize(gen.det_randn("e2e/L2/img", (b, 512)).to(DEV))
    head, lhead = build()
    loss = lhead(img, head(aud, normalized=True), None, normalized=True)
    loss.backward()
    ref = {k: p.grad.clone() for k, p in head.named_parameters()}

    # (i) partial backward first, then the full one
    head, lhead = build()
    loss = lhead(img, head(aud, normalized=True), None, normalized=True)
    post = [p for k, p in head.named_parameters() if k.startswith("post_encoder.")]
    gpart = torch.autograd.grad(loss, post, retain_graph=True)
    for gp, p in zip(gpart, post):
        k = [n for n, q in head.named_parameters() if q is p][0]
        assert torch.equal(gp, ref[k]), k
    loss.backward()
    for k, p in head.named_parameters():
        assert torch.equal(p.grad, ref[k]), (k, rel_l2(p.grad, ref[k]))
    # (ii) the stack's node gives its bf16 weight copies back in its backward: a second pass over a retained graph is refused loudly
    with pytest.raises(Exception, match="second backward"):
        loss2 = lhead(img, head(aud, normalized=True), None, normalized=True)
        loss2.backward(retain_graph=True)
        loss2.backward()

    # (iii) inspecting the stream's gradient: a tensor handed DIRECTLY to the next node with retain_grad() / hooks set keeps the dense
    # path; a watcher on an intermediate view cannot be seen from the node (views are separate tensors) and observes the zero
    # placeholder -- VIPANT_NODE_HANDOFF=0 (ops.NODE_HANDOFF = False) is the switch for that, and gives the real gradient
    monkeypatch.setattr(ops, "NODE_HANDOFF", False)
    head, lhead = build()
    pre = head.pre_encoder(aud, positional_embedding=head.misc.pos_embedding, class_embedding=head.misc.cls_embedding)
    x = head.encoder(pre)
    x.retain_grad()
    feat = head.post_encoder(x)
    feat = feat / feat.norm(dim=-1, keepdim=True)
    lhead(img, feat, None, normalized=True).backward()
    assert x.grad is not None and float(x.grad.abs().max()) > 0
    for k, p in head.named_parameters():        # (called piecewise the stack evaluates its full last block: bf16 rounding apart)
        assert rel_l2(p.grad, ref[k]) < 3e-2, (k, rel_l2(p.grad, ref[k]))


def test_node_handoff_with_other_consumers(M, monkeypatch):
    """When something else also consumes the stack's input or output, autograd delivers a real gradient and the hand-off is
    added to it: same gradients as with the hand-off switched off."""
    l0, g0, _ = _audio_step(M, False, monkeypatch, extra_consumer=True)
    l1, g1, _ = _audio_step(M, True, monkeypatch, extra_consumer=True)
    assert abs(l0 - l1) < 1e-6
    for k in g0:
        assert rel_l2(g1[k], g0[k]) < 2e-3, (k, rel_l2(g1[k], g0[k]))     # bf16 rounding of (a + b) vs a, b separately


def _tower_inputs(kind, layers):
    if kind == "audio":
        b, w = 6, None
        x = gen.det_randn("rows/aud", (b, 1, 256, 64))
    else:
        b = 5 if kind == "text" else 2          # "text2": two clips, the smallest batch with a negative
        w = gen.det_weights("text/l2", gen.text_head_shapes(512, layers, 512))
        x = gen.det_tokens("rows/tok", b)
    other = gen.det_randn(f"rows/other/{kind}", (b, 512))
    return b, w, x, other


def _tower_step(M, kind, last_block_rows, layers, ctx_form=True):
    """One loss + backward through a trainable tower; returns (features, loss, gradients)."""
    from vipant_amd import ops
    b, w, x, other = _tower_inputs(kind, layers)
    if kind == "audio":
        head = M.build_audio_head(audio_cfg(256, 64, layers))
        S = head.misc.positional_embedding.shape[0]
        head.load_state_dict(gen.det_weights("e2e/L2", gen.vit_head_shapes(768, layers, 512, S)), strict=True)
    else:
        head = M.build_text_head(text_cfg(layers))
        head.load_state_dict(w, strict=True)
    x = x.to(DEV)
    lhead = M.build_loss_head(NS(name="CELossHead", layers=[], scaling=True, scale_max=None))
    head, lhead = head.to(DEV).train(), lhead.to(DEV).train()
    head.encoder.last_block_rows = last_block_rows
    other = ops.l2_normalize(other.to(DEV))
    keep, ops.LAST_BLOCK_CTX = ops.LAST_BLOCK_CTX, ctx_form
    try:
        feat = head(x, normalized=True)
        loss = lhead(other, feat, None, normalized=True)
        loss.backward()
    finally:
        ops.LAST_BLOCK_CTX = keep
    return feat.detach(), float(loss), {k: p.grad.clone() for k, p in head.named_parameters() if p.grad is not None}


def _tower_step_oracle(kind, layers):
    """The same step through the CPU restatement of the reference (fp32): the value both HIP realisations approximate."""
    from oracle import ref_cpu as R
    b, w, x, other = _tower_inputs(kind, layers)
    if kind == "audio":
        stride, S, pr = R.vit_position_resolution([256, 64], 32, [16, 24])
        w = gen.det_weights("e2e/L2", gen.vit_head_shapes(768, layers, 512, S))
    w = {k: v.clone().requires_grad_() for k, v in w.items()}
    if kind == "audio":
        feat = R.vit_head_forward(x, w, width=768, layers=layers, stride=stride, position_resolution=pr)
    else:
        feat = R.text_head_forward(x, w, layers=layers)
    loss = R.ce_loss_head(R.l2_normalize(other), feat, torch.tensor(math.log(1 / 0.07)))
    loss.backward()
    return feat.detach(), float(loss), {k: v.grad for k, v in w.items() if v.grad is not None}


@pytest.mark.parametrize("kind,layers", [("audio", 2), ("audio", 1), ("text", 2), ("text2", 1)])
def test_last_block_on_readout_rows_matches_full_block(M, kind, layers):
    """`running.last_block_rows` (ops.BackboneFn `rows`): the last block evaluated on the read-out rows only -- class token of the
    audio ViT, end-of-text token of the causal text tower -- in both of its forms (key / value projection folded into the query
    side: csrc/readout_ctx.hip, the default; K and V of every token: csrc/readout_rows.hip) against the full block AND against the
    CPU restatement of the reference: the three HIP realisations are bf16 roundings of the same function in different orders, so
    they must sit equally close to the fp32 value -- features, loss, the gradient of EVERY parameter (the last block's included)."""
    fo, lo, go = _tower_step_oracle(kind, layers)
    runs = {"full": _tower_step(M, kind, False, layers), "rows_kv": _tower_step(M, kind, True, layers, ctx_form=False),
            "rows_ctx": _tower_step(M, kind, True, layers, ctx_form=True)}
    f0, l0, g0 = runs["full"]
    err = {}
    for name, (f1, l1, g1) in runs.items():
        assert set(g1) <= set(go)
        worst = max((rel_l2(g1[k], go[k]), k) for k in g1)
        err[name] = (rel_err(f1, fo), abs(l1 - lo), worst[0])
        observe(f"last_block_{name}_{kind}_L{layers}:{worst[1]}", feat_rel_err_oracle=rel_err(f1, fo), loss_abs_err_oracle=abs(l1 - lo),
                worst_grad_rel_l2_oracle=worst[0], feat_rel_err_full=rel_err(f1, f0), loss_abs_diff_full=abs(l1 - l0))
    for name, (f1, l1, g1) in runs.items():
        # against the oracle: the budgets of the golden-vector tests of these towers (features 2e-2: test_text_head_golden; loss:
        # E2E_LOSS_BUDGET of a batch this small, 8e-3 for the two-clip case), and no further from it than the full block is
        assert err[name][0] < 2e-2 and err[name][0] < 1.5 * err["full"][0] + 2e-3, (name, err)
        assert err[name][1] < (8e-3 if kind == "text2" else 4e-3), (name, err)
        assert err[name][2] < 3e-2 and err[name][2] < 1.5 * err["full"][2] + 2e-3, (name, err)
        # against the full block: two bf16 realisations of the same function (observed: profiles/r5_parity_observed.jsonl: at these batch
        # sizes -- 2 to 6 clips -- the FULL block's own loss error against the oracle is 0.4e-3 ... 5.2e-3, so its distance to any other
        # realisation cannot be smaller; the folded form with bf16 pairs sits at 0.6e-3 ... 3.7e-3 from the oracle)
        assert rel_err(f1, f0) < 8e-3 and abs(l1 - l0) < (8e-3 if kind == "text2" else 4e-3), (name, rel_err(f1, f0), l0, l1)
        # (two clips: the 2 x 2 logits at scale 1 / 0.07 turn the features' rounding into the largest gradient differences seen here)
        for k in g0:
            assert rel_l2(g1[k], g0[k]) < (3e-2 if kind == "text2" else 2e-2), (name, k, rel_l2(g1[k], g0[k]))


@pytest.mark.parametrize("rows", [True, False], ids=["rows", "fullblock"])
@pytest.mark.parametrize("tag,L,b,T,Fq", [("L12", 12, 32, 256, 64), ("cfg2", 12, 64, 1024, 128)])
def test_end_to_end_golden_e4m3(M, golden, tag, L, b, T, Fq, rows):
    """`running.fp8_gemm` (BASELINE.json configs[4]) against the REFERENCE's own outputs, not against the bf16 HIP run: the same
    fixtures as above with e4m3 operands in the eight NT contractions of every block.  The reference has no fp8 path, so the budgets
    are about twice the errors observed on MI355X for this format (per-row power-of-two scales): recorded in
    gpurun_out/parity_observed.jsonl next to the bf16 numbers."""
    g = golden(f"e2e_{tag}")
    head = M.build_audio_head(audio_cfg(T, Fq, L))
    S = head.misc.positional_embedding.shape[0]
    head.load_state_dict(gen.det_weights(f"e2e/{tag}", gen.vit_head_shapes(768, L, 512, S)), strict=True)
    lhead = M.build_loss_head(NS(name="CELossHead", layers=[], scaling=True, scale_max=None))
    head, lhead = head.to(DEV).train(), lhead.to(DEV).train()
    head.encoder.fp8 = True
    # with the last block on its read-out rows that block's contractions run in bf16 (its per-token K / V projection included: see
    # ops.BackboneFn); `fullblock` keeps the e4m3 forward and backward of the last block under the golden vectors too
    head.encoder.last_block_rows = rows
    from vipant_amd import ops
    aud = gen.det_randn(f"e2e/{tag}/aud", (b, 1, T, Fq)).to(DEV)
    img = ops.l2_normalize(gen.det_randn(f"e2e/{tag}/img", (b, 512)).to(DEV))
    feat = head(aud, normalized=True)
    loss = lhead(img, feat, None, normalized=True)
    loss.backward()
    cos = torch.nn.functional.cosine_similarity(feat.detach().double().cpu(), torch.from_numpy(g["feat"]).double(), dim=-1)
    grads = {k: p.grad for k, p in head.named_parameters()}
    keys = list(g["keys"])
    ratio = np.array([float(grads[k].norm()) for k in keys]) / g["gnorm"]
    cls = rel_l2(grads["misc.class_embedding"], g["g_cls"])
    fcb = rel_l2(grads[f"encoder.resblocks.{L - 1}.mlp.c_fc.bias"], g["g_last_fc_bias"])
    observe(f"e2e_{tag}_e4m3[{'rows' if rows else 'fullblock'}]", loss_hip=float(loss), loss_ref=float(g["loss"]), loss_abs_err=abs(float(loss) - float(g["loss"])),
            feat_rel_err=rel_err(feat, g["feat"]), min_cos=float(cos.min()), gnorm_ratio_max_dev=float(np.abs(ratio - 1).max()),
            cls=cls, fc_bias_last=fcb)
    assert abs(float(loss) - float(g["loss"])) < E4M3_BUDGET[tag]["loss"], (float(loss), float(g["loss"]))
    assert rel_err(feat, g["feat"]) < E4M3_BUDGET[tag]["feat"] and float(cos.min()) > 0.99, (rel_err(feat, g["feat"]), float(cos.min()))
    assert np.all(np.abs(ratio - 1) < E4M3_BUDGET[tag]["gnorm"]), (ratio.min(), ratio.max())
    assert cls < E4M3_BUDGET[tag]["grad"] and fcb < E4M3_BUDGET[tag]["grad"], (cls, fcb)


# about 2x the errors observed on MI355X (profiles/r5_parity_observed.jsonl, e2e_*_e4m3; rows / full last block).  Round 5: activations in
# the MX block format (one scale per 32 elements instead of per row): loss 2.7e-3 / 2.1e-3 (L12), 2.9e-3 / 4.9e-3 (cfg2) -- round 4, row
# scales: 6.6e-3 / 4.8e-3 and 5.0e-3 / 1.05e-2; largest feature component 8.3e-2 / 7.4e-2 (cosine >= 0.9955), gradient norms within
# 3 % / 4 %, rel-L2 of the two deepest gradients 0.16 / 0.19
# Round 6 (block-uniform 32 x 32 scales from the epilogues, static LayerNorm scales, e4m3 weight gradients): this ONE batch reads
# 6.4e-3 / 4.4e-3 (L12) where round 5 read 2.7e-3 / 2.1e-3.  That is the draw, not the scheme: over 160 random batches of the L12 shape
# |loss_e4m3 - loss_bf16| is 1.35e-2 mean / 1.67e-2 rms with the round-5 library and 1.24e-2 / 1.54e-2 with this one, 6.7e-3 vs 5.9e-3
# mean at the cfg2 shape (tools/fp8_loss_noise.py, profiles/r6_fp8_loss_noise.txt; both libraries on one box) -- the fixture batch is
# a quiet draw of a quantity whose standard deviation is ~1.5e-2.  The L12 budget follows that statistic (one rms), not one draw x 2.
E4M3_BUDGET = {"L12": {"loss": 1.6e-2, "feat": 2e-1, "gnorm": 6e-2, "grad": 3.5e-1},
               "cfg2": {"loss": 1e-2, "feat": 2e-1, "gnorm": 1e-1, "grad": 3.5e-1}}


def test_trainer_step_matches_oracle_lars():
    """Two Monitor steps on a tiny VA config: loss finite and decreasing bookkeeping, LR schedule values, and the
    fused LARS update equal to the oracle's per-tensor rule applied to the same gradients."""
    from oracle import ref_cpu as R
    from vipant_amd.config import compose
    from vipant_amd.monitor import VAMonitor
    ov = ("+running=bimodal worker=CVALP mode=dp eval=False num_gpus=1 +model/image=vit_val +model/audio=vit_val "
          "+model/text=dummy +model/loss=ce +optimizer=standard +running/audio=default "
          "model.audio.pre_encoder.stride=[16,24] model.image.encoder.layers=2 running.audio.max_len=256 "
          "running.audio.num_mel_bins=64 running.batch_size=16 running.epochs=2 running.frame_emb=synthetic "
          "running.synthetic_steps=2 running.save_epoch=False optimizer.warmup_epoch=1").split()
    cfg = compose(ov)
    cfg.rank = 0
    torch.manual_seed(cfg.seed)
    logs = []
    mon = VAMonitor(cfg, logs.append, torch.device(DEV))
    mon.total_loss = mon.total_step = mon.total_inst = 0
    import time
    mon.start_time = time.time()
    names = [k for k, p in mon.model.named_parameters() if p.requires_grad]
    params = dict(mon.model.named_parameters())
    mus = {k: torch.zeros_like(params[k]).cpu() for k in names}
    step = 0
    for batch in mon.dataloader:
        images, audios, text, _, _ = mon.make_batch(batch)
        from vipant_amd.module import adjust_learning_rate
        adjust_learning_rate(cfg.optimizer, mon.optimizer, mon.dataloader, step)
        lw, lb = R.adjust_learning_rate(step, epochs=2, steps_per_epoch=2, warmup_epoch=1, batch_size=16, lr_weight=0.2,
                                        lr_bias=0.0048)
        assert math.isclose(mon.optimizer.param_groups[0]["lr"], lw) and math.isclose(mon.optimizer.param_groups[1]["lr"], lb)
        before = {k: params[k].detach().cpu().clone() for k in names}
        # run forward/backward by hand so the gradients can be captured before the optimizer consumes them
        mon.optimizer.zero_grad(set_to_none=True)
        loss = mon.model(images, audios, None)
        loss.backward()
        grads = {k: params[k].grad.detach().cpu().clone() for k in names}
        mon.optimizer.step()
        assert math.isfinite(float(loss)) and abs(float(loss) - 2 * math.log(16)) < 1.0
        for k in names:
            p_ref, mus[k] = R.lars_step(before[k], grads[k], mus[k], lw if before[k].ndim > 1 else lb)
            err = float((params[k].detach().cpu() - p_ref).abs().max())
            assert err <= 1e-6 + 1e-5 * float(p_ref.abs().max()), (k, step, err)
        step += 1
    assert step == 2


def test_at_step_with_frozen_text_tower():
    """AT fine-tuning layout (bash/run_bimodal_at.sh): trainable audio head + frozen causal CLIP text tower +
    VALCELossHead(al).  One VALMonitor step against the CPU oracle on the same weights / batch (loss), plus the
    trainer bookkeeping (which parameters moved)."""
    from oracle import ref_cpu as R
    from vipant_amd.config import compose
    from vipant_amd.monitor import VALMonitor
    ov = ("+running=trimodal monitor=VALMonitor worker=CVALP mode=dp eval=False num_gpus=1 +model/image=vit_val "
          "+model/audio=vit_val +model/text=transformer_val +model/loss=ce_val +optimizer=standard +running/audio=default "
          "model.audio.pre_encoder.stride=[16,24] running.siamese.alive=True running.imagine=False model.loss.va=False "
          "model.image.encoder.layers=2 model.text.encoder.layers=2 running.audio.max_len=256 running.audio.num_mel_bins=64 "
          "running.batch_size=8 running.epochs=2 running.synthetic_steps=1 running.save_epoch=False optimizer.warmup_epoch=1").split()
    cfg = compose(ov)
    cfg.rank = 0
    torch.manual_seed(cfg.seed)
    mon = VALMonitor(cfg, lambda *_: None, torch.device(DEV))
    assert mon.model.image_head is None and mon.model.text_head is not None
    batch = next(iter(mon.dataloader))
    images, audios, text, _, _ = mon.make_batch(batch)
    assert list(images.shape[1:]) == [1, 1, 1] and text.dtype == torch.int64 and audios.shape[1:] == (1, 256, 64)
    asd = {k: v.detach().cpu().clone() for k, v in mon.model.audio_head.state_dict().items()}
    tsd = {k: v.detach().cpu().clone() for k, v in mon.model.text_head.state_dict().items()}
    ls = mon.model.loss_head.loss_head_al.logit_scale.detach().cpu().clone()
    from vipant_amd.module import adjust_learning_rate
    adjust_learning_rate(cfg.optimizer, mon.optimizer, mon.dataloader, 1)
    before = {k: v.detach().clone() for k, v in mon.model.named_parameters()}
    loss = mon.step(images, audios, text)
    stride, S, pr = R.vit_position_resolution([256, 64], 32, [16, 24])
    ref = R.cvalp_forward(images.cpu(), audios.cpu(), text.cpu(), audio_sd=asd, text_sd=tsd, loss="valce",
                          scales={"al": ls}, loss_flags=dict(va=False, lv=False, al=True),
                          audio_cfg=dict(width=768, layers=2, stride=stride, position_resolution=pr),
                          text_cfg=dict(width=512, layers=2, ctx_len=77))
    assert abs(float(loss.detach()) - float(ref)) < 5e-3, (float(loss.detach()), float(ref))
    moved = {k for k, v in mon.model.named_parameters() if not torch.equal(v.detach(), before[k])}
    assert all(k.startswith(("audio_head.", "loss_head.")) for k in moved) and len(moved) > 30
    assert "al" in mon.model.report(nstep=1)


def test_train_entry_runs_both_launch_scripts(tmp_path, monkeypatch):
    """train.py with the override strings of run_bimodal_va.sh / run_bimodal_at.sh (shrunk model, synthetic batches):
    two optimisation steps each, checkpoint written in the reference's 4-tuple format."""
    import sys
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    import train as entry
    common = (f"alias_root={tmp_path} model_name=t port=1 num_gpus=1 mode=dp num_proc=2 eval=False verbose=False "
              "+model/image=vit_val +model/audio=vit_val +optimizer=standard +running/audio=default "
              "model.audio.pre_encoder.in_channels=3 model.audio.pre_encoder.stride=[16,24] optimizer.warmup=False "
              "running.audio.norms=[-4.93839311,5.75751113] model.image.encoder.layers=1 running.audio.max_len=256 "
              "running.audio.num_mel_bins=64 running.synthetic_steps=2 running.epochs=1 running.peep_rate=1").split()
    va = ["+running=bimodal", "worker=CVALP", "+model/text=dummy", "+model/loss=ce", "running.batch_size=4"] + common
    entry.train(va)
    at = ["+running=trimodal", "monitor=VALMonitor", "worker=CVALP", "+model/text=transformer_val", "+model/loss=ce_val",
          "running.siamese.alive=True", "running.imagine=False", "model.loss.va=False", "running.batch_size=4",
          "model.text.encoder.layers=1", "model_file=notafile", "+running.rnd_cap=True"] + common
    entry.train(at)
    ck = torch.load(tmp_path / "t" / "00000002.pth", weights_only=False)
    assert len(ck["model"]) == 4 and "encoder.resblocks.0.attn.in_proj_weight" in ck["model"][1]
    assert "loss_head_al.logit_scale" in ck["model"][3] and ck["cfg"]["worker"] == "CVALP"


def test_torch_optim_branch_adam_with_warmup():
    """`optimizer.use_lars=False` (cvap/monitor/cvalp.py:338-342, 185-209): torch.optim.Adam + MultiStepLR from the config,
    linear warm-up over `warmup_steps`, per-batch scheduler stepping after warm-up; forward / backward still the HIP path."""
    from vipant_amd.config import compose
    from vipant_amd.monitor import VAMonitor
    ov = ("+running=bimodal worker=CVALP mode=dp eval=False num_gpus=1 +model/image=vit_val +model/audio=vit_val "
          "+model/text=dummy +model/loss=ce +optimizer=standard +running/audio=default "
          "model.audio.pre_encoder.stride=[16,24] model.image.encoder.layers=1 running.audio.max_len=256 "
          "running.audio.num_mel_bins=64 running.batch_size=8 running.epochs=1 running.frame_emb=synthetic "
          "running.synthetic_steps=3 running.save_epoch=False optimizer.use_lars=False optimizer.warmup_steps=2 "
          "optimizer.batch_sch=True optimizer.steps=[1] optimizer.lr=1e-3").split()
    cfg = compose(ov)
    cfg.rank = 0
    torch.manual_seed(cfg.seed)
    logs = []
    mon = VAMonitor(cfg, logs.append, torch.device(DEV))
    assert isinstance(mon.optimizer, torch.optim.Adam) and isinstance(mon.scheduler, torch.optim.lr_scheduler.MultiStepLR)
    before = {k: v.detach().clone() for k, v in mon.model.named_parameters() if v.requires_grad}
    mon.learn()
    assert mon.total_step == 3 and math.isfinite(float(mon.total_loss))
    warm = [m for m in logs if m.startswith("warmup lr")]
    assert len(warm) == 2 and "5.00e-04" in warm[0] and "1.00e-03" in warm[1], warm       # ratio 1/2, 2/2 of lr = 1e-3
    assert abs(mon.optimizer.param_groups[0]["lr"] - 1e-3 * 0.5) < 1e-12                    # one scheduler step past milestone 1
    moved = sum(int(not torch.equal(v.detach(), before[k])) for k, v in mon.model.named_parameters() if k in before)
    assert moved == len(before)


def _va_step(extra):
    from vipant_amd.config import compose
    from vipant_amd.module import adjust_learning_rate
    from vipant_amd.monitor import VAMonitor
    ov = ("+running=bimodal worker=CVALP mode=dp eval=False num_gpus=1 +model/image=vit_val +model/audio=vit_val "
          "+model/text=dummy +model/loss=ce +optimizer=standard +running/audio=default "
          "model.audio.pre_encoder.stride=[16,24] model.image.encoder.layers=2 running.audio.max_len=256 "
          "running.audio.num_mel_bins=64 running.batch_size=24 running.epochs=2 "
          "running.synthetic_steps=2 running.save_epoch=False optimizer.warmup_epoch=1").split() + extra
    cfg = compose(ov)
    cfg.rank = 0
    torch.manual_seed(cfg.seed)
    mon = VAMonitor(cfg, lambda *_: None, torch.device(DEV))
    images, audios, text, _, _ = mon.make_batch(next(iter(mon.dataloader)))
    adjust_learning_rate(cfg.optimizer, mon.optimizer, mon.dataloader, 1)
    loss = mon.step(images, audios, None)
    return float(loss.detach()), {k: v.detach().clone() for k, v in mon.model.named_parameters() if v.requires_grad}


def test_activation_memory_plans_keep_the_step():
    """`running.recompute_mlp` (the [M, 4D] MLP activations re-made in the backward by one more c_fc contraction) is bit-identical
    to the plain step; `running.micro_batch` (towers 8 clips at a time under ONE 24-way loss, image tower included) gives the
    same loss and the same parameters up to fp32 summation order of the weight gradients."""
    l0, p0 = _va_step([])
    l1, p1 = _va_step(["running.recompute_mlp=True"])
    assert l0 == l1
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
    l2, p2 = _va_step(["running.micro_batch=8", "running.recompute_mlp=True"])
    assert abs(l0 - l2) < 1e-6, (l0, l2)
    for k in p0:
        err = float((p0[k] - p2[k]).abs().max())
        assert err <= 1e-6 + 1e-4 * float(p0[k].abs().max()), (k, err)


def test_no_grad_forward_sees_the_optimizer_update():
    """The fused LARS kernel writes the parameters through raw pointers; torch must still learn that they changed, or the bf16
    weight cache of no-grad forwards keeps serving the first weights it saw.  (Found in round 2: evaluation after training steps
    and the feature pass of `running.micro_batch` read stale weights from the second step on.)  After one optimiser step the loss
    of the updated model is the same through the no-grad path and through the recording path, and a micro-batched run follows the
    one-pass trajectory over three steps."""
    from vipant_amd.config import compose
    from vipant_amd.module import adjust_learning_rate
    from vipant_amd.monitor import VAMonitor
    base = ("+running=bimodal worker=CVALP mode=dp eval=False num_gpus=1 +model/image=vit_val +model/audio=vit_val "
            "+model/text=dummy +model/loss=ce +optimizer=standard +running/audio=default "
            "model.audio.pre_encoder.stride=[16,24] model.image.encoder.layers=2 running.audio.max_len=256 "
            "running.audio.num_mel_bins=64 running.batch_size=24 running.epochs=2 "
            "running.synthetic_steps=2 running.save_epoch=False optimizer.warmup_epoch=1").split()
    traj = {}
    for mb in (0, 8):
        cfg = compose(base + [f"running.micro_batch={mb}"])
        cfg.rank = 0
        torch.manual_seed(cfg.seed)
        mon = VAMonitor(cfg, lambda *_: None, torch.device(DEV))
        images, audios, _, _, _ = mon.make_batch(next(iter(mon.dataloader)))
        with torch.no_grad():
            before = float(mon.model(images, audios, None))              # fills the no-grad weight cache with the initial weights
        losses = []
        for i in range(3):
            adjust_learning_rate(cfg.optimizer, mon.optimizer, mon.dataloader, i + 1)
            losses.append(float(mon.step(images, audios, None).detach()))
            with torch.no_grad():
                after = float(mon.model(images, audios, None))
            recorded = float(mon.model(images, audios, None).detach())
            assert after == recorded, (mb, i, after, recorded)
        assert abs(losses[0] - before) < 1e-6 and after != before, (before, losses, after)
        traj[mb] = losses
    for a, b in zip(traj[0], traj[8]):
        assert abs(a - b) < 2e-5, (traj[0], traj[8])


def test_e4m3_step_under_the_activation_memory_plans():
    """`running.fp8_gemm` through the trainer (VAMonitor.step): the loss stays within the e4m3 noise of the bf16 step, recompute is
    bit-identical, micro-batching equal up to fp32 summation order -- the three switches of the configs[4] bench leg together."""
    l0, _ = _va_step([])
    l1, p1 = _va_step(["running.fp8_gemm=True"])
    assert abs(l1 - l0) < 2e-2 * abs(l0), (l0, l1)
    l2, p2 = _va_step(["running.fp8_gemm=True", "running.recompute_mlp=True"])
    assert l1 == l2
    for k in p1:
        assert torch.equal(p1[k], p2[k]), k
    l3, p3 = _va_step(["running.fp8_gemm=True", "running.micro_batch=8", "running.recompute_mlp=True"])
    assert abs(l1 - l3) < 1e-6, (l1, l3)
    for k in p1:
        err = float((p1[k] - p3[k]).abs().max())
        assert err <= 1e-6 + 1e-4 * float(p1[k].abs().max()), (k, err)


@pytest.mark.parametrize("rows", [True, False], ids=["rows", "fullblock"])
@pytest.mark.parametrize("stream", ["fp16", "fp32"])
@pytest.mark.parametrize("tag", ["va", "at"])
def test_trainer_trajectory_golden(M, golden, monkeypatch, tag, stream, rows):
    """Four optimisation steps of the product path -- heads, loss head, fused LARS, `adjust_learning_rate` -- against the
    trajectory the REFERENCE's own pieces produced on the same weights and batches (tests/golden/make_golden.py, section viii):
    learning rates exactly, every step's loss, every tunable tensor's update norm, and the final small tensors.  All four corners
    of the two round-3 defaults (`running.stream_dtype`, `running.last_block_rows`), so that a loss-error change can be attributed."""
    g = golden(f"traj_{tag}")
    L, b, T, Fq = 2, 8, 256, 64
    head = M.build_audio_head(audio_cfg(T, Fq, L))
    head.encoder.stream_f16 = stream == "fp16"
    head.encoder.last_block_rows = rows
    S = head.misc.positional_embedding.shape[0]
    w0 = gen.det_weights(f"traj/{tag}", gen.vit_head_shapes(768, L, 512, S))
    head.load_state_dict(w0, strict=True)
    if tag == "at":
        thead = M.build_text_head(text_cfg(L))
        thead.load_state_dict(gen.det_weights("traj/text", gen.text_head_shapes(512, L, 512)), strict=True)
        thead = thead.to(DEV).eval()
        for p in thead.parameters():
            p.requires_grad = False
        lhead = M.build_loss_head(NS(name="VALCELossHead", layers=[], scaling=True, scale_max=None, va=False, lv=False, al=True))
    else:
        lhead = M.build_loss_head(NS(name="CELossHead", layers=[], scaling=True, scale_max=None))
    head, lhead = head.to(DEV).train(), lhead.to(DEV).train()
    named = [(f"audio_head.{k}", p) for k, p in head.named_parameters()] + [(f"loss_head.{k}", p) for k, p in lhead.named_parameters()]
    assert [k for k, _ in named] == list(g["keys"])
    params = [p for _, p in named]
    opt = M.LARS([{"params": [p for p in params if p.ndim > 1]}, {"params": [p for p in params if p.ndim < 2]}], lr=0.,
                 weight_decay=1e-6, weight_decay_filter=M.exclude_bias_or_norm, lars_adaptation_filter=M.exclude_bias_or_norm)
    ocfg = NS(epochs=3, warmup_epoch=1, batch_size=b, lr_weight=0.2, lr_bias=0.0048)
    from vipant_amd import ops
    # which entry points this corner really goes through: the read-out-row kernels of the last block must run iff `rows`
    # (the budgets below cannot tell a corner that silently took another corner's path)
    calls = {}
    real_call = ops.call

    counting = [True]          # off while the frozen text tower runs (it keeps its own default last-block mode in every corner)

    def counting_call(name, *a):
        if counting[0]:
            calls[name] = calls.get(name, 0) + 1
        return real_call(name, *a)
    monkeypatch.setattr(ops, "call", counting_call)
    worst_loss, worst_norm, losses, worst_norm_at, worst_step = 0.0, 0.0, [], "", -1
    for step in range(4):
        M.adjust_learning_rate(ocfg, opt, range(2), step)
        assert abs(opt.param_groups[0]["lr"] - float(g["lrs"][step][0])) < 1e-9 and abs(opt.param_groups[1]["lr"] - float(g["lrs"][step][1])) < 1e-10
        before = [p.detach().clone() for p in params]
        opt.zero_grad(set_to_none=True)
        feat = head(gen.det_randn(f"traj/{tag}/aud/{step}", (b, 1, T, Fq)).to(DEV), normalized=True)
        if tag == "at":
            counting[0] = False
            with torch.no_grad():
                tf = thead(gen.det_tokens(f"traj/tok/{step}", b).to(DEV), normalized=True)
            counting[0] = True
            loss = lhead(None, feat, tf, normalized=True)
        else:
            loss = lhead(ops.l2_normalize(gen.det_randn(f"traj/{tag}/img/{step}", (b, 512)).to(DEV)), feat, None, normalized=True)
        loss.backward()
        opt.step()
        if abs(float(loss.detach()) - float(g["losses"][step])) > worst_loss:
            worst_loss, worst_step = abs(float(loss.detach()) - float(g["losses"][step])), step
        losses.append(float(loss.detach()))
        for i, (p, q) in enumerate(zip(params, before)):
            ref = float(g["dnorm"][step][i])
            dn = float((p.detach() - q).norm())
            if ref == 0.0:
                assert dn == 0.0, (step, named[i][0])
            elif abs(dn / ref - 1) > worst_norm:
                worst_norm, worst_norm_at = abs(dn / ref - 1), f"step {step} {named[i][0]}"
    n_rows = sum(n for k, n in calls.items() if "rows_ctx" in k or "mha_rows" in k)
    n_f16 = int(head.encoder.stream_f16)
    assert (n_rows > 0) == bool(rows), (rows, calls)
    TRAJ_CORNERS[(tag, stream, rows)] = (tuple(losses), n_rows)
    observe(f"traj_{tag}[{stream},{'rows' if rows else 'fullblock'}]", worst_loss_err=worst_loss, worst_update_norm_dev=worst_norm,
            worst_loss_step=worst_step, loss_step0=losses[0], loss_step1=losses[1], loss_step2=losses[2], loss_step3=losses[3],
            rows_kernel_calls=n_rows, stream_f16=n_f16, abi_calls=sum(calls.values()))
    print(f"traj_{tag}[{stream},{'rows' if rows else 'fullblock'}]: losses {losses}; worst update-norm deviation at {worst_norm_at}")
    # b = 8, two blocks: the bf16 towers' loss error at this batch size.  Observed on MI355X in round 5 (profiles/r5_parity_observed.jsonl),
    # worst of four steps over the (stream, last-block) corners: VA 1.6e-3 ... 3.2e-3, AT 1.6e-3 (fp32, rows), 2.2e-3 (fp32, full block),
    # 2.5e-3 / 2.6e-3 (fp16 stream); round 4 had 4.05e-3 with both defaults on (bf16 qk / contexts in the folded last block).  One budget
    # for every corner again: 5e-3, the value the (fp32, full block) corner always had.
    assert worst_loss < 5e-3, worst_loss
    assert worst_norm < 6e-2, worst_norm                  # update norms: LARS trust ratio x gradient norm, bf16 gradient noise
    for k, p in named:
        if f"final_{k}" in g.files:
            k0 = k[len("audio_head."):] if k.startswith("audio_head.") else None
            init = w0[k0] if k0 is not None else torch.ones([]) * math.log(1 / 0.07)
            d_ref = torch.from_numpy(g[f"final_{k}"]).double() - init.double()
            d_hip = p.detach().cpu().double() - init.double()
            if float(d_ref.norm()) > 0:
                assert float((d_hip - d_ref).norm() / d_ref.norm()) < 0.25, k      # accumulated update direction of a small tensor


@pytest.mark.parametrize("tag", ["va", "at"])
def test_trajectory_corners_are_distinct(tag):
    """two corners of `test_trainer_trajectory_golden` recorded bit-identical worst errors.  Every corner must
    have run its own path: the four (stream, last-block) corners of a script give four different loss sequences (bitwise -- a corner
    that silently fell back to another's path reproduces that one's losses exactly), and the rows corners alone call the read-out-row
    kernels.  (Equal WORST errors are still possible: the worst step's error is measured against the same reference loss.)"""
    got = {k: v for k, v in TRAJ_CORNERS.items() if k[0] == tag}
    if len(got) < 4:
        pytest.skip("needs the four corners of test_trainer_trajectory_golden in the same session")
    seqs = [v[0] for v in got.values()]
    assert len(set(seqs)) == 4, got
    for (t, stream, rows), (_, n_rows) in got.items():
        assert (n_rows > 0) == rows, (t, stream, rows, n_rows)


def test_frozen_towers_full_depth_golden(M, golden):
    """Both frozen towers at their full 12-block depth against the reference's features (forward-only path: cached bf16 weights,
    one set of temporaries, causal attention with the EOT read-out for the text tower)."""
    g = golden("towers_l12")
    thead = M.build_text_head(text_cfg(12))
    thead.load_state_dict(gen.det_weights("text/l12", gen.text_head_shapes(512, 12, 512)), strict=True)
    ihead = M.build_image_head(image_cfg(12))
    ihead.load_state_dict(gen.det_weights("img/l12", gen.vit_head_shapes(768, 12, 512, 50)), strict=True)
    thead, ihead = thead.to(DEV).eval(), ihead.to(DEV).eval()
    with torch.no_grad():
        tf = thead(gen.det_tokens("text/l12/tok", 6).to(DEV), normalized=True)
        ts = thead(gen.det_tokens("text/l12/short", 5, L=40)[:, :40].contiguous().to(DEV), normalized=True)
        imf = ihead(gen.det_randn("img/l12/x", (3, 3, 224, 224)).to(DEV), normalized=True)
    errs = dict(text=rel_err(tf, g["text_feat"]), text_short=rel_err(ts, g["text_feat_short"]), image=rel_err(imf, g["image_feat"]))
    observe("towers_l12", **errs)
    assert max(errs.values()) < 3e-2, errs                     # 12 bf16 blocks: observed ~1 % of the largest component
    for got, ref in ((tf, g["text_feat"]), (ts, g["text_feat_short"]), (imf, g["image_feat"])):
        cos = torch.nn.functional.cosine_similarity(got.double().cpu(), torch.from_numpy(ref).double(), dim=-1)
        assert float(cos.min()) > 0.999, float(cos.min())
