"""Host logic on CPU: override grammar of the two launch scripts, registries / builders, state-dict key contract,
from-scratch initialisation under the run seed, CLIP weight remapping and positional-table surgery, LR schedule,
trainer parameter grouping -- against golden values produced by the imported reference where arithmetic is involved."""
import math
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

import gen
import vipant_amd.module as M
from vipant_amd.config import compose
from vipant_amd.model import build_main_model, VAL_MODELS_REGISTRY

VA = ("+running=bimodal model_name=test worker=CVALP port=1234 num_gpus=1 mode=dp num_proc=2 eval=False verbose=True "
      "+model/image=vit_val +model/audio=vit_val +model/text=dummy +model/loss=ce +optimizer=standard +running/audio=default "
      "model.audio.pre_encoder.in_channels=3 model.audio.pre_encoder.stride=[16,24] "
      "optimizer.warmup=False running.audio.norms=[-4.93839311,5.75751113] "
      "running.epochs=1 running.batch_size=2 running.peep_rate=50 running.save_rate=100 running.eval_samples=100").split()
AT = ("+running=trimodal model_name=test monitor=VALMonitor worker=CVALP port=1234 num_gpus=1 mode=dp num_proc=8 eval=False verbose=True "
      "+model/image=vit_val +model/audio=vit_val +model/text=transformer_val +model/loss=ce_val +optimizer=standard +running/audio=default "
      "model.audio.pre_encoder.in_channels=3 model.audio.pre_encoder.stride=[16,24] "
      "optimizer.warmup=False running.audio.norms=[-4.93839311,5.75751113] "
      "running.siamese.alive=True running.imagine=False model.loss.va=False "
      "running.batch_size=64 running.peep_rate=1 running.prompt=\"\" model_file=notafile +running.rnd_cap=True "
      "running.data_name=audiocaps_train running.eval_name=audiocaps_val running.test_name=audiocaps_test "
      "running.eval_samples=250 running.test_samples=250 running.train_samples=0.1").split()


def cfg_ns(T, Fq, layers):
    return NS(name="CLIPAudioHead", width=768, embed_dim=512, resolution=[T, Fq], ctx_len=77,
              encoder=NS(name="TransformerBackbone", layers=layers, skip_attn_mask=True),
              pre_encoder=NS(name="ViTPreEncoder", patch_size=32, stride=[16, 24], in_channels=3),
              post_encoder=NS(name="ViTPostEncoder"), misc=NS(name="CLIPMisc"),
              pre_encoder_addon=NS(name="AddonEncoder"), post_encoder_addon=NS(name="AddonEncoder"))


def test_va_script_overrides():
    """bash/run_bimodal_va.sh:22-33 of the reference, verbatim."""
    cfg = compose(VA)
    assert cfg.worker == "CVALP" and cfg.monitor == "VAMonitor" and cfg.mode == "dp" and cfg.seed == 1213
    assert cfg.model.audio.name == "CLIPAudioHead" and cfg.model.audio.width == 768 and cfg.model.audio.embed_dim == 512
    assert cfg.model.audio.resolution == [1000, 128] and cfg.model.audio.pre_encoder.stride == [16, 24]
    assert cfg.model.audio.pre_encoder.patch_size == 32 and cfg.model.audio.encoder.layers == 12
    assert cfg.model.image.pre_encoder.stride == 32 and cfg.model.image.freeze and not cfg.model.audio.freeze
    assert cfg.model.text.name == "DummyHead" and cfg.model.audio.ctx_len is None
    assert cfg.model.loss.name == "CELossHead" and cfg.model.loss.scale_max is None
    assert cfg.optimizer.use_lars and cfg.optimizer.batch_size == 2 and cfg.optimizer.epochs == 1
    assert cfg.running.audio.norms == [-4.93839311, 5.75751113] and cfg.dist_url == "tcp://localhost:1234"
    with pytest.raises(KeyError):
        compose(VA + ["running.no_such_key=1"])
    with pytest.raises(KeyError):
        compose(["+model/audio=nope"])


def test_at_script_overrides():
    """bash/run_bimodal_at.sh:25-43 of the reference, verbatim."""
    cfg = compose(AT)
    assert cfg.monitor == "VALMonitor" and cfg.running.siamese.alive and not cfg.running.imagine
    assert cfg.model.loss.name == "VALCELossHead" and (cfg.model.loss.va, cfg.model.loss.lv, cfg.model.loss.al) == (False, False, True)
    assert cfg.model.text.name == "CLIPTextHead" and cfg.model.text.ctx_len == 77 and cfg.model.audio.ctx_len == 77
    assert cfg.running.rnd_cap is True and cfg.running.prompt == "" and cfg.running.batch_size == 64
    assert cfg.model.text.freeze and cfg.model.text.pre_encoder.vocab_size == 49408


def test_registries_and_builders():
    for reg, names in ((M.ENCODER_MODULES_REGISTRY, ["ViTPreEncoder", "ViTPostEncoder", "GPTPreEncoder", "GPTPostEncoder",
                                                      "TransformerBackbone", "CLIPMisc", "AddonEncoder"]),
                       (M.AUDIO_HEADS_REGISTRY, ["CLIPAudioHead", "DummyHead"]), (M.IMAGE_HEADS_REGISTRY, ["CLIPImageHead", "DummyHead"]),
                       (M.TEXT_HEADS_REGISTRY, ["CLIPTextHead", "DummyHead"]), (M.LOSS_HEADS_REGISTRY, ["CELossHead", "VALCELossHead", "DummyHead"])):
        for n in names:
            assert reg.get(n).__name__ in (n, "DummyLossHead")
    assert VAL_MODELS_REGISTRY.get("CVALP").__name__ == "CVALP"
    with pytest.raises(KeyError):
        M.AUDIO_HEADS_REGISTRY.get("NaiveCLIPAudioHead")
    import cvap.module as cm        # import-path compatibility
    assert cm.build_audio_head is M.build_audio_head and cm.LARS is M.LARS


def test_state_dict_keys_and_seeded_init_match_reference(golden):
    """Same keys, shapes and -- because parameter containers are built in the reference's order with torch's default
    initialisers -- the SAME from-scratch values under seed 1213 (configs/default.yaml:9)."""
    g = golden("init_seed1213")
    from tests_cfg import image_cfg, text_cfg  # noqa: F401
    for tag, builder, cfg in (("audio", M.build_audio_head, cfg_ns(256, 64, 2)), ("image", M.build_image_head, image_cfg(1)),
                              ("text", M.build_text_head, text_cfg(1))):
        torch.manual_seed(1213)
        sd = builder(cfg).state_dict()
        assert list(sd.keys()) == list(g[f"{tag}_keys"]), tag
        assert [str(tuple(v.shape)) for v in sd.values()] == list(g[f"{tag}_shapes"])
        sums = np.array([gen.checksum(v) for v in sd.values()])
        assert np.allclose(sums, g[f"{tag}_sums"], rtol=1e-6, atol=1e-6), tag
    lh = M.build_loss_head(NS(name="CELossHead", layers=[], scaling=True, scale_max=None))
    assert list(lh.state_dict().keys()) == list(g["loss_keys"]) and math.isclose(float(lh.logit_scale), float(g["logit_scale"]), rel_tol=1e-6)
    vh = M.build_loss_head(NS(name="VALCELossHead", layers=[], scaling=True, scale_max=None, va=False, lv=False, al=True))
    assert list(vh.state_dict().keys()) == ["loss_head_al.logit_scale"]


def test_clip_weight_remap_and_pos_interpolation(golden):
    g = golden("init_remap")
    old = gen.det_randn("interp/pos50", (50, 64))
    for key, shape in (("pos_15x2", (15, 2)), ("pos_63x5", (63, 5)), ("pos_same", (7, 7))):
        assert np.allclose(M.interp_clip_vp_embedding(old, shape).numpy(), g[key], atol=1e-6)
    old_a = gen.det_randn("interp/pos306", (61 * 5 + 1, 64))
    for nm, new_shape in (("slice_same_extra0", (61, 5)), ("slice_31x5", (31, 5)), ("interp_40x3", (40, 3))):
        od, nd = {}, {"misc.positional_embedding": torch.zeros(int(np.prod(new_shape)) + 1, 64)}
        M.load_pos_embedding({"misc.positional_embedding": old_a.clone()}, od, nd, "misc.positional_embedding", 1, (61, 5), new_shape)
        assert np.allclose(nd["misc.positional_embedding"].numpy(), g["lpe_" + nm], atol=1e-6), nm
    # copy_state_dict from a CLIP-visual-shaped state dict (keys from the reference's VisualTransformer)
    src_keys = list(g["csd_src_keys"])
    ah = M.build_audio_head(cfg_ns(256, 64, 1))
    ref_sd = ah.state_dict()
    fake = {}
    for k in src_keys:
        if k == "conv1.weight":
            fake[k] = gen.det_randn("csd/conv", (768, 3, 32, 32))
        elif k == "positional_embedding":
            fake[k] = gen.det_randn("csd/pos", (50, 768))
        elif k == "class_embedding":
            fake[k] = gen.det_randn("csd/cls", (768,))
        elif k == "proj":
            fake[k] = gen.det_randn("csd/proj", (768, 512))
        else:
            kk = k.replace("transformer.", "encoder.").replace("ln_pre.", "pre_encoder.ln.").replace("ln_post.", "post_encoder.ln.")
            fake[k] = gen.det_randn("csd/" + k, tuple(ref_sd[kk].shape))
    n_o, o_n = ah.copy_state_dict({k: v.clone() for k, v in fake.items()})
    assert sorted(n_o) == list(g["csd_missing"]) and sorted(o_n) == list(g["csd_unexpected"])
    sd = ah.state_dict()
    assert sorted(sd.keys()) == list(g["csd_keys"])
    assert torch.equal(sd["pre_encoder.ln.weight"], fake["ln_pre.weight"]) and torch.equal(sd["post_encoder.proj"], fake["proj"])
    assert torch.equal(sd["encoder.resblocks.0.attn.in_proj_weight"], fake["transformer.resblocks.0.attn.in_proj_weight"])
    assert torch.equal(sd["pre_encoder.conv1.weight"], fake["conv1.weight"])          # 3 stored channels kept
    assert torch.allclose(sd["misc.positional_embedding"], M.interp_clip_vp_embedding(fake["positional_embedding"], (15, 2)))
    assert sd["misc.positional_embedding"].shape == (31, 768)


def test_lr_schedule_matches_reference(golden):
    g = golden("lars")
    ocfg = NS(epochs=3, warmup_epoch=1, batch_size=64, lr_weight=0.2, lr_bias=0.0048)
    opt = NS(param_groups=[{"lr": 0.0}, {"lr": 0.0}])
    for step in range(12):
        M.adjust_learning_rate(ocfg, opt, list(range(5)), step)
        assert np.allclose([opt.param_groups[0]["lr"], opt.param_groups[1]["lr"]], g[f"lr_{step}"], rtol=1e-12)
    assert M.exclude_bias_or_norm(torch.zeros(3)) and not M.exclude_bias_or_norm(torch.zeros(3, 3))


def test_worker_build_and_tunable_sets():
    """cvap/model/cvalp.py:217-267 (VA) and :130-215 (AT): which parameters train."""
    cfg = compose(VA + ["model.image.encoder.layers=1", "running.audio.max_len=256", "running.audio.num_mel_bins=64"])
    cfg.rank = 0
    logs = []
    model = build_main_model(cfg, logs.append)
    tun = model.build()
    assert model.text_head is None and any("Destory text encoder" in m for m in logs)
    assert set(k.split(".")[0] for k in tun) == {"loss_head", "audio_head"} and "loss_head.logit_scale" in tun
    assert len([k for k in tun if k.startswith("audio_head.")]) == len(list(model.audio_head.parameters()))
    st = model.collect_state_dict()
    assert len(st) == 4 and len(st[0]) == 0 and len(st[2]) == 0 and "logit_scale" in st[3]
    cfg = compose(AT + ["model.image.encoder.layers=1", "model.text.encoder.layers=1", "running.audio.max_len=256",
                        "running.audio.num_mel_bins=64"])
    cfg.rank = 0
    model = build_main_model(cfg, logs.append)
    tun = model.build()
    assert model.image_head is None and model.text_head is not None
    assert "loss_head.loss_head_al.logit_scale" in tun and not any(k.startswith("text_head") for k in tun)
    assert model.text_head.misc.positional_embedding.shape == (78, 512)
    assert model.text_head.encoder.causal and not model.audio_head.encoder.causal


def test_patch_grid_geometry():
    assert M.position_resolution([1000, 128], 32, [16, 24]) == (61, 5)
    assert M.position_resolution([1024, 128], 32, [16, 24]) == (63, 5)
    assert M.position_resolution(224, 32, None) == (7, 7)
    h = M.build_audio_head(cfg_ns(1024, 128, 1))
    assert h.misc.positional_embedding.shape == (316, 768) and tuple(h.misc.position_resolution) == (63, 5)


def test_checkpoint_round_trip_into_at_build_and_eval(tmp_path):
    """VA pre-training checkpoint (Monitor.save: cvap/monitor/cvalp.py:302-309 format) -> the AT fine-tuning build
    (`from_pretrained`, cvalp.py:130-215 / clip_head.py:172-191) and the eval build (cvalp.py:105-120): the stored run config
    must come back with attribute access and the audio tower must carry the saved weights, including the re-gridded
    positional table when the clip length changes."""
    from vipant_amd.config import to_plain
    from vipant_amd.model.helper import load_checkpoint
    small = ["model.image.encoder.layers=1", "running.audio.max_len=256", "running.audio.num_mel_bins=64",
             f"model_root={tmp_path}", f"alias_root={tmp_path}", "model_name=ck"]
    cfg = compose(VA + small)
    cfg.rank = 0
    torch.manual_seed(5)
    va = build_main_model(cfg, lambda *_: None)
    va.build()
    (tmp_path / "ck").mkdir()
    torch.save({"cfg": to_plain(cfg), "model": va.collect_audio_state_dict()}, tmp_path / "ck" / "00000002.pth")   # Monitor.save

    at_cfg = compose(AT + small + ["model.text.encoder.layers=1", "model_file=00000002.pth"])
    at_cfg.rank = 0
    local_cfg, _, audio_sd, _, loss_sd = load_checkpoint(at_cfg, lambda *_: None)
    assert local_cfg.model.audio.resolution == [256, 64] and local_cfg.running.audio.max_len == 256
    logs = []
    at = build_main_model(at_cfg, logs.append)
    at.build()
    assert any("Initialize audio encoder from `audio_head`" in m for m in logs), logs
    for k, v in va.audio_head.state_dict().items():
        assert torch.equal(at.audio_head.state_dict()[k], v), k

    # a longer clip at fine-tuning time: the time axis of the positional grid is re-sliced / interpolated, everything else copied
    at_cfg = compose(AT + small + ["model.text.encoder.layers=1", "model_file=00000002.pth", "running.audio.max_len=512"])
    at_cfg.rank = 0
    at = build_main_model(at_cfg, lambda *_: None)
    at.build()
    assert at.audio_head.misc.positional_embedding.shape[0] == 31 * 2 + 1
    k = "encoder.resblocks.0.mlp.c_fc.weight"
    assert torch.equal(at.audio_head.state_dict()[k], va.audio_head.state_dict()[k])

    ev_cfg = compose(VA + small + ["model_file=00000002.pth", "eval=True"])
    ev_cfg.rank = 0
    ev = build_main_model(ev_cfg, lambda *_: None)
    ev.build()
    assert torch.equal(ev.audio_head.state_dict()[k], va.audio_head.state_dict()[k])
    assert torch.equal(ev.loss_head.logit_scale, va.loss_head.logit_scale)

    # a file that exists but cannot be read is an error, not a silent from-scratch run
    (tmp_path / "ck" / "broken.pth").write_bytes(b"not a checkpoint")
    bad = compose(AT + small + ["model_file=broken.pth"])
    bad.rank = 0
    with pytest.raises(Exception):
        build_main_model(bad, lambda *_: None).build()


def test_fp8_switch_is_off_by_default_and_composes():
    """`running.fp8_gemm` (BASELINE.json configs[4]) defaults to False -- the headline configuration is bf16 -- and parses as a bool."""
    from vipant_amd.config import compose
    base = ("+running=bimodal worker=CVALP mode=dp eval=False +model/image=vit_val +model/audio=vit_val +model/text=dummy "
            "+model/loss=ce +optimizer=standard +running/audio=default").split()
    assert compose(base).running.get("fp8_gemm", None) is False
    assert compose(base + ["running.fp8_gemm=True"]).running.fp8_gemm is True


def test_bench_flop_count_follows_the_last_block_form(monkeypatch):
    """`bench.py` counts the work the step does: full last block > last block on its read-out rows with the K / V projection of
    every token (round 3) > the folded form (round 4: no per-token projection; SURVEY.md 8-D4 for the full block)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    S, D, Lyr = 316, 768, 12
    full = bench.tower_fwd_flops(S, 1024, S - 1, width=D, layers=Lyr)
    assert full == Lyr * S * (24 * D * D + 4 * S * D) + 2 * (S - 1) * 1024 * D + 2 * D * 512
    monkeypatch.setenv("VIPANT_LAST_BLOCK_CTX", "0")
    kv = bench.tower_fwd_flops(S, 1024, S - 1, width=D, layers=Lyr, last_block_rows=True)
    monkeypatch.delenv("VIPANT_LAST_BLOCK_CTX")
    folded = bench.tower_fwd_flops(S, 1024, S - 1, width=D, layers=Lyr, last_block_rows=True)
    assert full - kv == S * (24 * D * D + 4 * S * D) - (4 * S * D * D + 20 * D * D + 4 * S * D)
    assert kv - folded == (4 * S * D * D + 20 * D * D + 4 * S * D) - (24 * D * D + 4 * S * (D // 64) * D)
    assert 0.012 < (kv - folded) / kv < 0.015                     # 1.4 % of a 12-block tower at S = 316
    # a width or a length the folded kernels do not take is counted in the round-3 form
    assert bench.tower_fwd_flops(2000, 1024, 1999, width=D, layers=2, last_block_rows=True) == \
        S * 0 + (2 - 1) * 2000 * (24 * D * D + 4 * 2000 * D) + (4 * 2000 * D * D + 20 * D * D + 4 * 2000 * D) + 2 * 1999 * 1024 * D + 2 * D * 512
