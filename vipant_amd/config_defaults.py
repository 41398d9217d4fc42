"""Default run configuration: the nine config files the two reference launch scripts compose
(SURVEY.md section 5.6), restated as plain dicts with the reference's key names so that the override strings
of bash/run_bimodal_va.sh:22-33 and bash/run_bimodal_at.sh:25-43 apply unchanged.  Only the keys the
contrastive training step reads are kept; paths default to local, dataset-free (synthetic) operation.

ROOT            <- configs/default.yaml
GROUPS[g][opt]  <- configs/<g>/<opt>.yaml
"""

ROOT = {
    "alias_root": "./outputs", "model_root": "./outputs", "model_name": "test", "model_file": "notafile",
    "blockprint": False, "monitor": "VAMonitor", "worker": "CVAP", "verbose": False, "seed": 1213,
    "eval": True, "rank": -1, "mode": "ddp", "num_proc": 0, "num_gpus": 4, "port": 22829,
    "dist_url": "tcp://localhost:${port}",
}

_siamese = {"alive": False, "keep_hp": True, "amodules": [], "lmodules": []}

_running_common = {
    "clip_model_root": "./clip", "clip_model_name": "ViT-B32", "data_root": "", "peep_rate": 1, "save_rate": 1e9,
    "epochs": 1000, "save_epoch": True, "frame_key": "frame", "frame_emb": None, "text_emb": None, "imagine": True,
    "embed_dim": "${model.image.embed_dim}", "resolution": "${model.image.resolution}",
    "max_audio_len": "${running.audio.max_len}", "num_mel_bins": "${running.audio.num_mel_bins}",
    "train_samples": 1.0, "test_samples": 5000,
    # dataset-free operation (this build): synthetic spectrogram / token batches, SURVEY.md 8-D2
    "synthetic": True, "synthetic_steps": 8, "precomputed_image": False,
    # activation-memory plan for batches that do not fit: towers in micro-batches under one global-batch loss,
    # and / or the MLP activations recomputed in the backward
    "micro_batch": 0, "recompute_mlp": False, "fp8_gemm": False, "stream_dtype": "fp16", "grad_stream": None, "last_block_rows": True,
    # replicas: when a block's gradient bucket is reduced (vipant_amd/parallel.py GradSync): block | step
    "comm_overlap": "block",
}

GROUPS = {
    "running": {
        "bimodal": dict(_running_common, **{
            "data_name": "src_unbalanced_train_segments", "eval_name": "src_balanced_train_segments", "test_name": "",
            "eval_samples": 5184, "batch_size": 432, "multi_view": False, "siamese": dict(_siamese),
        }),
        "trimodal": dict(_running_common, **{
            "prompt": "the sound of", "cat_label": False, "filter_set": None, "label_map": "ontology,eval_segments",
            "data_name": "", "eval_name": "", "test_name": "", "eval_samples": 250, "test_samples": 250,
            "batch_size": 64, "force_npz": False, "clf": False, "np_rnd": False, "mixup_rate": 0.0,
            "weighted_sampling": False, "siamese": dict(_siamese),
        }),
    },
    "running/audio": {
        "default": {
            "max_len": 1000, "norms": [], "eval_norms": False, "normalized": False, "dither": 0.0, "tile_audio": False,
            "frame_shift": 10, "htk_compat": True, "use_energy": False, "window_type": "hanning", "num_mel_bins": 128,
            "zero_mean_wf": True, "transform_audio": False, "audio_transforms": [], "transform_fbank": True,
            "fbank_transforms": [["FrequencyMasking", [32]], ["TimeMasking", [200]]],
        },
    },
    "model/image": {
        "vit_val": {
            "name": "CLIPImageHead", "freeze": True, "from_scratch": False, "width": 768, "embed_dim": 512,
            "resolution": 224, "ctx_len": "${model.text.ctx_len}",
            "encoder": {"name": "TransformerBackbone", "layers": 12, "skip_attn_mask": True},
            "pre_encoder": {"name": "ViTPreEncoder", "patch_size": 32,
                            "stride": "${model.image.pre_encoder.patch_size}", "in_channels": 3},
            "post_encoder": {"name": "ViTPostEncoder"}, "misc": {"name": "CLIPMisc"},
            "pre_encoder_addon": {"name": "AddonEncoder"}, "post_encoder_addon": {"name": "AddonEncoder"},
        },
    },
    "model/audio": {
        "vit_val": {
            "name": "CLIPAudioHead", "freeze": False, "from_scratch": False, "width": "${model.image.width}",
            "embed_dim": "${model.image.embed_dim}",
            "resolution": ["${running.max_audio_len}", "${running.num_mel_bins}"], "ctx_len": "${model.text.ctx_len}",
            "encoder": {"name": "TransformerBackbone", "layers": "${model.image.encoder.layers}", "skip_attn_mask": True},
            "pre_encoder": {"name": "ViTPreEncoder", "patch_size": "${model.image.pre_encoder.patch_size}",
                            "stride": [16, 16], "in_channels": 3},
            "post_encoder": {"name": "ViTPostEncoder"}, "misc": {"name": "CLIPMisc"},
            "pre_encoder_addon": {"name": "AddonEncoder"}, "post_encoder_addon": {"name": "AddonEncoder"},
        },
    },
    "model/text": {
        "dummy": {"name": "DummyHead", "freeze": True, "from_scratch": True, "ctx_len": None},
        "transformer_val": {
            "name": "CLIPTextHead", "freeze": True, "from_scratch": False, "from_text": True, "width": 512,
            "embed_dim": "${model.image.embed_dim}", "resolution": None, "ctx_len": 77,
            "encoder": {"name": "TransformerBackbone", "layers": 12, "skip_attn_mask": False},
            "pre_encoder": {"name": "GPTPreEncoder", "vocab_size": 49408},
            "post_encoder": {"name": "GPTPostEncoder"}, "misc": {"name": "CLIPMisc"},
            "pre_encoder_addon": {"name": "AddonEncoder"}, "post_encoder_addon": {"name": "AddonEncoder"},
        },
    },
    "model/loss": {
        "ce": {"name": "CELossHead", "layers": [], "scaling": True, "scale_max": None},
        "ce_val": {"name": "VALCELossHead", "layers": [], "scaling": True, "scale_max": None,
                   "va": True, "lv": False, "al": True},
    },
    "optimizer": {
        "standard": {
            "use_lars": True, "name": "Adam", "warmup": True, "warmup_steps": 1000, "warmup_epoch": 10, "lr": 5e-4,
            "weight_decay": 1e-6, "betas": [0.9, 0.999], "max_norm": 0.5, "lr_weight": 0.2, "lr_bias": 0.0048,
            "batch_size": "${running.batch_size}", "epochs": "${running.epochs}", "steps": [], "gamma": 0.5,
            "batch_sch": False,
            "optimizer": ["Adam", {"lr": "${optimizer.lr}", "betas": "${optimizer.betas}",
                                   "weight_decay": "${optimizer.weight_decay}"}],
            "scheduler": ["MultiStepLR", {"milestones": "${optimizer.steps}", "gamma": "${optimizer.gamma}"}],
        },
    },
}
