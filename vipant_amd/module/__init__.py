"""Operator API of the reference's `cvap.module` package (cvap/module/__init__.py), MI355X-native."""
from .val import (  # noqa: F401
    ENCODER_MODULES_REGISTRY, build_encoder_module, interp_clip_vp_embedding, interp_conv_weight_channel,
    interp_conv_weight_spatial, AddonEncoder, CLIPMisc, GPTPreEncoder, GPTPostEncoder, ViTPreEncoder, ViTPostEncoder,
    TransformerBackbone, ResidualAttentionBlock, LayerNorm, QuickGELU,
)
from .lars import *  # noqa: F401,F403
from .heads import (  # noqa: F401
    IMAGE_HEADS_REGISTRY, AUDIO_HEADS_REGISTRY, TEXT_HEADS_REGISTRY, build_image_head, build_audio_head,
    build_text_head, MetaHead, CLIPImageHead, CLIPAudioHead, CLIPTextHead, DummyHead, position_resolution,
    load_pos_embedding,
)
from .loss_head import (  # noqa: F401
    LOSS_HEADS_REGISTRY, build_loss_head, LossHead, CELossHead, VALCELossHead, DummyLossHead, zero_shot_report,
)

LOSS_HEADS_REGISTRY._do_register("DummyHead", DummyLossHead)
