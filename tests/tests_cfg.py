from types import SimpleNamespace as NS


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
