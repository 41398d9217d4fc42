"""`cvap.model` surface: model registry + builder (cvap/model/__init__.py:12-26)."""
from ..registry import Registry
from .helper import *  # noqa: F401,F403
from .cvalp import CVALP

VAL_MODELS_REGISTRY = Registry("VAL_MODELS")
VAL_MODELS_REGISTRY.__doc__ = "Registry for vision-audio-language models."
VAL_MODELS_REGISTRY.register(CVALP)


def build_main_model(cfg, echo, **kwargs):
    return VAL_MODELS_REGISTRY.get(cfg.worker)(cfg, echo)
