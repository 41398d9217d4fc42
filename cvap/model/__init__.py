from vipant_amd.model import *  # noqa: F401,F403
from vipant_amd.model import VAL_MODELS_REGISTRY, build_main_model, CVALP  # noqa: F401
