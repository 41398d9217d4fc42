from vipant_amd.util import *  # noqa: F401,F403
from vipant_amd.util import seed_all_rng, setup_logger, numel, detect_nan, AverageMeter  # noqa: F401
