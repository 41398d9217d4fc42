from vipant_amd.monitor import *  # noqa: F401,F403
