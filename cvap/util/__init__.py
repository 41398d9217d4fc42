"""Import-path alias of the reference package name."""
from vipant_amd.util import seed_all_rng, setup_logger  # noqa: F401
