"""Helper functions of the reference's ``utils.py`` (same as the base package)."""
from ...sac_cbf_clf.utils import *  # noqa: F401,F403
from ...sac_cbf_clf import utils as _u

globals().update({k: getattr(_u, k) for k in dir(_u) if not k.startswith("__")})
