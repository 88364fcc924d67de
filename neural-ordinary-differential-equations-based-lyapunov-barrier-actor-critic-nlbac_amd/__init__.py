"""nlbac_amd — MI355X-native hot path of NLBAC (neural-ODE rollout + SAC/CLF/CBF update).

Importing the package is cheap; the HIP library is loaded on first use by
``nlbac_amd._lib`` and its absence is a hard error (there is no CPU fallback).
"""
__version__ = "0.1.0"


def install_reference_names(barrier=False):
    """Make the reference drivers' own import lines resolve to this build: after this call
    ``from sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF``, ``from sac_cbf_clf.replay_memory import ReplayMemory``,
    ``from sac_cbf_clf.dynamics import DynamicsModel``, ``from sac_cbf_clf.utils import ...`` and
    ``from sac_cbf_clf.model import ...`` (``U/main.py:6-10`` and the same lines of the other copies) import the
    modules of ``nlbac_amd.sac_cbf_clf`` — or, with ``barrier=True``, of
    ``nlbac_amd.neural_barrier_certificate.sac_cbf_clf`` (the learned-barrier drivers ``NU/main.py``,
    ``NP/main.py``).  Call it before the driver's imports run."""
    import importlib
    import sys
    base = __name__ + (".neural_barrier_certificate.sac_cbf_clf" if barrier else ".sac_cbf_clf")
    sys.modules["sac_cbf_clf"] = importlib.import_module(base)
    for sub in ("sac_cbf_clf", "replay_memory", "dynamics", "model", "utils"):
        sys.modules["sac_cbf_clf." + sub] = importlib.import_module(base + "." + sub)
    return sys.modules["sac_cbf_clf"]
