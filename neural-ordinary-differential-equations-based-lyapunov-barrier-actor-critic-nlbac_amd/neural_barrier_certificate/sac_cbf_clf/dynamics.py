"""obs <-> state glue (same as the base package; NU/sac_cbf_clf/dynamics.py:12-70)."""
from ...sac_cbf_clf.dynamics import DynamicsModel  # noqa: F401
