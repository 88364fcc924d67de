"""Same network containers as the base package plus ``BarrierNetwork`` (NU/sac_cbf_clf/model.py:67-84)."""
from ...sac_cbf_clf.model import (BarrierNetwork, GaussianPolicy, LyaNetwork, NeuralODEModel,  # noqa: F401
                                  QNetwork, train_step)
