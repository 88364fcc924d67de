"""``SAC_CBF_CLF`` of the learned-barrier-certificate copies
(NU = neural_barrier_certificate/neural_barrier_certificate_NLBAC_Unicycle_RL_training/Unicycle_RL_training/
sac_cbf_clf/sac_cbf_clf.py:26-493): same constructor and methods; ``update_parameters`` consumes the 11-field
replay rows; ``save_model`` / ``load_weights`` / ``load_model`` also handle ``barrier.pkl``.  The device update is
the shared one (``nlbac_amd.sac_cbf_clf.sac_cbf_clf``) with the ``*Barrier`` task."""
from ...sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF as _Base


class SAC_CBF_CLF(_Base):
    variant = "Barrier"
