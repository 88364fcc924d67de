"""Host-side mirror of the reference's ``sac_cbf_clf`` package (agent, models,
dynamics glue, replay) with the update path running on MI355X HIP kernels.

Put this package's parent directory on ``sys.path`` and the reference-shaped
driver (``from sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF`` ...) resolves here.
"""
