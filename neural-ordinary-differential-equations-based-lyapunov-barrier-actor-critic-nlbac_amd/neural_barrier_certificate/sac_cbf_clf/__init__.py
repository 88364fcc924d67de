"""Mirror of ``neural_barrier_certificate/*/sac_cbf_clf`` (same module names as the reference package)."""
