"""nlbac_amd — MI355X-native hot path of NLBAC (neural-ODE rollout + SAC/CLF/CBF update).

Importing the package is cheap; the HIP library is loaded on first use by
``nlbac_amd._lib`` and its absence is a hard error (there is no CPU fallback).
"""
__version__ = "0.1.0"
