"""Import alias: ``import nlbac_amd`` loads the package that lives in the
(non-identifier) directory
``neural-ordinary-differential-equations-based-lyapunov-barrier-actor-critic-nlbac_amd/``.

The directory name is fixed by the build contract; Python cannot import a
name with hyphens, so this shim registers that directory as the package
``nlbac_amd`` (sub-modules resolve inside it as usual).
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(
    os.path.dirname(os.path.abspath(__file__)),
    "neural-ordinary-differential-equations-based-lyapunov-barrier-actor-critic-nlbac_amd",
)

_spec = importlib.util.spec_from_file_location(
    "nlbac_amd",
    os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR],
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nlbac_amd"] = _mod
_spec.loader.exec_module(_mod)
