"""Import shim: the package directory is `snark-verifier_amd/` (the repo's
naming contract), which is not a valid Python identifier.  `import
snark_verifier_amd` loads that directory as this module."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "snark-verifier_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_d, "__init__.py"), submodule_search_locations=[_d]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
