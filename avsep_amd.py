"""Import alias: ``import avsep_amd`` == the package directory
``audio-visual-sepatation-in-visual-agnostic-situtation_amd`` (whose name is not a Python identifier).

Every submodule the package has loaded is registered under the alias too, so ``from avsep_amd.models import
fusion_net`` returns THE module object of the package instead of importing a second copy (two copies of ``lib`` would
mean two ctypes descriptor classes that do not accept each other)."""
import importlib
import sys

_REAL = "audio-visual-sepatation-in-visual-agnostic-situtation_amd"
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + "."):
        sys.modules[__name__ + _name[len(_REAL):]] = _mod
sys.modules[__name__] = _pkg
