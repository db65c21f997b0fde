"""Import alias: ``import avsep_amd`` == the package directory
``audio-visual-sepatation-in-visual-agnostic-situtation_amd`` (whose name is not a Python identifier)."""
import importlib
import sys

_pkg = importlib.import_module("audio-visual-sepatation-in-visual-agnostic-situtation_amd")
sys.modules[__name__] = _pkg
