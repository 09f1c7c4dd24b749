"""Importable alias for the package whose directory name is ``sahs-deformable-nerf_amd``."""
import importlib
import sys

_pkg = importlib.import_module("sahs-deformable-nerf_amd")
sys.modules[__name__] = _pkg
