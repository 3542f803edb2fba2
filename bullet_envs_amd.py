"""Importable alias of the package directory `bullet-envs_amd/` (hyphen in the name)."""
import importlib as _importlib
import sys as _sys

_pkg = _importlib.import_module("bullet-envs_amd")
_sys.modules[__name__] = _pkg
