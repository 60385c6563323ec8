"""Import shim: makes the hyphen-named package directory importable as ``kvq_amd``.

The product package lives in ``kvq-challenge-cvpr-ntire2024_amd/`` (the name the
build contract fixes); a hyphen is not a legal Python identifier, so this module
loads that directory under the module name ``kvq_amd`` and replaces itself in
``sys.modules``.  ``import kvq_amd`` / ``from kvq_amd.models import model`` work
from the repo root (and from anywhere once the repo root is on ``sys.path``).
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "kvq-challenge-cvpr-ntire2024_amd")
_spec = importlib.util.spec_from_file_location(
    "kvq_amd", os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["kvq_amd"] = _mod
_spec.loader.exec_module(_mod)
