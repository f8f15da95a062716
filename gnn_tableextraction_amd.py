"""Import shim: ``import gnn_tableextraction_amd`` loads the package that lives in the directory
``gnn-tableextraction_amd/`` (a hyphen is not importable)."""
import importlib.util
import os
import sys

_real = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gnn-tableextraction_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_real, "__init__.py"),
                                               submodule_search_locations=[_real])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
