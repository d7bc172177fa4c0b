"""Importable alias for the ``lcr-net_amd/`` package directory (a hyphen is not a valid Python identifier).

``import lcrnet_amd`` executes ``lcr-net_amd/__init__.py`` as the package ``lcrnet_amd``; sub-modules
(``lcrnet_amd.modules.ops`` …) resolve inside that directory.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lcr-net_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
