"""Import alias for the hyphenated package directory.

The product package lives in
``multi-modal-retrieval-system-image-search-and-data-governance_amd/`` (the name the
project contract fixes).  Hyphens are not legal in a Python module name, so this
shim loads that directory under the importable name ``mmr_amd``; submodules
(``mmr_amd.search``, ``mmr_amd.clip`` ...) resolve inside the hyphenated directory.
"""
import importlib.util
import os
import sys

PACKAGE_DIRNAME = "multi-modal-retrieval-system-image-search-and-data-governance_amd"
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_pkg_dir = os.path.join(_root, PACKAGE_DIRNAME)

_spec = importlib.util.spec_from_file_location(
    "mmr_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mmr_amd"] = _mod
_spec.loader.exec_module(_mod)
