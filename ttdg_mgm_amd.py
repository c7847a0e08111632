"""Import shim: the package directory is ``ttdg-mgm_amd/`` (hyphenated, as the
project layout prescribes), which Python cannot import by name.  Importing
``ttdg_mgm_amd`` executes this file, which loads ``ttdg-mgm_amd/__init__.py``
as the package ``ttdg_mgm_amd`` and replaces itself in ``sys.modules``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ttdg-mgm_amd")
_spec = importlib.util.spec_from_file_location(
    "ttdg_mgm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ttdg_mgm_amd"] = _mod
_spec.loader.exec_module(_mod)
