"""Import shim: ``import emdenoise`` loads the package that lives in the (non-importable,
hyphenated) directory ``ai-cv-automation-elect-micr_amd/`` next to this file."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "ai-cv-automation-elect-micr_amd")
_spec = _ilu.spec_from_file_location("emdenoise", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["emdenoise"] = _mod
_spec.loader.exec_module(_mod)
