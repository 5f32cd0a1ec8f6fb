"""Import shim: ``import legion1_amd`` loads the package that lives in ``legion-1_amd/``
(the directory name the project layout prescribes is not a valid Python identifier)."""
import importlib.util as _u
import os as _os
import sys as _sys

_root = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "legion-1_amd")
_spec = _u.spec_from_file_location(__name__, _os.path.join(_root, "__init__.py"), submodule_search_locations=[_root])
_mod = _u.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
