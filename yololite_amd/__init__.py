"""Import alias: ``import yololite_amd`` loads the package that lives in ``yolo-lite_amd/``.

The product package directory is named ``yolo-lite_amd`` (a hyphen cannot appear in a Python import name), so this
module asks the standard import machinery to load that directory AS the package ``yololite_amd``
(``importlib.util.spec_from_file_location`` with ``submodule_search_locations``) and puts the result in ``sys.modules``
in its own place: submodules, relative imports and ``__path__`` then work exactly as for any package.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "yolo-lite_amd")
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_real, "__init__.py"), submodule_search_locations=[_real])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
