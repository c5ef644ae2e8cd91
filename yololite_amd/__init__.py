"""Import shim: ``import yololite_amd`` resolves to the package directory ``yolo-lite_amd/``.

The product package lives in ``yolo-lite_amd/`` (a hyphen cannot appear in a Python import name), so
this module points its ``__path__`` there and executes that directory's ``__init__.py``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "yolo-lite_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__, "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), __file__, "exec"))
