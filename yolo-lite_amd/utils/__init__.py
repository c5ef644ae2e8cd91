"""Small shared utilities of the predict path (logger, thread count, simple repr base class)."""
from __future__ import annotations

import logging
import os
import sys

# reference utils/__init__.py:32 -- intra-op threads the CPU path would use
NUM_THREADS = min(8, max(1, (os.cpu_count() or 1) - 1))
VERBOSE = str(os.getenv("YOLO_VERBOSE", True)).lower() == "true"


def _make_logger(name="yololite_amd"):
    lg = logging.getLogger(name)
    if not lg.handlers:
        h = logging.StreamHandler(sys.stdout)
        h.setFormatter(logging.Formatter("%(message)s"))
        lg.addHandler(h)
        lg.propagate = False
    lg.setLevel(logging.INFO if VERBOSE else logging.ERROR)
    return lg


LOGGER = _make_logger()


def _public_state(obj):
    """(name, value) of every public, non-callable attribute of ``obj`` (properties included), sorted by name."""
    for name in dir(obj):
        if name[:1] == "_":
            continue
        value = getattr(obj, name)
        if not callable(value):
            yield name, value


class SimpleClass:
    """Base of the result / metric containers: printing one lists its public state, and a misspelt attribute names the
    class and shows its documentation instead of a bare ``AttributeError`` (reference utils/__init__.py:236-289)."""

    def __repr__(self):
        kind = f"{type(self).__module__}.{type(self).__name__}"
        rows = [f"{n}: {type(v).__module__}.{type(v).__name__} object" if isinstance(v, SimpleClass) else f"{n}: {v!r}"
                for n, v in _public_state(self)]
        return "\n".join([f"{kind} object with attributes:", "", *rows])

    __str__ = __repr__

    def __getattr__(self, attr):  # only reached when normal lookup failed
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{attr}'. See valid attributes below.\n{self.__doc__}")
