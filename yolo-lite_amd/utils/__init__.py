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


class SimpleClass:
    """Readable ``str``/``repr`` for result containers (reference utils/__init__.py:236-289)."""

    def __str__(self):
        attr = []
        for a in dir(self):
            v = getattr(self, a)
            if not callable(v) and not a.startswith("_"):
                if isinstance(v, SimpleClass):
                    s = f"{a}: {v.__module__}.{v.__class__.__name__} object"
                else:
                    s = f"{a}: {repr(v)}"
                attr.append(s)
        return f"{self.__module__}.{self.__class__.__name__} object with attributes:\n\n" + "\n".join(attr)

    def __repr__(self):
        return self.__str__()

    def __getattr__(self, attr):
        name = self.__class__.__name__
        raise AttributeError(f"'{name}' object has no attribute '{attr}'. See valid attributes below.\n{self.__doc__}")
