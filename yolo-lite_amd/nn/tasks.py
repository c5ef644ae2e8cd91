"""Model object and ``.pt`` loader (reference ``yololite/nn/tasks.py:79-276, 406-522``).

``DetectionModel`` here is a plain container (architecture dict + fp32 state dict + names +
strides): the network itself runs in the HIP engine, not as ``nn.Module`` layers.

Checkpoints written by the reference (and by upstream Ultralytics) pickle a whole ``nn.Module``
(``trainer.py:360-388``) and the reference loads them with ``weights_only=False``
(``utils/patches.py:63-85``), i.e. arbitrary code execution.  ``torch_safe_load`` below uses a
restricted unpickler instead: tensors, storages and containers are allow-listed; every other global
(``yololite.nn.*``, ``ultralytics.nn.*``, ``torch.nn.modules.*`` ...) is replaced by an inert shell
that only records its ``__dict__``; the state dict, ``yaml``, ``names`` and ``args`` are then read
from the shells.
"""
from __future__ import annotations

import collections
import pickle
from pathlib import Path
from typing import Dict, Optional, Tuple, Union

import torch

from ..cfg import DEFAULT_CFG_DICT, DEFAULT_CFG_KEYS
from ..utils import LOGGER
from .arch import count_parameters, parse_model, yaml_model_load, yolo11_dict  # noqa: F401
from .synth import synth_state_dict


class DetectionModel:
    """YOLO11 detection model description (reference ``tasks.py:231-276``).

    ``DetectionModel("yolo11n.yaml")`` builds the architecture with deterministic random weights
    (the reference leaves PyTorch's default init; no pretrained weights ship offline).
    """

    def __init__(self, cfg: Union[str, Path, dict] = "yolo11n.yaml", ch: int = 3, nc: Optional[int] = None,
                 verbose: bool = True, state_dict: Optional[Dict[str, torch.Tensor]] = None, seed: int = 0):
        self.yaml = cfg if isinstance(cfg, dict) else yaml_model_load(cfg)
        if nc and nc != self.yaml["nc"]:
            self.yaml["nc"] = nc
        self.yaml.setdefault("ch", ch)
        self.specs, self.save = parse_model(self.yaml, ch=self.yaml["ch"])
        self.nc = self.specs[-1].nc
        self.names = {i: f"{i}" for i in range(self.nc)}  # tasks.py:251
        self.stride = torch.tensor([8.0, 16.0, 32.0])     # tasks.py:256-268 (P3-P5 of yolo11.yaml)
        self.inplace = True
        self.end2end = False
        self.task = "detect"
        self.args: dict = {}
        self.pt_path = None
        self._sd = state_dict if state_dict is not None else synth_state_dict(
            self.yaml.get("scale") or "n", seed, self.nc, model_dict=self.yaml)
        self._version = 0  # bumped when weights change so backends rebuild their engines
        self.training = False
        self.fp16 = False

    # --- nn.Module-like surface used by callers of the reference
    def state_dict(self) -> Dict[str, torch.Tensor]:
        return self._sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        if strict:
            missing = [k for k in self._sd if k not in sd]
            unexpected = [k for k in sd if k not in self._sd]
            if missing or unexpected:
                raise RuntimeError(f"state_dict mismatch: missing {missing[:5]}, unexpected {unexpected[:5]}")
        for k, v in sd.items():
            if k in self._sd:
                if tuple(v.shape) != tuple(self._sd[k].shape):
                    raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(self._sd[k].shape)}")
                self._sd[k] = v.detach().to("cpu", torch.float32 if v.is_floating_point() else v.dtype)
        self._version += 1

    def parameters(self):
        return (v for k, v in self._sd.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))

    def eval(self):
        self.training = False
        return self

    def float(self):
        self.fp16 = False
        return self

    def half(self):
        self.fp16 = True
        return self

    def to(self, *args, **kwargs):
        return self

    def fuse(self, verbose=True):
        """Conv+BN folding happens when the engine is built (``lower.fold_conv_bn``); kept so that
        ``AutoBackend(fuse=True)`` works (the reference crashes here: autobackend.py:74)."""
        return self

    def info(self):
        n = count_parameters(self.specs)
        LOGGER.info(f"YOLO11{self.yaml.get('scale', '')} summary: {len(self.specs)} layers, {n} parameters")
        return n


# ----------------------------------------------------------------------------- .pt loading
class _Shell:
    """Inert stand-in for any non-allow-listed pickled class: keeps state, runs nothing."""

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[0], dict):
            self.__dict__.update(state[0])

    def __call__(self, *args, **kwargs):
        return _Shell()


_SHELLS: Dict[Tuple[str, str], type] = {}
_ALLOWED = {
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("builtins", "set"): set, ("builtins", "frozenset"): frozenset, ("builtins", "slice"): slice,
    ("builtins", "complex"): complex, ("builtins", "dict"): dict, ("builtins", "list"): list,
    ("builtins", "tuple"): tuple, ("builtins", "int"): int, ("builtins", "float"): float,
    ("builtins", "str"): str, ("builtins", "bool"): bool,
}


def _allowed_global(module: str, name: str):
    if module == "__builtin__":  # protocol-2 pickles (torch.save default) use the Python 2 module name
        module = "builtins"
    if (module, name) in _ALLOWED:
        return _ALLOWED[(module, name)]
    if module == "torch._utils" and name in {"_rebuild_tensor_v2", "_rebuild_parameter", "_rebuild_tensor",
                                              "_rebuild_parameter_with_state"}:
        return getattr(torch._utils, name)
    if module == "torch" and (name.endswith("Storage") or name in {"Size", "device", "dtype"} or
                              name in {"float32", "float16", "bfloat16", "int64", "int32", "uint8", "bool"}):
        return getattr(torch, name)
    if module == "torch.serialization" and name == "_get_layout":
        return torch.serialization._get_layout
    if module in ("numpy.core.multiarray", "numpy._core.multiarray") and name in {"scalar", "_reconstruct"}:
        import numpy as np

        return getattr(np.core.multiarray if hasattr(np, "core") else np._core.multiarray, name)
    if module == "numpy" and name in {"dtype", "ndarray"}:
        import numpy as np

        return getattr(np, name)
    return None


class RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        g = _allowed_global(module, name)
        if g is not None:
            return g
        key = (module, name)
        if key not in _SHELLS:
            _SHELLS[key] = type(name, (_Shell,), {"_yl_module": module, "_yl_name": name})
        return _SHELLS[key]


class _RestrictedPickle:
    """``pickle_module`` for ``torch.load``: only ``Unpickler`` differs from the stdlib."""
    __name__ = "yl_restricted_pickle"
    Unpickler = RestrictedUnpickler
    load = staticmethod(lambda f, **kw: RestrictedUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    PickleError = pickle.PickleError
    UnpicklingError = pickle.UnpicklingError
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL


def _module_state_dict(m, prefix="", out=None) -> Dict[str, torch.Tensor]:
    """state_dict() of a shell tree: parameters, persistent buffers, then children, like nn.Module."""
    out = collections.OrderedDict() if out is None else out
    d = getattr(m, "__dict__", {})
    non_persistent = d.get("_non_persistent_buffers_set", set()) or set()
    for k, v in (d.get("_parameters") or {}).items():
        if v is not None:
            out[prefix + k] = v.detach() if isinstance(v, torch.Tensor) else getattr(v, "data", v)
    for k, v in (d.get("_buffers") or {}).items():
        if v is not None and k not in non_persistent:
            out[prefix + k] = v
    for k, child in (d.get("_modules") or {}).items():
        if child is not None:
            _module_state_dict(child, prefix + k + ".", out)
    return out


def torch_safe_load(weight: Union[str, Path]):
    """``.pt`` file -> (ckpt dict of plain data and shells, path) without executing pickled code."""
    file = str(weight)
    if not Path(file).is_file():
        raise FileNotFoundError(f"'{file}' does not exist (no network: weights cannot be downloaded)")
    ckpt = torch.load(file, map_location="cpu", pickle_module=_RestrictedPickle, weights_only=False)
    if not isinstance(ckpt, dict):
        ckpt = {"model": getattr(ckpt, "model", ckpt)}
    return ckpt, file


def attempt_load_one_weight(weight, device=None, inplace=True, fuse=False):
    """Reference ``tasks.py:499-522``: pick ``ema`` else ``model``, fp32, attach args / pt_path / task.
    ``*.safetensors`` files written by ``nn/export.py`` load without any unpickling."""
    if Path(str(weight)).suffix == ".safetensors":
        import json

        from .export import load_safetensors

        model, meta = load_safetensors(weight)
        args = {**DEFAULT_CFG_DICT, **json.loads(meta.get("args", "{}"))}
        model.args = {k: v for k, v in args.items() if k in DEFAULT_CFG_KEYS}
        model.task = "detect"
        return model, {"model": None, "train_args": model.args, "safetensors": meta}
    ckpt, weight = torch_safe_load(weight)
    args = {**DEFAULT_CFG_DICT, **{k: v for k, v in (ckpt.get("train_args") or {}).items() if not isinstance(v, _Shell)}}
    shell = ckpt.get("ema") or ckpt["model"]
    if isinstance(shell, (dict, collections.OrderedDict)) and not hasattr(shell, "_modules"):
        sd, yaml_d, names = shell, ckpt.get("yaml"), ckpt.get("names")  # plain state-dict checkpoint
    else:
        sd = _module_state_dict(shell)
        yaml_d = getattr(shell, "yaml", None)
        names = getattr(shell, "names", None)
    if not isinstance(yaml_d, dict):
        raise ValueError(f"'{weight}' carries no model yaml; cannot rebuild the architecture")
    sd = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items() if isinstance(v, torch.Tensor)}
    model = DetectionModel(dict(yaml_d), verbose=False, state_dict=None)
    ref_keys = set(model.state_dict().keys())
    extra = [k for k in sd if k not in ref_keys]
    missing = [k for k in ref_keys if k not in sd]
    if missing:
        raise RuntimeError(f"checkpoint '{weight}' lacks {len(missing)} tensors, e.g. {missing[:3]}")
    model.load_state_dict({k: v for k, v in sd.items() if k in ref_keys})
    if extra:
        LOGGER.warning(f"ignored {len(extra)} unexpected checkpoint tensors, e.g. {extra[:3]}")
    if isinstance(names, (list, tuple)):
        names = dict(enumerate(names))
    if isinstance(names, dict) and names:
        model.names = {int(k): str(v) for k, v in names.items()}
    st = getattr(shell, "stride", None)
    if isinstance(st, torch.Tensor) and st.numel() == 3:
        model.stride = st.float()
    model.args = {k: v for k, v in args.items() if k in DEFAULT_CFG_KEYS}
    model.pt_path = weight
    model.task = "detect"
    return model.eval(), ckpt


def attempt_load_weights(weights, device=None, inplace=True, fuse=False):
    """Single-model form of reference ``tasks.py:461-496`` (ensembles are out of scope)."""
    if isinstance(weights, (list, tuple)):
        if len(weights) != 1:
            raise NotImplementedError("model ensembles are outside the accelerated predict path")
        weights = weights[0]
    return attempt_load_one_weight(weights, device, inplace, fuse)[0]


def guess_model_task(model) -> str:
    return "detect"
