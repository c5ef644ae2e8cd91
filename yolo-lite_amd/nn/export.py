"""Safe-tensor export / import of a detection model ("an explicit safe-tensor export of
folded/packed weights").

A ``.pt`` written by the reference pickles the whole ``nn.Module`` (``engine/trainer.py:360-388``) and must be
unpickled to be read; a ``.safetensors`` file is a flat table of tensors plus a JSON header, so loading it runs no
code at all.  Two forms:

* ``folded=True`` (default): every Conv unit carries its BN-folded fp32 filter and bias
  (``<unit>.conv.weight``, ``<unit>.conv.bias``; the folding of ``torch_utils.py:182-209`` done once, here) --
  exactly the numbers the engine packs, so an engine built from the file is bit-identical to one built from the
  checkpoint (``tests/test_export.py``).
* ``folded=False``: the reference's own state-dict keys (conv + BN statistics), for tools that want them back.

Header metadata: ``format`` ("yololite_amd.detect/1"), ``folded``, ``yaml`` (architecture dict), ``names``,
``stride``, ``args`` -- what ``attempt_load_one_weight`` (``tasks.py:499-522``) reads off the pickled module.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, Union

import torch

from .lower import fold_conv_bn

FORMAT = "yololite_amd.detect/1"
_BN_SUFFIXES = (".bn.weight", ".bn.bias", ".bn.running_mean", ".bn.running_var", ".bn.num_batches_tracked")


def fold_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Reference-keyed state dict -> folded form: Conv units lose their BN tensors and gain ``conv.bias``."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k.endswith(_BN_SUFFIXES):
            continue
        unit = k[: -len(".conv.weight")] if k.endswith(".conv.weight") else None
        if unit is not None and unit + ".bn.weight" in sd:
            w, b = fold_conv_bn(sd, unit)
            out[unit + ".conv.weight"] = torch.from_numpy(w).clone()
            out[unit + ".conv.bias"] = torch.from_numpy(b).clone()
        else:
            out[k] = v.detach().to("cpu").contiguous().clone()
    return out


def is_folded(sd: Dict[str, torch.Tensor]) -> bool:
    return not any(k.endswith(".bn.weight") for k in sd)


def save_safetensors(model, path: Union[str, Path], folded: bool = True) -> str:
    """Write ``model`` (a ``DetectionModel``) to ``path``; returns the path."""
    from safetensors.torch import save_file

    sd = model.state_dict()
    if folded and not is_folded(sd):
        sd = fold_state_dict(sd)
    tensors = {k: (v.float() if v.is_floating_point() else v).contiguous() for k, v in sd.items()}
    meta = {
        "format": FORMAT,
        "folded": "1" if is_folded(tensors) else "0",
        "yaml": json.dumps(model.yaml),
        "names": json.dumps({str(k): v for k, v in model.names.items()}),
        "stride": json.dumps([float(s) for s in model.stride.tolist()]),
        "args": json.dumps({k: v for k, v in (model.args or {}).items() if isinstance(v, (str, int, float, bool, type(None)))}),
    }
    path = str(path)
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    save_file(tensors, path, metadata=meta)
    return path


def load_safetensors(path: Union[str, Path]):
    """``.safetensors`` written by ``save_safetensors`` -> ``(DetectionModel, header dict)``.  Fails loudly on a
    foreign file, a tensor-name mismatch or a shape mismatch."""
    from safetensors import safe_open

    from .tasks import DetectionModel

    path = str(path)
    if not Path(path).is_file():
        raise FileNotFoundError(f"'{path}' does not exist (no network: weights cannot be downloaded)")
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata() or {}
        if meta.get("format") != FORMAT:
            raise ValueError(f"'{path}' is not a {FORMAT} file (format = {meta.get('format')!r})")
        sd = {k: f.get_tensor(k) for k in f.keys()}
    model = DetectionModel(json.loads(meta["yaml"]), verbose=False)
    want = model.state_dict()
    if meta.get("folded") == "1":
        want = fold_state_dict(want)
    missing = [k for k in want if k not in sd]
    extra = [k for k in sd if k not in want]
    if missing or extra:
        raise RuntimeError(f"'{path}': tensor names do not match the architecture in its header: "
                           f"missing {missing[:3]}, unexpected {extra[:3]}")
    for k, v in sd.items():
        if tuple(v.shape) != tuple(want[k].shape):
            raise RuntimeError(f"'{path}': size mismatch for {k}: {tuple(v.shape)} vs {tuple(want[k].shape)}")
    model._sd = {k: sd[k] for k in want}  # header order = architecture order
    model._version += 1
    model.names = {int(k): str(v) for k, v in json.loads(meta.get("names", "{}")).items()} or model.names
    model.stride = torch.tensor(json.loads(meta.get("stride", "[8.0, 16.0, 32.0]")))
    model.pt_path = path
    return model.eval(), meta
