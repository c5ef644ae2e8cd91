"""Device selection and inference-mode decorator (reference ``yololite/utils/torch_utils.py``)."""
from __future__ import annotations

import os

import torch

from . import LOGGER, NUM_THREADS


def smart_inference_mode():
    """Decorator applying ``torch.inference_mode`` (reference torch_utils.py:37-47)."""

    def decorate(fn):
        if torch.is_inference_mode_enabled():
            return fn
        return torch.inference_mode()(fn)

    return decorate


def select_device(device="", batch=0, newline=False, verbose=True) -> torch.device:
    """'' | 'cpu' | 'cuda' | '0' | 'cuda:0' | '0,1' -> torch.device (reference torch_utils.py:92-172).

    Like the reference, a multi-GPU string is validated but a single device ``cuda:0`` (of the
    visible set) is returned; data-parallel sharding lives in ``yololite_amd.dist``.  ``device='cpu'``
    is accepted for host-side plumbing, but the HIP network itself refuses to run there.
    """
    if isinstance(device, torch.device):
        return device
    s = str(device).lower()
    for r in ("cuda:", "none", "(", ")", "[", "]", "'", " "):
        s = s.replace(r, "")
    if s == "cpu":
        torch.set_num_threads(NUM_THREADS)
        return torch.device("cpu")
    if s in ("mps", "mps:0"):
        raise ValueError("Invalid device 'mps': this build targets AMD MI355X (HIP) only")
    if s:
        if s == "cuda":
            s = "0"
        ids = [x for x in s.split(",") if x]
        if not all(x.isdigit() for x in ids):
            raise ValueError(f"Invalid CUDA 'device={device}' requested. Use 'device=cpu' or pass valid device(s).")
        if not torch.cuda.is_available() or torch.cuda.device_count() <= max(int(x) for x in ids):
            raise ValueError(f"Invalid CUDA 'device={device}' requested. Use 'device=cpu' or pass valid CUDA "
                             f"device(s) if available, i.e. 'device=0' or 'device=0,1,2,3' for Multi-GPU.\n"
                             f"\ntorch.cuda.is_available(): {torch.cuda.is_available()}"
                             f"\ntorch.cuda.device_count(): {torch.cuda.device_count()}"
                             f"\nos.environ['CUDA_VISIBLE_DEVICES']: {os.environ.get('CUDA_VISIBLE_DEVICES')}\n")
        if len(ids) > 1 and batch > 0 and batch % len(ids) != 0:
            raise ValueError(f"'batch={batch}' must be a multiple of GPU count {len(ids)}.")
        dev = torch.device(f"cuda:{ids[0]}")
    elif torch.cuda.is_available():
        dev = torch.device("cuda:0")
    else:
        torch.set_num_threads(NUM_THREADS)
        dev = torch.device("cpu")
    if verbose:
        LOGGER.info(f"yololite_amd torch-{torch.__version__} device {dev}"
                    + (f" ({torch.cuda.get_device_name(dev)})" if dev.type == "cuda" else ""))
    return dev
