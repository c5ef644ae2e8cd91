"""ctypes binding of libyl_amd.so (the C ABI declared in include/yl_amd.h).

The product path has no CPU fallback: if the HIP library cannot be loaded, or no GPU is present
when a compute entry point is called, a ``RuntimeError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# The hot path replays as a hipGraph with exactly two parallel branches (network trunk + the 80x80 class
# branch).  With the runtime's default graph-queue heuristic the branch often ends up serialised behind the
# trunk (measured: 27.0 k vs 33.3 k images/s) -- two forced graph queues make the mapping deterministic.
# Must be in the environment before the HIP runtime initialises, hence at import time.
os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "2")
# Four batches in flight x two graph queues each = eight streams: give each its own hardware queue
# (the runtime's default of four makes them share; 37.4 k -> 38.6 k images/s).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["YL_AMD_LIB"]) if os.environ.get("YL_AMD_LIB") else _PKG / "libyl_amd.so"  # dev override

YL_BF16, YL_F32, YL_F16, YL_U8, YL_FP8 = 0, 1, 2, 3, 4
OP_CONV, OP_DWCONV3, OP_SPPF_POOL, OP_UPSAMPLE, OP_ATTENTION, OP_STEM = 1, 2, 3, 4, 5, 6

EXPORTS = (
    "yl_last_error", "yl_abi_version", "yl_device_count", "yl_engine_create", "yl_engine_destroy",
    "yl_engine_prepare_workspace", "yl_engine_forward", "yl_engine_forward_graph", "yl_engine_profile", "yl_engine_num_ops",
    "yl_engine_op_info", "yl_engine_detect_graph", "yl_detect_decode", "yl_nms_workspace_bytes", "yl_nms", "yl_decode_nms",
    "yl_preprocess_u8", "yl_letterbox_u8", "yl_nms_ml_workspace_bytes", "yl_nms_ml",
)


class View(C.Structure):
    _fields_ = [("offset", C.c_int64), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("pitch", C.c_int32), ("coff", C.c_int32), ("dtype", C.c_int32)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32), ("act", C.c_int32),
                ("accumulate", C.c_int32), ("heads", C.c_int32), ("key_dim", C.c_int32),
                ("head_dim", C.c_int32), ("scale", C.c_float), ("grp", C.c_int32),
                ("grp_stride", C.c_int32), ("grp_off", C.c_int32), ("cin_weight", C.c_int32),
                ("lane", C.c_int32),
                ("inp", View), ("out", View), ("res", View),
                ("weight", C.c_void_p), ("bias", C.c_void_p),
                ("post_cin", C.c_int32), ("post_act", C.c_int32),
                ("post_weight", C.c_void_p), ("post_bias", C.c_void_p),
                ("inp2", View), ("up", C.c_int32), ("reserved", C.c_int32),
                ("stem_weight", C.c_void_p), ("stem_bias", C.c_void_p),
                ("stem_act", C.c_int32), ("reserved2", C.c_int32),
                ("in_scale", C.c_float), ("out_scale", C.c_float), ("res_scale", C.c_float), ("out8_c0", C.c_int32),
                ("out8", View), ("out8_scale", C.c_float), ("out8_only", C.c_int32)]


class Level(C.Structure):
    _fields_ = [("offset", C.c_int64), ("offset_cls", C.c_int64), ("H", C.c_int32), ("W", C.c_int32),
                ("stride", C.c_float), ("nc8", C.c_int32)]


_lib = None


def build_library(verbose: bool = False) -> Path:
    """Compile csrc/*.hip for gfx950 into libyl_amd.so (hipcc cross-compiles without a GPU)."""
    import subprocess

    r = subprocess.run(["make", "-C", str(_PKG / "csrc"), "-j4"], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError("building libyl_amd.so failed:\n" + r.stderr[-4000:])
    return LIB_PATH


def lib() -> C.CDLL:
    """Load (once) and type the shared library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        if os.environ.get("YL_AMD_AUTOBUILD", "1") == "1" and (_PKG / "csrc" / "Makefile").exists():
            build_library()
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the HIP path)")
    # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import torch first so
    # that libyl_amd.so binds to the HIP runtime torch already initialised: two runtimes in one
    # process cannot share device pointers or streams ("no ROCm-capable device is detected").
    import torch  # noqa: F401

    L = C.CDLL(str(LIB_PATH))
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    L.yl_last_error.restype = C.c_char_p
    L.yl_last_error.argtypes = []
    L.yl_abi_version.restype = i32
    L.yl_device_count.restype = i32
    L.yl_engine_create.restype = i32
    L.yl_engine_create.argtypes = [C.POINTER(Op), i32, i32, i32, i32, C.POINTER(View), C.POINTER(Level), i32, i32,
                                   i64, C.POINTER(vp)]
    L.yl_engine_destroy.restype = None
    L.yl_engine_destroy.argtypes = [vp]
    L.yl_engine_prepare_workspace.restype = i32
    L.yl_engine_prepare_workspace.argtypes = [vp, vp, vp]
    for name in ("yl_engine_forward", "yl_engine_forward_graph"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp, vp, i32, vp, vp, vp]
    L.yl_engine_profile.restype = i32
    L.yl_engine_profile.argtypes = [vp, vp, i32, vp, vp, vp, i32, C.POINTER(f32)]
    L.yl_engine_num_ops.restype = i32
    L.yl_engine_num_ops.argtypes = [vp]
    L.yl_engine_op_info.restype = i32
    L.yl_engine_op_info.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_char_p)]
    L.yl_detect_decode.restype = i32
    L.yl_detect_decode.argtypes = [vp, C.POINTER(Level), i32, i32, i32, vp, vp]
    L.yl_nms_workspace_bytes.restype = i64
    L.yl_nms_workspace_bytes.argtypes = [i32, i32, i32]
    L.yl_nms.restype = i32
    L.yl_nms.argtypes = [vp, i32, i32, i32, f32, f32, i32, i32, vp, i32, i32, f32, vp, vp, vp, vp, vp]
    L.yl_nms_ml_workspace_bytes.restype = i64
    L.yl_nms_ml_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.yl_nms_ml.restype = i32
    L.yl_nms_ml.argtypes = L.yl_nms.argtypes
    L.yl_decode_nms.restype = i32
    L.yl_decode_nms.argtypes = [vp, C.POINTER(Level), i32, i32, i32, f32, f32, i32, i32, vp, i32, i32, f32, vp, vp, vp,
                                vp, vp]
    L.yl_engine_detect_graph.restype = i32
    L.yl_engine_detect_graph.argtypes = [vp, vp, i32, vp, f32, f32, i32, i32, vp, i32, i32, f32, vp, vp, vp, vp, vp]
    L.yl_preprocess_u8.restype = i32
    L.yl_preprocess_u8.argtypes = [vp, i32, i32, i32, vp, vp]
    L.yl_letterbox_u8.restype = i32
    L.yl_letterbox_u8.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    if L.yl_abi_version() != 11:
        raise RuntimeError("libyl_amd.so ABI version mismatch; rebuild it")
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    """Turn a negative status into a Python exception carrying yl_last_error()."""
    if rc == 0:
        return
    msg = lib().yl_last_error().decode(errors="replace")
    if rc == -4:  # threshold assertion of the reference (utils/ops.py:186-187)
        raise AssertionError(msg)
    if rc == -3:
        raise NotImplementedError(f"{what}: {msg}")
    raise RuntimeError(f"{what} failed ({rc}): {msg}")
