"""ctypes binding of libyl_amd.so (the C ABI declared in include/yl_amd.h).

The product path has no CPU fallback: if the HIP library cannot be loaded, or no GPU is present
when a compute entry point is called, a ``RuntimeError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# Runtime knobs of the HIP runtime that the hot path's speed depends on.  They are NOT set at import: call
# ``apply_runtime_knobs()`` (bench.py, tests/conftest.py, __graft_entry__ and AutoBackend do) before the first GPU
# call of the process; ``runtime_knob_status()`` says whether that happened in time.
#   GPU_MAX_HW_QUEUES=16  several batches in flight on several streams: give each stream its own hardware queue
#       (the runtime's default of four makes them share: 37.4 k -> 38.6 k images/s with 8; round 2, same box, alternating
#       runs: 8 queues / 8 streams 43.6-43.9 k, 16 / 16 46.5-47.1 k, 20 / 20 46.9-47.1 k, 24 / 24 43.6 k, 32 / 32 32.7 k
#       images/s -- past ~20 the queues are time-sliced; and the stream count should EQUAL the queue count: 16 queues /
#       8 streams 34.5 k, 20 / 16 37.6 k.  bench.py therefore sets 20 for its 20 batches in flight, 8 for a rank of the
#       sharded path (7 batch streams beside RCCL's); a caller that pipelines N batches itself should export N.)
# NOT set any more (round 2): DEBUG_HIP_FORCE_GRAPH_QUEUES=2.  Round 1 forced two graph queues so that the captured
# graph's class-branch lane would overlap the trunk (27.0 k -> 33.3 k images/s).  tests/test_gpu_dist.py showed that
# with that debug flag a replay does NOT honour the fork edge of the graph: the branch starts on the second queue
# before the trunk node it depends on has finished and reads the previous replay's tensor -- invisible while every
# replay sees the same input (all of round 1's tests and its bench), wrong as soon as the input changes between replays.
# The hot path's graph is therefore captured as ONE chain (nn/lower.py: side lanes are opt-in), and the overlap comes
# from several batches in flight on several streams only.  ``graph_branch_probe()`` still reports what the runtime does
# with a two-branch graph.
RUNTIME_KNOBS = {"GPU_MAX_HW_QUEUES": "16"}
_knob_status = {"applied": False, "before_hip_init": None, "values": {}}


def apply_runtime_knobs() -> dict:
    """Put RUNTIME_KNOBS into the environment (values already present win) and record whether the HIP runtime of this
    process can still see them: it reads its flags once, when it initialises.  Returns ``runtime_knob_status()``."""
    import sys

    early = True
    t = sys.modules.get("torch")
    if t is not None and t.cuda.is_initialized():
        early = False
    if _lib is not None and not _knob_status["applied"]:
        early = False  # libyl_amd.so was already driven without them
    for k, v in RUNTIME_KNOBS.items():
        os.environ.setdefault(k, v)
    if not _knob_status["applied"]:
        _knob_status.update(applied=True, before_hip_init=early)
    _knob_status["values"] = {k: os.environ.get(k) for k in RUNTIME_KNOBS}
    return runtime_knob_status()


def runtime_knob_status() -> dict:
    return {"applied": _knob_status["applied"], "before_hip_init": _knob_status["before_hip_init"],
            **{k: os.environ.get(k) for k in RUNTIME_KNOBS}}


def graph_branch_probe(device=None, spin_us: int = 300) -> dict:
    """Device-side check that two parallel branches of a captured graph overlap (yl_graph_branch_probe)."""
    import torch

    dev = torch.device(device if device is not None else "cuda:0")
    ms, conc = C.c_float(), C.c_int()
    with torch.cuda.device(dev):
        check(lib().yl_graph_branch_probe(torch.cuda.current_stream(dev).cuda_stream, int(spin_us), C.byref(ms), C.byref(conc)),
              "yl_graph_branch_probe")
    return {"spin_us": spin_us, "graph_ms": round(ms.value, 4), "concurrent": bool(conc.value)}


NUM_XCD, CUS_PER_XCD = 8, 32  # MI355X


def cu_slice_stream(device, part: int, nparts: int):
    """A ``torch.cuda.ExternalStream`` whose dispatches are confined to 1 / ``nparts`` of the compute units of EVERY XCD
    (``yl_stream_create_masked``).  Measured layout of the 256-bit mask on MI355X (tools/xcd_probe.py): bit
    32 c + u selects CU c (0..7) of one (shader engine, XCD) pair u; a queue cannot be confined to one XCD -- the
    dispatcher spreads workgroups over all eight regardless, and a mask that leaves an XCD empty is ignored.  The
    stream is never destroyed (a handful per process)."""
    import torch

    dev = torch.device(device)
    words = (C.c_uint32 * (NUM_XCD * CUS_PER_XCD // 32))()
    for c in range(8):
        if c % nparts == part % nparts:
            words[c] = 0xFFFFFFFF
    out = C.c_void_p()
    with torch.cuda.device(dev):
        check(lib().yl_stream_create_masked(words, len(words), C.byref(out)), "yl_stream_create_masked")
    return torch.cuda.ExternalStream(out.value, device=dev)


def xcc_census(stream, n_workgroups: int = 4096):
    """Workgroups of a probe launch on ``stream`` per (XCD, shader engine, CU): an (8, 4, 16) int tensor."""
    import torch

    dev = stream.device
    hist = torch.zeros(512, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib().yl_xcc_census(hist.data_ptr(), int(n_workgroups), stream.cuda_stream), "yl_xcc_census")
    stream.synchronize()
    return hist.view(8, 4, 16).cpu()


_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["YL_AMD_LIB"]) if os.environ.get("YL_AMD_LIB") else _PKG / "libyl_amd.so"  # dev override

YL_BF16, YL_F32, YL_F16, YL_U8, YL_FP8 = 0, 1, 2, 3, 4
OP_CONV, OP_DWCONV3, OP_SPPF_POOL, OP_UPSAMPLE, OP_ATTENTION, OP_STEM = 1, 2, 3, 4, 5, 6

EXPORTS = (  # include/yl_amd.h: the product ABI
    "yl_last_error", "yl_abi_version", "yl_device_count", "yl_engine_create", "yl_engine_destroy",
    "yl_engine_prepare_workspace", "yl_engine_forward", "yl_engine_forward_graph", "yl_engine_profile", "yl_engine_num_ops",
    "yl_engine_op_info", "yl_engine_detect_graph", "yl_detect_decode", "yl_nms_workspace_bytes", "yl_nms", "yl_decode_nms",
    "yl_preprocess_u8", "yl_letterbox_u8", "yl_nms_ml_workspace_bytes", "yl_nms_ml", "yl_engine_profile_detect",
    "yl_engine_prefilter_fused", "yl_engine_detect_graph_prepare", "yl_graph_launch_many", "yl_engine_detect_graph_rebind",
    "yl_comm_get_unique_id", "yl_comm_create", "yl_comm_destroy", "yl_comm_rank", "yl_comm_world", "yl_comm_scatter",
    "yl_comm_gather", "yl_comm_barrier",
)
COMM_ID_BYTES = 128  # YL_COMM_ID_BYTES
DIAG_EXPORTS = (  # include/yl_amd_diag.h: probes for tools/ and the runtime self-checks, not part of the product ABI
    "yl_graph_branch_probe", "yl_stream_create_masked", "yl_stream_destroy", "yl_xcc_census", "yl_diag_set_stamps",
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
                ("inp2", View), ("up", C.c_int32), ("group", C.c_int32),
                ("stem_weight", C.c_void_p), ("stem_bias", C.c_void_p),
                ("stem_act", C.c_int32), ("reserved2", C.c_int32),
                ("in_scale", C.c_float), ("out_scale", C.c_float), ("res_scale", C.c_float), ("out8_c0", C.c_int32),
                ("out8", View), ("out8_scale", C.c_float), ("out8_only", C.c_int32),
                ("n_extra", C.c_int32), ("reserved3", C.c_int32), ("extra_weight", C.c_void_p), ("extra_bias", C.c_void_p),
                ("dfl_level", C.c_int32), ("dfl_keep", C.c_int32)]


class Level(C.Structure):
    _fields_ = [("offset", C.c_int64), ("offset_cls", C.c_int64), ("H", C.c_int32), ("W", C.c_int32),
                ("stride", C.c_float), ("nc8", C.c_int32), ("offset_xywh", C.c_int64)]


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
    L.yl_engine_detect_graph_rebind.restype = i32
    L.yl_engine_detect_graph_rebind.argtypes = L.yl_engine_detect_graph.argtypes
    L.yl_engine_detect_graph_prepare.restype = i32
    L.yl_engine_detect_graph_prepare.argtypes = L.yl_engine_detect_graph.argtypes + [C.POINTER(vp)]
    L.yl_graph_launch_many.restype = i32
    L.yl_graph_launch_many.argtypes = [C.POINTER(vp), C.POINTER(vp), i32]
    L.yl_preprocess_u8.restype = i32
    L.yl_preprocess_u8.argtypes = [vp, i32, i32, i32, vp, vp]
    L.yl_letterbox_u8.restype = i32
    L.yl_letterbox_u8.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    L.yl_engine_profile_detect.restype = i32
    L.yl_engine_profile_detect.argtypes = [vp, vp, i32, vp, f32, f32, i32, i32, vp, vp, vp, vp, i32, C.POINTER(f32),
                                           C.POINTER(f32)]
    L.yl_graph_branch_probe.restype = i32
    L.yl_graph_branch_probe.argtypes = [vp, i32, C.POINTER(f32), C.POINTER(C.c_int)]
    L.yl_stream_create_masked.restype = i32
    L.yl_stream_create_masked.argtypes = [C.POINTER(C.c_uint32), i32, C.POINTER(vp)]
    L.yl_stream_destroy.restype = i32
    L.yl_stream_destroy.argtypes = [vp]
    L.yl_xcc_census.restype = i32
    L.yl_xcc_census.argtypes = [vp, i32, vp]
    L.yl_diag_set_stamps.restype = i32
    L.yl_diag_set_stamps.argtypes = [vp]
    L.yl_engine_prefilter_fused.restype = i32
    L.yl_engine_prefilter_fused.argtypes = [vp, f32]
    L.yl_comm_get_unique_id.restype = i32
    L.yl_comm_get_unique_id.argtypes = [vp]
    L.yl_comm_create.restype = i32
    L.yl_comm_create.argtypes = [vp, i32, i32, C.POINTER(vp)]
    L.yl_comm_destroy.restype = None
    L.yl_comm_destroy.argtypes = [vp]
    for name in ("yl_comm_rank", "yl_comm_world"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp]
    for name in ("yl_comm_scatter", "yl_comm_gather"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp, vp, vp, i64, i32, vp]
    L.yl_comm_barrier.restype = i32
    L.yl_comm_barrier.argtypes = [vp, vp]
    if L.yl_abi_version() != 21:
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
