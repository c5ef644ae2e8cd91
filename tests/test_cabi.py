"""CPU: the C-ABI library builds/loads and exports every symbol include/yl_amd.h (product ABI) and include/yl_amd_diag.h
(diagnostics) declare."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _declared(header="yl_amd.h"):
    txt = (ROOT / "include" / header).read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(yl_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from yololite_amd import _lib

    L = _lib.lib()  # builds with hipcc if the .so is missing; no GPU needed to load it
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/yl_amd.h but not exported"
    assert set(names) == set(_lib.EXPORTS), "python binding list and header drifted apart"
    diag = _declared("yl_amd_diag.h")
    for n in diag:
        assert hasattr(L, n), f"{n} declared in include/yl_amd_diag.h but not exported"
    assert set(diag) == set(_lib.DIAG_EXPORTS), "diagnostic binding list and header drifted apart"
    assert not set(diag) & set(names), "a diagnostic entry point leaked into the product header"
    assert L.yl_abi_version() == 21


def test_struct_layouts_match_header():
    from yololite_amd import _lib

    assert ctypes.sizeof(_lib.View) == 32       # int64 + 6 x int32
    assert ctypes.sizeof(_lib.Level) == 40      # ABI v21: + offset_xywh
    # yl_op: 13 x int32/float (52) + pad to 8 (56) + 3 views (96) + 2 pointers (16)
    assert ctypes.sizeof(_lib.Op) == 56 + 96 + 16 + 8 + 16 + 32 + 8 + 16 + 8 + 16 + 40 + 24 + 8  # + post_*, in2, up, reserved, stem_*, fp8 scales, out8, extra filters, dfl_*


def test_struct_layouts_match_the_compiled_header(tmp_path):
    """The ctypes mirrors against the header ITSELF: a C program that includes include/yl_amd.h prints sizeof and every field's
    offsetof for yl_view / yl_level / yl_op; the ctypes structures must agree field by field (names differ where a C name is a
    Python keyword or was shortened: in -> inp, in2 -> inp2)."""
    import shutil
    import subprocess

    from yololite_amd import _lib

    if shutil.which("gcc") is None:
        import pytest

        pytest.skip("no C compiler")
    cname = {"inp": "in", "inp2": "in2"}
    structs = {"yl_view": _lib.View, "yl_level": _lib.Level, "yl_op": _lib.Op}
    lines = []
    for cn, cls in structs.items():
        lines.append(f'printf("{cn} %zu\\n", sizeof({cn}));')
        for f, _t in cls._fields_:
            lines.append(f'printf("{cn}.{f} %zu\\n", offsetof({cn}, {cname.get(f, f)}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "yl_amd.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cn, cls in structs.items():
        assert int(got[cn]) == ctypes.sizeof(cls), cn
        for f, _t in cls._fields_:
            assert int(got[f"{cn}.{f}"]) == getattr(cls, f).offset, f"{cn}.{f}"


def test_no_gpu_calls_fail_loudly():
    """Without a GPU the product path must raise, never fall back to a CPU implementation."""
    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from yololite_amd import YOLOLite
    from yololite_amd.utils import ops

    with pytest.raises(RuntimeError):
        YOLOLite("yolo11n.yaml").predict(torch.rand(1, 3, 64, 64), device="cpu", save=False, verbose=False)
    with pytest.raises(RuntimeError):
        ops.non_max_suppression(torch.rand(1, 84, 100), 0.25, 0.7)


def test_comm_group_fails_loudly_without_gpu():
    """yl_comm_*: the sharded path's exchanges have no CPU transport; without a HIP device communicator creation reports it."""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("GPU present")
    from yololite_amd import _lib

    L = _lib.lib()
    ident = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
    h = ctypes.c_void_p()
    rc = L.yl_comm_get_unique_id(ident)
    if rc == 0:
        rc = L.yl_comm_create(ident, 0, 1, ctypes.byref(h))
    assert rc < 0 and not h.value
    assert L.yl_last_error()  # a message, not a silent code
    assert L.yl_comm_rank(None) == -1 and L.yl_comm_world(None) == -1
    assert L.yl_comm_scatter(None, None, None, 16, 0, None) < 0
    assert L.yl_comm_gather(None, None, None, 16, 0, None) < 0
    assert L.yl_comm_barrier(None, None) < 0


def test_product_never_imports_oracle():
    pkg = ROOT / "yolo-lite_amd"
    for f in pkg.rglob("*.py"):
        src = f.read_text()
        assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle: product code must not use it"
