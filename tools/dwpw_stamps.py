"""Phase breakdown of the Detect class branch's last launch (dwpw_group_kernel: depthwise + pointwise + last 1x1 of the three levels)
from dwpw_body's in-kernel stamps.  Needs a -DYL_STAMPS build (rm csrc/conv_mfma.o csrc/engine.o; make -C yolo-lite_amd/csrc EXTRA=-DYL_STAMPS)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from yololite_amd import _lib  # noqa: E402

_lib.apply_runtime_knobs()
from yololite_amd.nn import arch  # noqa: E402
from yololite_amd.nn.engine import HipEngine  # noqa: E402
from yololite_amd.nn.synth import synth_state_dict  # noqa: E402

NAMES = ["start", "filter slabs requested", "barrier", "depthwise phase done", "barrier", "pointwise GEMM done", "SiLU + tile written", "barrier",
         "last 1x1 + pre-filter done", "stored"]


def main(B=32, H=640, W=640):
    dev = torch.device("cuda:0")
    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    eng = HipEngine.from_model(specs, synth_state_dict("n", 0), B, H, W, dev)
    x = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device=dev)
    nwg = 4096
    buf = torch.zeros(nwg * 4 * 32 * 2, dtype=torch.int64, device=dev)
    L = _lib.lib()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eng.detect(x, 0.25, 0.7, graph=False)
        torch.cuda.synchronize()
        if not L.yl_diag_set_stamps(C.c_void_p(buf.data_ptr())):
            print("this libyl_amd.so is the production build: rebuild with EXTRA=-DYL_STAMPS")
            return
        eng.detect(x, 0.25, 0.7, graph=False)
        torch.cuda.synchronize()
    L.yl_diag_set_stamps(None)
    s = buf.view(nwg, 4, 32, 2).cpu().numpy().astype(np.int64)
    ran = s[:, 0, 9, 1] > 0
    s = s[ran]
    n0 = (H // 8) * (W // 8) * B // 128  # level 0's workgroups come first
    t0 = s[:, :, 0, 1].min()
    print(f"{len(s)} workgroups stamped; launch span {(s[:, :, 9, 1].max() - t0) * 10 / 1e3:.1f} us; level 0 = the first {n0}")
    for label, sel in (("level 0", slice(0, n0)), ("levels 1-2", slice(n0, None))):
        q = s[sel]
        ns = (q[..., 1] - q[:, :, :1, 1].min(axis=1, keepdims=True)) * 10
        prev = 0.0
        print(label)
        for i, name in enumerate(NAMES):
            t = ns[:, :, i] / 1e3
            print(f"  {i} {name:28s} median {np.median(t):6.2f} us (+{np.median(t) - prev:5.2f})  p90 {np.percentile(t, 90):6.2f}")
            prev = float(np.median(t))
        life = (q[:, :, 9, 1].max(axis=1) - q[:, :, 0, 1].min(axis=1)) * 10 / 1e3
        print(f"  workgroup lifetime median {np.median(life):.2f} us")
    life = (s[:, :, 9, 1].max(axis=1) - s[:, :, 0, 1].min(axis=1)) * 10 / 1e3
    print(f"mean concurrency {life.sum() / ((s[:, :, 9, 1].max() - t0) * 10 / 1e3):.0f} workgroups")


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:]])
