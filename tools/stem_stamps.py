"""Phase breakdown of stem2_kernel (csrc/stem2.hip) from its in-kernel stamps, inside the network's first launch.

Needs a library built with -DYL_STAMPS (rm csrc/stem2.o csrc/engine.o; make -C yolo-lite_amd/csrc EXTRA=-DYL_STAMPS).
"""
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

NAMES = ["start", "window loaded + converted", "barrier", "stem tiles done", "barrier", "filters in registers", "layer 1 done"]


def main(B=32, H=640, W=640):
    dev = torch.device("cuda:0")
    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    eng = HipEngine.from_model(specs, synth_state_dict("n", 0), B, H, W, dev)
    x = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device=dev)
    nwg = B * ((H // 4 + 4) // 5) * ((W // 4 + 31) // 32)
    buf = torch.zeros(nwg * 4 * 32 * 2, dtype=torch.int64, device=dev)
    L = _lib.lib()
    if not L.yl_diag_set_stamps(C.c_void_p(buf.data_ptr())):
        print("this libyl_amd.so is the production build: rebuild with EXTRA=-DYL_STAMPS")
        return
    for _ in range(3):
        eng.forward(x)
        torch.cuda.synchronize()
    L.yl_diag_set_stamps(None)
    st = buf.view(nwg, 4, 32, 2).cpu().numpy().astype(np.int64)
    t0 = st[:, :, 0, 1].min()
    print(f"{nwg} workgroups; kernel span {(st[:, :, 6, 1].max() - t0) * 10 / 1e3:.1f} us; starts spread over {(st[:, :, 0, 1].max() - t0) * 10 / 1e3:.1f} us")
    ns = (st[..., 1] - st[:, :, :1, 1].min(axis=1, keepdims=True)) * 10
    cyc = st[..., 0] - st[:, :, :1, 0].min(axis=1, keepdims=True)
    prev = 0.0
    for i, name in enumerate(NAMES):
        t = ns[:, :, i] / 1e3
        print(f"{i} {name:28s} median {np.median(t):6.2f} us (+{np.median(t) - prev:5.2f})  p90 {np.percentile(t, 90):6.2f}  cycles median {int(np.median(cyc[:, :, i])):6d}")
        prev = float(np.median(t))
    # how many workgroups are alive at once: sum of lifetimes / span
    life = (st[:, :, 6, 1].max(axis=1) - st[:, :, 0, 1].min(axis=1)) * 10 / 1e3
    print(f"workgroup lifetime median {np.median(life):.2f} us; mean concurrency {life.sum() / ((st[:, :, 6, 1].max() - t0) * 10 / 1e3):.0f} workgroups")


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:]])
