"""Phase breakdown of psa_tail_kernel (csrc/psa_tail.hip) from its in-kernel stamps, inside the latency profile of the network.

Needs a library built with -DYL_STAMPS (rm csrc/psa_tail.o csrc/engine.o; make -C yolo-lite_amd/csrc EXTRA=-DYL_STAMPS).
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

NAMES = ["start", "o, b loaded + three filters staged", "barrier", "proj", "ffn.0 + ffn.1", "cv2's filter staged, a loaded", "barrier", "cv2 + stores"]


def main(B=32, H=640, W=640):
    dev = torch.device("cuda:0")
    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    eng = HipEngine.from_model(specs, synth_state_dict("n", 0), B, H, W, dev, one_batch=True)
    x = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device=dev)
    import os

    per = 64 if os.environ.get("YL_PSA_TAIL_NT", "1") == "1" and B * (H // 32) * (W // 32) <= 256 * 64 else 128
    if os.environ.get("YL_PSA_TAIL_NT") == "2":
        per = 128
    nwg = (B * (H // 32) * (W // 32) + per - 1) // per
    buf = torch.zeros(nwg * 4 * 32 * 2, dtype=torch.int64, device=dev)
    L = _lib.lib()
    for _ in range(3):
        eng.forward(x)
        torch.cuda.synchronize()
    if not L.yl_diag_set_stamps(C.c_void_p(buf.data_ptr())):
        print("this libyl_amd.so is the production build: rebuild with EXTRA=-DYL_STAMPS")
        return
    eng.forward(x)
    torch.cuda.synchronize()
    L.yl_diag_set_stamps(None)
    st = buf.view(nwg, 4, 32, 2).cpu().numpy().astype(np.int64)
    t0 = st[:, :, 0, 1].min()
    print(f"{nwg} workgroups; kernel span {(st[:, :, 7, 1].max() - t0) * 10 / 1e3:.1f} us; starts spread over {(st[:, :, 0, 1].max() - t0) * 10 / 1e3:.1f} us")
    ns = (st[..., 1] - t0) * 10
    prev = 0.0
    for i, name in enumerate(NAMES):
        t = ns[:, :, i] / 1e3
        print(f"{i} {name:36s} median {np.median(t):6.2f} us (+{np.median(t) - prev:5.2f})  max {t.max():6.2f}")
        prev = float(np.median(t))


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:]])
