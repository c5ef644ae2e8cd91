"""Phase breakdown of decode_nms_fused_kernel (csrc/nms.hip) from its in-kernel stamps, inside the network's last launch.

Needs a library built with -DYL_STAMPS (rm csrc/nms.o csrc/engine.o; make -C yolo-lite_amd/csrc EXTRA=-DYL_STAMPS).
YL_NMS_LDS=0 shows the general path (intermediates through the NMS workspace).
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

NAMES = ["start", "list lengths read", "exact decode", "sorted rows", "suppression mask", "scan + output"]


def main(B=32, H=640, W=640):
    dev = torch.device("cuda:0")
    import bench  # bench.py's head calibration: ~126 candidates per image

    x = torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(0), dtype=torch.uint8).to(dev)
    _, _, eng = bench.build_model("n", dev, B, H, W, x)
    buf = torch.zeros(B * 16 * 32 * 2, dtype=torch.int64, device=dev)
    L = _lib.lib()
    for _ in range(3):
        out, cnt = eng.detect(x, 0.25, 0.7, graph=False)
        torch.cuda.synchronize()
    if not L.yl_diag_set_stamps(C.c_void_p(buf.data_ptr())):
        print("this libyl_amd.so is the production build: rebuild with EXTRA=-DYL_STAMPS")
        return
    out, cnt = eng.detect(x, 0.25, 0.7, graph=False)
    torch.cuda.synchronize()
    L.yl_diag_set_stamps(None)
    st = buf.view(B, 16, 32, 2).cpu().numpy().astype(np.int64)
    print(f"detections per image: mean {cnt.float().mean().item():.1f}")
    t0 = st[:, :, 0, 1].min()
    print(f"kernel span {(st[:, :, 5, 1].max() - t0) * 10 / 1e3:.1f} us; starts spread over {(st[:, :, 0, 1].max() - t0) * 10 / 1e3:.1f} us")
    ns = (st[..., 1] - st[:, :, :1, 1].min(axis=1, keepdims=True)) * 10
    inner = ["list entry read", "anchor record", "maps loaded", "class scores", "DFL + shuffles"]
    print("inside the exact decode (first round of 256 candidates; waves that have candidates):")
    act = st[:, :, 6, 1] > 0
    for k, name in enumerate(inner):
        t = np.where(act, ns[:, :, 6 + k], 0).max(axis=1) / 1e3
        print(f"  {6 + k} {name:18s} median {np.median(t):6.2f} us  max {t.max():6.2f}")
    prev = 0.0
    for i, name in enumerate(NAMES):
        t = ns[:, :, i].max(axis=1) / 1e3  # the last wave of the image
        print(f"{i} {name:20s} median {np.median(t):6.2f} us (+{np.median(t) - prev:5.2f})  max {t.max():6.2f}")
        prev = float(np.median(t))


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:]])
