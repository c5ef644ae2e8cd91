"""Phase breakdown of the resident kernels (csrc/resident.hip: C3k, `psa`; csrc/sppf_resident.hip: `sppf`) from their in-kernel stamps.

Needs a library built with -DYL_STAMPS (on the GPU box: rm csrc/resident.o csrc/engine.o; make -C yolo-lite_amd/csrc EXTRA=-DYL_STAMPS).
Prints, per phase boundary, the median / max over workgroups and waves of the time since the kernel's first stamp, in
shader cycles and in microseconds (100 MHz clock), plus the clock the kernel ran at.
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from yololite_amd import _lib  # noqa: E402

_lib.apply_runtime_knobs()
from test_gpu_resident import _rand_block, _np  # noqa: E402
from hip_util import Mini, bf16_round  # noqa: E402
from yololite_amd.nn.lower import OP_C3K  # noqa: E402

NAMES = ["start", "x staged", "barrier", "cv1|cv2 done", "barrier", "T cleared", "m0.cv1 done", "barrier", "m0.cv2 done", "barrier",
         "m1.cv1 done", "barrier", "m1.cv2 done", "barrier", "cv3 done"]


PSA_NAMES = ["start", "qkv h0 done", "barrier", "attention h0 done", "barrier", "o map + pe h0", "qkv h1 done", "barrier", "attention h1 done",
             "barrier", "o map + pe h1", "tail filters landed", "barrier", "proj done", "ffn done"]


def psa(B=32, H=20, W=20, iters=5):
    """The same for the resident PSABlock kernel."""
    from test_gpu_resident import _psa_ops, _rand_psa

    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    blk = _rand_psa(gen)
    m = Mini(B, dev)
    u, u2 = m.buf(H, W, 256), m.buf(H, W, 256)
    bv = u.slice(128, 128)
    m.fill(bv, bf16_round(torch.randn(B, 128, H, W, generator=gen)))
    _psa_ops(m, bv, blk, H, W, True, u=u, u2=u2)
    buf = torch.zeros(4 * B * 8 * 32 * 2, dtype=torch.int64, device=dev)
    L = _lib.lib()
    if not L.yl_diag_set_stamps(C.c_void_p(buf.data_ptr())):
        print("this libyl_amd.so is the production build: rebuild with EXTRA=-DYL_STAMPS")
        return
    m.run()
    im = torch.zeros(B, 3, 32, 32, device=dev)
    for _ in range(iters):
        m.eng.forward(im, decode=False)
        torch.cuda.synchronize()
    st = buf.view(4 * B, 8, 32, 2).cpu().numpy().astype(np.int64)
    st = st[st[:, 0, 14, 1] > 0]  # the workgroups that ran (B x workgroups per image)
    print(f'{len(st)} workgroups')
    ns = (st[..., 1] - st[:, :, :1, 1].min(axis=1, keepdims=True)) * 10
    prev = 0.0
    for i, name in enumerate(PSA_NAMES):
        t = ns[:, :, i] / 1e3
        print(f"{i:2d} {name:20s} median {np.median(t):7.2f} us  (+{np.median(t) - prev:6.2f})  max {t.max():7.2f} us   per-wave spread {np.median(t.max(axis=1) - t.min(axis=1)):5.2f} us")
        prev = float(np.median(t))
    L.yl_diag_set_stamps(None)


SPPF_NAMES = ["start", "cv1 done, ring primed", "barrier", "pools done", "cv2 x k-steps", "cv2 pooled k-steps", "tail k-steps", "stored"]


def sppf(B=32, H=20, W=20, iters=5, tail=1):
    """The same for the resident SPPF (+ follow-up conv) kernel."""
    from test_gpu_resident import _rand_sppf
    from yololite_amd.nn.lower import OP_SPPF

    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    blk = _rand_sppf(gen)
    m = Mini(B, dev)
    xin, out = m.buf(H, W, 256), m.buf(H, W, 256)
    m.fill(xin, bf16_round(torch.randn(B, 256, H, W, generator=gen)))
    m.op(kind=OP_SPPF, name="sppf", act=1, inp=xin, out=out, weight=_np(blk["cv1"][0].reshape(128, -1)), bias=_np(blk["cv1"][1]),
         extra_weight=[_np(blk["cv2"][0].reshape(256, -1))] + ([_np(blk["tail"][0].reshape(256, -1))] if tail else []),
         extra_bias=[_np(blk["cv2"][1])] + ([_np(blk["tail"][1])] if tail else []))
    buf = torch.zeros(2 * B * 8 * 32 * 2, dtype=torch.int64, device=dev)
    L = _lib.lib()
    if not L.yl_diag_set_stamps(C.c_void_p(buf.data_ptr())):
        print("this libyl_amd.so is the production build: rebuild with EXTRA=-DYL_STAMPS")
        return
    m.run()
    im = torch.zeros(B, 3, 32, 32, device=dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(iters):
        ev0.record()
        m.eng.forward(im, decode=False)
        ev1.record()
        torch.cuda.synchronize()
    print(f"B={B} {H}x{W} tail={tail}: last launch {ev0.elapsed_time(ev1) * 1e3:.1f} us (event to event)")
    st = buf.view(2 * B, 8, 32, 2).cpu().numpy().astype(np.int64)
    st = st[st[:, 0, 7, 1] > 0]
    print(f"{len(st)} workgroups")
    ns = (st[..., 1] - st[:, :, :1, 1].min(axis=1, keepdims=True)) * 10
    prev = 0.0
    for i, name in enumerate(SPPF_NAMES):
        if not tail and i == 6:
            continue
        t = ns[:, :, i] / 1e3
        print(f"{i:2d} {name:24s} median {np.median(t):7.2f} us  (+{np.median(t) - prev:6.2f})  max {t.max():7.2f} us   per-wave spread {np.median(t.max(axis=1) - t.min(axis=1)):5.2f} us")
        prev = float(np.median(t))
    L.yl_diag_set_stamps(None)


def main(B=32, H=20, W=20, iters=5):
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    blk = _rand_block(gen)
    x = bf16_round(torch.randn(B, 128, H, W, generator=gen))
    m = Mini(B, dev)
    tin = m.buf(H, W, 384)
    xin, yv = tin.slice(128, 128), tin.slice(256, 128)
    m.fill(xin, x)
    m.op(kind=OP_C3K, name="c3k", act=1, inp=xin, out=yv, weight=_np(blk["cv12"][0].reshape(128, -1)), bias=_np(blk["cv12"][1]),
         extra_weight=[_np(w) for w, _ in blk["m"]] + [_np(blk["cv3"][0].reshape(128, -1))],
         extra_bias=[_np(b) for _, b in blk["m"]] + [_np(blk["cv3"][1])])
    buf = torch.zeros(B * 8 * 32 * 2, dtype=torch.int64, device=dev)
    L = _lib.lib()
    on = L.yl_diag_set_stamps(C.c_void_p(buf.data_ptr()))
    if not on:
        print("this libyl_amd.so is the production build: rebuild with EXTRA=-DYL_STAMPS")
        return
    m.run()
    im = torch.zeros(B, 3, 32, 32, device=dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(iters):
        ev0.record()
        m.eng.forward(im, decode=False)
        ev1.record()
        torch.cuda.synchronize()
    print(f"B={B} {H}x{W}: last launch {ev0.elapsed_time(ev1) * 1e3:.1f} us (event to event)")
    st = buf.view(B, 8, 32, 2).cpu().numpy().astype(np.int64)
    cyc = st[..., 0] - st[:, :, :1, 0].min(axis=1, keepdims=True)   # since the workgroup's earliest wave start
    ns = (st[..., 1] - st[:, :, :1, 1].min(axis=1, keepdims=True)) * 10
    clk = (st[:, :, 14, 0] - st[:, :, 0, 0]) / np.maximum((st[:, :, 14, 1] - st[:, :, 0, 1]) * 10, 1)
    print(f"clock {np.median(clk):.2f} GHz; workgroup start spread {(st[:, 0, 0, 1].max() - st[:, 0, 0, 1].min()) * 10} ns")
    prev = 0.0
    for i, name in enumerate(NAMES):
        c = cyc[:, :, i]
        t = ns[:, :, i] / 1e3
        print(f"{i:2d} {name:14s} median {np.median(t):7.2f} us  (+{np.median(t) - prev:6.2f})  max {t.max():7.2f} us   cycles median {int(np.median(c)):7d}"
              f"   per-wave spread {np.median(t.max(axis=1) - t.min(axis=1)):5.2f} us")
        prev = float(np.median(t))
    L.yl_diag_set_stamps(None)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "psa":
        psa(*[int(v) for v in sys.argv[2:]])
    elif len(sys.argv) > 1 and sys.argv[1] == "sppf":
        sppf(*[int(v) for v in sys.argv[2:]])
    else:
        main(*[int(v) for v in sys.argv[1:]])
