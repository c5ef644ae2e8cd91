#!/usr/bin/env python3
"""Phase stamps of conv3_cs2_kernel (64 -> 64 3x3, segment form) on its own shape: where a wave's time goes, and whether the two
waves that share a SIMD are in the same phase.  DIAGNOSTIC build only:

    make -C yolo-lite_amd/csrc -B conv3_cs.o EXTRA=-DYL_STAMPS && make -C yolo-lite_amd/csrc && python tools/cs_stamps.py [H W batch]

Stamp ids (lane 0 of every wave, {s_memtime, s_memrealtime}): 0 start, 1 filter + DMA requested, 2 everything landed, 3 barrier
passed, then per tile 4 + 2k = MFMA loop issued, 5 + 2k = epilogue issued; id 31 = {HW_ID, XCC_ID}."""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch

from yololite_amd import _lib

_lib.apply_runtime_knobs()
from hip_util import Mini, bf16_round  # noqa: E402
from yololite_amd.nn.lower import OP_CONV  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 80
W = int(sys.argv[2]) if len(sys.argv) > 2 else 80
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
V3 = os.environ.get("YL_CS_V3", "1") != "0"  # conv3_cs3_kernel: one workgroup of eight waves per CU; else conv3_cs2_kernel (two of four)
NWG, WAVES = (256, 8) if V3 else (512, 4)
stamp_buf = torch.zeros(NWG * WAVES * 32 * 2, dtype=torch.int64, device=dev)
_lib.lib().yl_diag_set_stamps(C.c_void_p(stamp_buf.data_ptr()))  # (its return value says whether engine.o is a diagnostic object)
os.environ["YL_CONV_CS"] = "1"
os.environ["YL_CONV_CS_MIN"] = "1"
m = Mini(B, dev)
xin, y = m.buf(H, W, 64), m.buf(H, W, 64)
m.fill(xin, bf16_round(torch.randn(B, 64, H, W, generator=gen)))
wt = torch.randn(64, 64, 3, 3, generator=gen) / (64 * 9) ** 0.5
m.op(kind=OP_CONV, name="c", k=3, stride=1, act=1, inp=xin, out=y, group=1, weight=np.ascontiguousarray(wt.numpy()),
     bias=np.zeros(64, np.float32))
m.run()
print("kernel:", m.eng.op_kernel(0))
im = torch.zeros(B, 3, 32, 32, device=dev)
for _ in range(5):
    m.eng.forward(im, decode=False)
torch.cuda.synchronize()
st = stamp_buf.view(NWG, WAVES, 32, 2).cpu().numpy().astype(np.int64)
live = st[:, 0, 0, 1] > 0
if not live.any():
    sys.exit("no stamps: conv3_cs.o is not a diagnostic object (make -B conv3_cs.o EXTRA=-DYL_STAMPS)")
st = st[live]
n = st.shape[0]
hw = st[:, :, 31, 0]
xcc = st[:, :, 31, 1] & 0xF
slot, simd, cu, se = hw & 0xF, (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 13) & 7
clk = st[..., 0]  # shader clock (s_memtime)
rt = st[..., 1]   # 100 MHz
t0 = rt[:, :, 0].min()
print(f"{n} workgroups; start spread {(rt[:, :, 0].max() - t0) * 10 / 1e3:.2f} us; last stamp {(rt[:, :, :31].max() - t0) * 10 / 1e3:.2f} us after the first start")
print("wave slots seen:", np.unique(slot), " SIMDs:", np.unique(simd))
names = ["start", "requested", "filter in LDS", "filter in regs"] if V3 else ["start", "requested", "landed", "barrier"]
ntile = 0
for k in range(12):
    if (clk[:, :, 4 + 2 * k] > 0).any():
        ntile = k + 1
for k in range(ntile):
    names += [f"tile{k} mfma", f"tile{k} epi"]
# per-wave phase durations in shader clocks (s_memtime counts at the constant 100 MHz x ... on gfx9: use realtime for us, memtime for order)
ref = rt[:, :, 0:1]
for i, nm in enumerate(names):
    v = (rt[:, :, i] - t0) * 10 / 1e3
    ok = rt[:, :, i] > 0
    if ok.any():
        print(f"  {nm:14s} median {np.median(v[ok]):6.2f} us  p10 {np.percentile(v[ok], 10):6.2f}  p90 {np.percentile(v[ok], 90):6.2f}  max {v[ok].max():6.2f}  (waves with it: {ok.mean() * 100:.0f} %)")
if V3:
    v = (rt[:, :, 30] - t0) * 10 / 1e3
    print(f"  {'2nd barrier':14s} median {np.median(v):6.2f} us  p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
# in-kernel clock over the compute part (guide: delta s_memtime / delta s_memrealtime x 100 MHz)
last = 3 + 2 * ntile
okc = (clk[:, :, last] > 0) & (rt[:, :, last] > rt[:, :, 3])
ghz = (clk[:, :, last] - clk[:, :, 3])[okc] / ((rt[:, :, last] - rt[:, :, 3])[okc] * 10.0)
print(f"in-kernel clock from the barrier to the last epilogue: median {np.median(ghz):.2f} GHz (p10 {np.percentile(ghz, 10):.2f}, p90 {np.percentile(ghz, 90):.2f})")
# durations of the phases by shader clock
print("phase durations (s_memtime ticks -> us at 100 MHz if constant; else relative):")
for i in range(1, len(names)):
    ok = (clk[:, :, i] > 0) & (clk[:, :, i - 1] > 0)
    d = (clk[:, :, i] - clk[:, :, i - 1])[ok]
    print(f"  {names[i - 1]:12s} -> {names[i]:12s} median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}")
# pairs of waves on one SIMD: overlap of MFMA phases
key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
flat_key = key.reshape(-1)
flat_slot = slot.reshape(-1)
flat_clk = clk.reshape(-1, 32)
order = np.argsort(flat_key, kind="stable")
pairs = {}
for idx in order:
    pairs.setdefault(int(flat_key[idx]), []).append(idx)
sizes = np.bincount([len(v) for v in pairs.values()])
print("waves per (xcc, se, cu, simd):", {i: int(c) for i, c in enumerate(sizes) if c})
both_m = []
tot = []
for v in pairs.values():
    if len(v) != 2:
        continue
    a_, b_ = flat_clk[v[0]], flat_clk[v[1]]

    def mf(c):
        iv = []
        for k in range(ntile):
            if c[4 + 2 * k] > 0:
                s = c[3] if k == 0 else c[5 + 2 * (k - 1)]
                iv.append((s, c[4 + 2 * k]))
        return iv

    ia, ib = mf(a_), mf(b_)
    ov = 0
    for s1, e1 in ia:
        for s2, e2 in ib:
            ov += max(0, min(e1, e2) - max(s1, s2))
    la = sum(e - s for s, e in ia)
    both_m.append(ov / max(1, la))
    tot.append(la)
if both_m:
    print(f"pairs on one SIMD: {len(both_m)}; share of a wave's MFMA-phase time during which its neighbour is ALSO in an MFMA phase: "
          f"median {np.median(both_m):.2f}, p10 {np.percentile(both_m, 10):.2f}, p90 {np.percentile(both_m, 90):.2f}")
