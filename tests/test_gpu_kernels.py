"""-m gpu: each HIP kernel against a plain PyTorch fp32 reference of the same op.

Inputs and weights are rounded to bf16 first, so the only differences left are fp32 accumulation
order and the final bf16 rounding of the output (<= 2^-8 relative): tolerances below say so.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hip_util import Mini, bf16_round, run_conv
from yololite_amd.nn.lower import BF16, F32, OP_ATTENTION, OP_CONV, OP_DWCONV3, OP_SPPF_POOL, OP_UPSAMPLE

pytestmark = pytest.mark.gpu


def _rand(g, *shape, scale=1.0):
    return bf16_round(torch.randn(*shape, generator=g) * scale)


def _assert_close_bf16(got, ref, what=""):
    # output is bf16: half an ulp is 2^-9 relative; accumulation-order noise is far below that
    tol = 2.0 ** -8 * ref.abs().clamp(min=1e-2) + 1e-3
    err = (got - ref).abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} / {bad.numel()} off, max err {float(err.max()):.4g}"


CONV_CASES = [
    # Cin, Cout, k, s, H, W, act, res      (YOLO11 n/s/x shapes)
    (8, 16, 3, 2, 64, 64, True, False),     # stem (Cin padded 3 -> 8)
    (16, 32, 3, 2, 40, 40, True, False),
    (16, 8, 3, 1, 40, 40, True, False),     # Bottleneck cv1 of layer 2 (Cout 8 -> half an MFMA tile)
    (8, 16, 3, 1, 40, 40, True, True),      # Bottleneck cv2 + shortcut
    (32, 32, 1, 1, 40, 40, True, False),
    (48, 64, 1, 1, 40, 40, True, False),    # C3k2 cv2, Cin 48 (non-uniform K path)
    (64, 64, 3, 2, 40, 40, True, False),
    (64, 64, 3, 1, 20, 20, True, True),
    (128, 128, 3, 2, 20, 20, True, False),
    (128, 256, 3, 2, 20, 20, True, False),
    (256, 64, 3, 1, 20, 20, True, False),   # Detect cv2[2][0], chunked Cin
    (384, 128, 1, 1, 20, 20, True, False),
    (512, 256, 1, 1, 10, 10, True, False),  # SPPF cv2
    (80, 80, 1, 1, 20, 20, False, False),   # Detect class head (bias, no act)
    (96, 128, 1, 1, 12, 20, True, False),
    (48, 48, 3, 1, 12, 20, True, True),     # x-scale C3k bottleneck
    (96, 96, 3, 1, 12, 20, True, True),     # x-scale widths of 96: six channel tiles per wave, slab restaged in chunks
    (192, 96, 1, 1, 16, 24, True, False),   # C3k cv1+cv2 of YOLO11x (six tiles, 1x1)
    (64, 96, 3, 2, 24, 24, True, False),
    (128, 192, 1, 1, 16, 24, True, True),   # wide layers prefer six tiles to four (two channel blocks of 96)
    (256, 384, 3, 2, 16, 16, True, False),
    (64, 64, 3, 1, 7, 9, True, False),      # ragged map
    (32, 16, 3, 1, 33, 17, True, False),
]


@pytest.mark.parametrize("cin,cout,k,s,h,w,res", [(256, 256, 1, 1, 20, 20, True), (128, 256, 3, 2, 24, 40, False), (128, 512, 3, 1, 9, 13, True)])
def test_conv_eight_tiles_per_wave(gpu_device, monkeypatch, cin, cout, k, s, h, w, res):
    """128 channels x 32 pixels per wave (MT = 8, NT = 2: half the activation re-reads of 64 x 64), the tiling wide layers with
    at least 256 output channels take when the grid stays large (YOLO11s at batch 64); forced here on small maps."""
    monkeypatch.setenv("YL_CONV_MT8_BLOCKS", "1")
    test_conv_vs_torch(gpu_device, cin, cout, k, s, h, w, True, res)
    monkeypatch.setenv("YL_CONV_MT8", "0")   # and the four-tile form of the same layer agrees with torch as well
    test_conv_vs_torch(gpu_device, cin, cout, k, s, h, w, True, res)


GEMM_CASES = [
    # cin, cout, h, w, act, res
    (128, 128, 20, 20, True, False),    # 128-channel blocks, four K-slices (= the ring's depth)
    (160, 128, 20, 12, True, False),    # five K-slices
    (256, 256, 12, 20, True, True),     # residual
    (192, 192, 16, 24, True, False),    # 192-channel blocks (YOLO11x widths)
    (384, 768, 9, 13, False, False),    # qkv of C2PSA: no activation, four blocks of 192
    (768, 384, 11, 13, True, True),     # 24 K-slices, ragged map
    (512, 256, 12, 12, True, False),    # SPPF cv2: 256-channel blocks, ring of three 32-channel slices
    (128, 384, 33, 17, False, True),
    (256, 64, 24, 20, True, False),     # 64-channel blocks (round 5: two tiles per wave, ring of three 64-channel slices)
    (160, 64, 17, 23, True, True),      # ... K not a multiple of 64 (four slices of 32), residual, ragged map
    (128, 64, 20, 20, False, False),    # ... two K-slices = exactly the ring's lead
]


@pytest.mark.parametrize("w8", ["0", "-1"])
@pytest.mark.parametrize("cin,cout,h,w,act,res", GEMM_CASES)
def test_conv_gemm_vs_torch_and_conv_kernel(gpu_device, monkeypatch, cin, cout, h, w, act, res, w8):
    """conv_gemm_kernel (wide 1x1 convs: both operands staged through LDS by LDS-DMA rings, 128 / 192 / 256-channel x 128-pixel
    macro tiles with four waves, x 256 pixels with eight: w8) against torch, and BIT-identical to conv_kernel on the same
    operands (same packed filters' k order, same accumulation chain)."""
    monkeypatch.setenv("YL_CONV_GEMM_W8", w8)
    g = torch.Generator().manual_seed(cin * 7 + cout + h)
    B = 5
    x = _rand(g, B, cin, h, w)
    wt = _rand(g, cout, cin, 1, 1, scale=1.0 / math.sqrt(cin))
    b = torch.randn(cout, generator=g) * 0.5
    ref = F.conv2d(x, wt, b)
    if act:
        ref = F.silu(ref)
    r = None
    if res:
        r = _rand(g, *ref.shape)
        ref = ref + r
    monkeypatch.setenv("YL_CONV_GEMM_BLOCKS", "1")
    monkeypatch.setenv("YL_CONV_GEMM_M64_PX", "1")  # (64-channel blocks whatever the map size)
    got = run_conv(gpu_device, x, wt, b, 1, 1, act, r, expect_kernel="conv1x1_gemm_mfma")
    _assert_close_bf16(got, ref, f"gemm conv {cin}->{cout}")
    monkeypatch.setenv("YL_CONV_GEMM", "0")
    direct = run_conv(gpu_device, x, wt, b, 1, 1, act, r, expect_kernel="conv1x1_mfma")
    assert torch.equal(got, direct), f"gemm vs conv_kernel differ: max {float((got - direct).abs().max())}"


GEMM3_CASES = [
    # Cin, Cout, stride, H, W, act, res      conv_gemm_kernel on 3x3 convs (nine shifted GEMMs; stride 2: per-lane source pixel)
    (64, 128, 2, 24, 32, True, False),     # YOLO11s layer 3 (64 -> 128): K = 18 k-steps, 128-channel blocks
    (128, 256, 2, 26, 22, True, False),    # layer 5: 256-channel blocks (ring of three slices)
    (128, 128, 2, 21, 35, True, False),    # odd input size: the last output row / column read the guards
    (128, 128, 1, 12, 12, True, True),     # stride 1 with the Bottleneck residual
    (64, 192, 1, 13, 11, False, False),    # 192-channel blocks, no activation
    (96, 128, 2, 30, 18, True, False),     # 96 channels: 27 k-steps (odd: ring of four)
]


@pytest.mark.parametrize("w8", ["0", "-1"])
@pytest.mark.parametrize("cin,cout,s,h,w,act,res", GEMM3_CASES)
def test_conv_gemm_3x3_vs_torch_and_conv_kernel(gpu_device, monkeypatch, cin, cout, s, h, w, act, res, w8):
    """The GEMM-class kernel on 3x3 convs against torch and against conv_kernel (YL_CONV_GEMM3=0).  Its K order -- (block
    of four planes, tap) -- differs from conv_kernel's, so the two agree to fp32 summation order, not bit for bit."""
    monkeypatch.setenv("YL_CONV_GEMM_W8", w8)
    monkeypatch.setenv("YL_CONV_GEMM_BLOCKS", "1")
    g = torch.Generator().manual_seed(cin * 3 + cout + h + s)
    B = 5
    x = _rand(g, B, cin, h, w)
    wt = _rand(g, cout, cin, 3, 3, scale=1.0 / math.sqrt(cin * 9))
    b = torch.randn(cout, generator=g) * 0.5
    ref = F.conv2d(x, wt, b, s, 1)
    if act:
        ref = F.silu(ref)
    r = None
    if res:
        r = _rand(g, *ref.shape)
        ref = ref + r
    name = "conv3x3s2_gemm_mfma" if s == 2 else "conv3x3_gemm_mfma"
    got = run_conv(gpu_device, x, wt, b, 3, s, act, r, expect_kernel=name)
    _assert_close_bf16(got, ref, f"gemm 3x3 conv {cin}->{cout} s{s}")
    monkeypatch.setenv("YL_CONV_GEMM3", "0")
    direct = run_conv(gpu_device, x, wt, b, 3, s, act, r)
    # same operands, different summation order: one bf16 ulp of the result plus the fp32 reordering noise
    assert bool(((got - direct).abs() <= 2.0 ** -7 * direct.abs() + 2e-3).all()), float((got - direct).abs().max())


@pytest.mark.parametrize("cin,cout,k,s,h,w,act,res", CONV_CASES)
def test_conv_vs_torch(gpu_device, cin, cout, k, s, h, w, act, res):
    g = torch.Generator().manual_seed(cin * 1000 + cout * 10 + k + s)
    B = 3
    x = _rand(g, B, cin, h, w)
    wt = _rand(g, cout, cin, k, k, scale=1.0 / math.sqrt(cin * k * k))
    b = torch.randn(cout, generator=g) * 0.5
    ref = F.conv2d(x, wt, b, s, k // 2)
    if act:
        ref = F.silu(ref)
    r = None
    if res:
        r = _rand(g, *ref.shape)
        ref = ref + r
    got = run_conv(gpu_device, x, wt, b, k, s, act, r)
    _assert_close_bf16(got, ref, f"conv {cin}->{cout} k{k} s{s}")


def test_conv_slices_and_f32_out(gpu_device):
    """Reads a channel slice, writes a channel slice of an fp32 buffer (Detect head layout)."""
    g = torch.Generator().manual_seed(7)
    x = _rand(g, 2, 64, 20, 20)
    wt = _rand(g, 80, 64, 1, 1, scale=0.125)
    b = torch.randn(80, generator=g)
    ref = F.conv2d(x, wt, b)
    got = run_conv(gpu_device, x, wt, b, 1, 1, False, None, pitch_extra=(16, 64), out_f32=True)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)  # fp32 out: order noise only


@pytest.mark.parametrize("cin,cmid,cout,k,s,h,w,act2,f32", [(16, 32, 32, 3, 2, 40, 40, True, False),
                                                             (64, 64, 64, 3, 1, 20, 24, False, True),
                                                             (32, 64, 48, 1, 1, 13, 9, True, False)])
def test_conv_with_fused_follow_up_1x1(gpu_device, cin, cmid, cout, k, s, h, w, act2, f32):
    """conv -> SiLU -> (bf16) -> 1x1 conv in ONE launch (yl_op.post_*): the intermediate tensor only
    exists in registers, rounded to bf16 exactly like a stored activation would be."""
    g = torch.Generator().manual_seed(cin + cmid + cout)
    x = _rand(g, 2, cin, h, w)
    w1 = _rand(g, cmid, cin, k, k, scale=1.0 / math.sqrt(cin * k * k))
    b1 = torch.randn(cmid, generator=g) * 0.5
    w2 = _rand(g, cout, cmid, 1, 1, scale=1.0 / math.sqrt(cmid))
    b2 = torch.randn(cout, generator=g) * 0.5
    mid = bf16_round(F.silu(F.conv2d(x, w1, b1, s, k // 2)))
    ref = F.conv2d(mid, w2, b2)
    if act2:
        ref = F.silu(ref)
    m = Mini(2, gpu_device)
    xin = m.buf(h, w, cin)
    yo = m.buf(ref.shape[2], ref.shape[3], cout, F32 if f32 else BF16)
    m.fill(xin, x)
    m.op(kind=OP_CONV, name="t", k=k, stride=s, act=1, inp=xin, out=yo, weight=np.ascontiguousarray(w1.numpy()),
         bias=np.ascontiguousarray(b1.numpy()), post_weight=np.ascontiguousarray(w2.numpy().reshape(cout, cmid)),
         post_bias=np.ascontiguousarray(b2.numpy()), post_act=int(act2))
    m.run()
    got = m.read(yo)
    # the intermediate's bf16 rounding can flip by one ulp (accumulation order): |w2| * 2^-8 * |mid| headroom
    tol = 2.0 ** -7 * ref.abs() + 2.0 ** -7 * F.conv2d(mid.abs(), w2.abs()) + 1e-3
    assert bool(((got - ref).abs() <= tol).all()), float(((got - ref).abs() - tol).max())


@pytest.mark.parametrize("form,cin,B,h,w,stride,keep", [("post", 64, 2, 20, 24, 8.0, False), ("post", 64, 3, 7, 9, 16.0, True),
                                                        ("post", 128, 2, 13, 5, 32.0, False), ("plain", 64, 2, 20, 20, 8.0, True),
                                                        ("plain", 96, 2, 12, 20, 16.0, False), ("plain", 64, 33, 2, 2, 32.0, False)])
def test_dfl_epilogue_vs_torch(gpu_device, form, cin, B, h, w, stride, keep):
    """Detect's box decode in the epilogue of the conv that computes a level's 64 DFL logits (yl_op.dfl_level, csrc/conv_common.h
    dfl_epilogue; reference head.py:95-126, block.py:51-69, tal.py:341-350): the 1x1 conv alone ("plain": YOLO11x, whose 96-channel
    box branch cannot fuse it) or fused behind a 3x3 conv ("post": cv2[i][1] + cv2[i][2] of YOLO11n / s) against torch fp32 -- the
    logits from the same bf16-rounded operands, then softmax expectation and dist2bbox(xywh) * stride in fp32.  The logits never
    pass through bf16, so the boxes agree to fp32 accumulation order: |d| <= 2e-4 * stride + 1e-5 * |value|.  keep: the bf16
    logits are stored as well (the parity tests' program) and equal the rounded torch logits."""
    from oracle import yolo11_oracle as oracle

    g = torch.Generator().manual_seed(cin + h + w)
    x = _rand(g, B, cin, h, w)
    w2 = _rand(g, 64, 64 if form == "post" else cin, 1, 1, scale=2.0 / math.sqrt(64))
    b2 = torch.randn(64, generator=g)
    m = Mini(B, gpu_device)
    xin, yo, cls = m.buf(h, w, cin), m.buf(h, w, 64), m.buf(h, w, 80)
    off = m.ar.alloc_bytes(B * h * w * 16)
    m.prog.levels.append(dict(offset=yo.offset if keep else -1, offset_cls=cls.offset, offset_xywh=off, nc8=80, H=h, W=w, stride=stride))
    m.prog.anchors = h * w
    m.fill(xin, x)
    if form == "post":
        w1 = _rand(g, 64, cin, 3, 3, scale=1.0 / math.sqrt(cin * 9))
        b1 = torch.randn(64, generator=g) * 0.5
        mid = bf16_round(F.silu(F.conv2d(x, w1, b1, 1, 1)))
        logits = F.conv2d(mid, w2, b2)
        m.op(kind=OP_CONV, name="t", k=3, stride=1, act=1, inp=xin, out=yo, weight=np.ascontiguousarray(w1.numpy()),
             bias=np.ascontiguousarray(b1.numpy()), post_weight=np.ascontiguousarray(w2.numpy().reshape(64, 64)),
             post_bias=np.ascontiguousarray(b2.numpy()), post_act=0, dfl_level=1, dfl_keep=int(keep))
    else:
        logits = F.conv2d(x, w2, b2)
        m.op(kind=OP_CONV, name="t", k=1, stride=1, act=0, inp=xin, out=yo, weight=np.ascontiguousarray(w2.numpy()),
             bias=np.ascontiguousarray(b2.numpy()), dfl_level=1, dfl_keep=int(keep))
    m.run()
    anchors, _ = oracle.make_anchors([(h, w)], [stride])
    ref = oracle.dist2bbox_xywh(oracle.dfl(logits.reshape(B, 64, h * w)), anchors.transpose(0, 1).unsqueeze(0)) * stride
    got = m.eng.level_boxes()[0].cpu()
    assert got.shape == ref.shape == (B, 4, h * w)
    tol = 2e-4 * stride + 1e-5 * ref.abs()
    if form == "post":  # (the intermediate's bf16 rounding can flip by one ulp with the accumulation order: a logit moves by
        tol = tol + 0.02 * stride  # |w2| 2^-8 |mid|, the expectation by a few hundredths of a bin)
    err = (got - ref).abs()
    assert bool((err <= tol).all()), f"max box error {float(err.max()):.5f} px (stride {stride})"
    if keep:
        _assert_close_bf16(m.read(yo), logits, "kept DFL logits")
        # ... and the readers decode from the boxes, not from the kept logits: yl_detect_decode returns them
        y = torch.empty(B, 84, h * w, device=gpu_device)
        from yololite_amd import _lib
        _lib.check(m.eng.lib.yl_detect_decode(m.eng.workspace.data_ptr(), m.eng._levels, 1, B, 80, y.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "yl_detect_decode")
        torch.cuda.synchronize()
        assert torch.equal(y[:, :4].cpu(), got)


@pytest.mark.parametrize("B,h,w,stride,keep", [(2, 20, 24, 8.0, True), (3, 7, 9, 16.0, False), (32, 80, 80, 8.0, False), (5, 33, 150, 8.0, True),
                                                (8, 79, 79, 8.0, False)])
def test_conv3_cs_post_dfl_vs_conv_kernel(gpu_device, monkeypatch, B, h, w, stride, keep):
    """Detect's cv2[i][1] + cv2[i][2] + box decode on the Cout-split filter-in-registers kernel (conv3_cs1b_kernel's POST form,
    round 6: the waves exchange their t2 halves through LDS, the 1x1 filter is read from LDS, dfl_box on the fp32 accumulators)
    against conv_kernel's POST + DFL form of the same op: boxes (and, kept, the bf16 logits) bit for bit -- same operands, same k
    order, same accumulation chains.  The last case fills whole rounds of 128-position tiles (no short first round)."""
    g = torch.Generator().manual_seed(B + h + w)
    x = _rand(g, B, 64, h, w)
    w1 = _rand(g, 64, 64, 3, 3, scale=1.0 / math.sqrt(64 * 9))
    b1 = torch.randn(64, generator=g) * 0.5
    w2 = _rand(g, 64, 64, 1, 1, scale=2.0 / math.sqrt(64))
    b2 = torch.randn(64, generator=g)

    def run(cs):
        monkeypatch.setenv("YL_CONV_CS", "1" if cs else "0")
        monkeypatch.setenv("YL_CONV_CS_MIN", "1")
        m = Mini(B, gpu_device)
        xin, yo, cls = m.buf(h, w, 64), m.buf(h, w, 64), m.buf(h, w, 80)
        off = m.ar.alloc_bytes(B * h * w * 16)
        m.prog.levels.append(dict(offset=yo.offset if keep else -1, offset_cls=cls.offset, offset_xywh=off, nc8=80, H=h, W=w, stride=stride))
        m.prog.anchors = h * w
        m.fill(xin, x)
        m.op(kind=OP_CONV, name="t", k=3, stride=1, act=1, inp=xin, out=yo, weight=np.ascontiguousarray(w1.numpy()),
             bias=np.ascontiguousarray(b1.numpy()), post_weight=np.ascontiguousarray(w2.numpy().reshape(64, 64)),
             post_bias=np.ascontiguousarray(b2.numpy()), post_act=0, dfl_level=1, dfl_keep=int(keep))
        m.run()
        assert m.eng.op_kernel(0) == ("conv3x3_cs+1x1+dfl_mfma" if cs else "conv3x3_mfma"), m.eng.op_kernel(0)
        return m.eng.level_boxes()[0].cpu(), (m.read(yo) if keep else None)

    box_cs, log_cs = run(True)
    box_ck, log_ck = run(False)
    assert torch.equal(box_cs, box_ck), float((box_cs - box_ck).abs().max())
    if keep:
        assert torch.equal(log_cs, log_ck)
    # ... and against torch (as test_dfl_epilogue_vs_torch)
    from oracle import yolo11_oracle as oracle

    mid = bf16_round(F.silu(F.conv2d(x, w1, b1, 1, 1)))
    logits = F.conv2d(mid, w2, b2)
    anchors, _ = oracle.make_anchors([(h, w)], [stride])
    ref = oracle.dist2bbox_xywh(oracle.dfl(logits.reshape(B, 64, h * w)), anchors.transpose(0, 1).unsqueeze(0)) * stride
    err = (box_cs - ref).abs()
    assert bool((err <= 2e-4 * stride + 1e-5 * ref.abs() + 0.02 * stride).all()), float(err.max())


def test_conv_reads_through_upsample_concat(gpu_device):
    """1x1 conv over cat(upsample(lo), skip): the conv reads `lo` itself (yl_op.in2); bit-identical to the
    conv over a materialised concat buffer."""
    g = torch.Generator().manual_seed(99)
    lo, skip = _rand(g, 2, 64, 5, 7), _rand(g, 2, 32, 10, 14)
    wt = _rand(g, 48, 96, 1, 1, scale=0.1)
    b = torch.randn(48, generator=g) * 0.3
    ref = F.silu(F.conv2d(torch.cat([F.interpolate(lo, scale_factor=2.0), skip], 1), wt, b))
    m = Mini(2, gpu_device)
    vlo, cat, out = m.buf(5, 7, 64), m.buf(10, 14, 96), m.buf(10, 14, 48)
    m.fill(vlo, lo)
    m.fill(cat.slice(64, 32), skip)
    m.op(kind=OP_CONV, name="t", k=1, stride=1, act=1, inp=cat.slice(64, 32), inp2=vlo, up=2, out=out,
         weight=np.ascontiguousarray(wt.numpy()), bias=np.ascontiguousarray(b.numpy()))
    m.run()
    _assert_close_bf16(m.read(out), ref, "conv over upsample+concat")


@pytest.mark.parametrize("clo,cskip,cout,h,w,w8", [(128, 64, 128, 10, 14, "0"), (256, 128, 384, 12, 22, "0"), (128, 128, 256, 12, 22, "1"),
                                                    (64, 192, 256, 9, 15, "-1"), (128, 128, 64, 11, 13, "0"), (64, 96, 64, 8, 10, "0")])
def test_conv_gemm_reads_through_upsample_concat(gpu_device, monkeypatch, clo, cskip, cout, h, w, w8):
    """conv_gemm_kernel with an upsampled second input (yl_op.in2: the LDS-DMA source address is per lane, pixel (y / 2, x / 2) of
    the low-resolution map): against torch and bit-identical to conv_kernel's read-through."""
    g = torch.Generator().manual_seed(clo + cskip + h)
    B = 3
    lo, skip = _rand(g, B, clo, h, w), _rand(g, B, cskip, 2 * h, 2 * w)
    wt = _rand(g, cout, clo + cskip, 1, 1, scale=1.0 / math.sqrt(clo + cskip))
    b = torch.randn(cout, generator=g) * 0.3
    ref = F.silu(F.conv2d(torch.cat([F.interpolate(lo, scale_factor=2.0), skip], 1), wt, b))
    got = {}
    for mode, kern in (("1", "conv1x1_gemm_mfma"), ("0", "conv1x1_mfma")):
        monkeypatch.setenv("YL_CONV_GEMM", mode)
        monkeypatch.setenv("YL_CONV_GEMM_BLOCKS", "1")
        monkeypatch.setenv("YL_CONV_GEMM_M64_PX", "1")
        monkeypatch.setenv("YL_CONV_GEMM_W8", w8)
        m = Mini(B, gpu_device)
        vlo, cat, out = m.buf(h, w, clo), m.buf(2 * h, 2 * w, clo + cskip), m.buf(2 * h, 2 * w, cout)
        m.fill(vlo, lo)
        m.fill(cat.slice(clo, cskip), skip)
        m.op(kind=OP_CONV, name="t", k=1, stride=1, act=1, inp=cat.slice(clo, cskip), inp2=vlo, up=2, out=out,
             weight=np.ascontiguousarray(wt.numpy()), bias=np.ascontiguousarray(b.numpy()))
        m.run()
        assert m.eng.op_kernel(0) == kern, m.eng.op_kernel(0)
        got[mode] = m.read(out)
    _assert_close_bf16(got["1"], ref, "gemm conv over upsample+concat")
    assert torch.equal(got["1"], got["0"])


@pytest.mark.parametrize("staged", [0, 1])
@pytest.mark.parametrize("c,cout,h,w,act2", [(64, 80, 20, 20, True), (80, 80, 13, 9, True), (32, 32, 40, 40, False),
                                             (128, 128, 40, 40, True), (256, 128, 20, 20, True), (128, 128, 13, 9, False)])
def test_depthwise_pointwise_fused(gpu_device, monkeypatch, c, cout, h, w, act2, staged):
    """DWConv 3x3 + SiLU -> 1x1 conv in one launch (Detect class branch); the depthwise result only exists
    in LDS, rounded to bf16 like a stored activation.  The depthwise conv runs on the matrix pipe (block-diagonal
    filter, bf16 taps).  staged = 1: the opt-in variant that stages the input through LDS (csrc/dwpw.hip).  128 output channels
    (YOLO11s' class branch): eight channel tiles per wave, one block."""
    monkeypatch.setenv("YL_DWPW_TILE", str(staged))
    g = torch.Generator().manual_seed(c * 7 + cout)
    x = _rand(g, 2, c, h, w)
    wd = torch.randn(c, 1, 3, 3, generator=g) * 0.3
    bd = torch.randn(c, generator=g) * 0.2
    wp = _rand(g, cout, c, 1, 1, scale=1.0 / math.sqrt(c))
    bp = torch.randn(cout, generator=g) * 0.3
    mid = bf16_round(F.silu(F.conv2d(x, wd, bd, 1, 1, 1, c)))
    ref = F.conv2d(mid, wp, bp)
    if act2:
        ref = F.silu(ref)
    m = Mini(2, gpu_device)
    xin, out = m.buf(h, w, c), m.buf(h, w, cout)
    m.fill(xin, x)
    m.op(kind=OP_DWCONV3, name="t", act=1, inp=xin, out=out, weight=np.ascontiguousarray(wd.numpy()),
         bias=np.ascontiguousarray(bd.numpy()), post_weight=np.ascontiguousarray(wp.numpy().reshape(cout, c)),
         post_bias=np.ascontiguousarray(bp.numpy()), post_act=int(act2))
    m.run()
    got = m.read(out)
    tol = 2.0 ** -7 * ref.abs() + 2.0 ** -7 * F.conv2d(mid.abs(), wp.abs()) + 1e-3
    assert bool(((got - ref).abs() <= tol).all()), float(((got - ref).abs() - tol).max())


@pytest.mark.parametrize("c,cout,h,w", [(256, 80, 20, 20), (128, 80, 13, 9), (256, 64, 10, 10), (160, 72, 7, 5)])
def test_depthwise_pointwise_deep_launch(gpu_device, monkeypatch, c, cout, h, w):
    """Deep depthwise + pointwise launches on small maps (Detect level 2: 256 channels @20x20): dwpw_deep_kernel -- one
    16-pixel patch per workgroup, the four waves split the channel blocks of the depthwise conv and the channel-tile pairs of
    the GEMM -- against torch and, bit for bit, against dwpw_kernel (YL_DWPW_DEEP=0)."""
    g = torch.Generator().manual_seed(c * 3 + cout)
    x = _rand(g, 2, c, h, w)
    wd = torch.randn(c, 1, 3, 3, generator=g) * 0.3
    bd = torch.randn(c, generator=g) * 0.2
    wp = _rand(g, cout, c, 1, 1, scale=1.0 / math.sqrt(c))
    bp = torch.randn(cout, generator=g) * 0.3
    mid = bf16_round(F.silu(F.conv2d(x, wd, bd, 1, 1, 1, c)))
    ref = F.silu(F.conv2d(mid, wp, bp))
    got = {}
    for deep in ("1", "0"):
        monkeypatch.setenv("YL_DWPW_DEEP", deep)
        m = Mini(2, gpu_device)
        xin, out = m.buf(h, w, c), m.buf(h, w, cout)
        m.fill(xin, x)
        m.op(kind=OP_DWCONV3, name="t", act=1, inp=xin, out=out, weight=np.ascontiguousarray(wd.numpy()),
             bias=np.ascontiguousarray(bd.numpy()), post_weight=np.ascontiguousarray(wp.numpy().reshape(cout, c)),
             post_bias=np.ascontiguousarray(bp.numpy()), post_act=1)
        m.run()
        got[deep] = m.read(out)
    tol = 2.0 ** -7 * ref.abs() + 2.0 ** -7 * F.conv2d(mid.abs(), wp.abs()) + 1e-3
    assert bool(((got["1"] - ref).abs() <= tol).all()), float(((got["1"] - ref).abs() - tol).max())
    assert torch.equal(got["1"], got["0"])


def test_conv_stem_cin3(gpu_device):
    g = torch.Generator().manual_seed(8)
    x3 = bf16_round(torch.rand(2, 3, 64, 64, generator=g))
    x = torch.cat([x3, torch.zeros(2, 5, 64, 64)], 1)
    wt = _rand(g, 16, 3, 3, 3, scale=0.2)
    b = torch.randn(16, generator=g) * 0.1
    ref = F.silu(F.conv2d(x3, wt, b, 2, 1))
    got = run_conv(gpu_device, x, wt, b, 3, 2, True, None, cin_weight=3)
    _assert_close_bf16(got, ref, "stem")


@pytest.mark.parametrize("c,h,w,act", [(64, 20, 20, True), (80, 13, 9, True), (128, 10, 10, False)])
def test_dwconv_vs_torch(gpu_device, c, h, w, act):
    g = torch.Generator().manual_seed(c)
    x = _rand(g, 2, c, h, w)
    wt = torch.randn(c, 1, 3, 3, generator=g) * 0.3
    b = torch.randn(c, generator=g) * 0.2
    ref = F.conv2d(x, wt, b, 1, 1, 1, c)
    ref = F.silu(ref) if act else ref
    m = Mini(2, gpu_device)
    xin, out = m.buf(h, w, c), m.buf(h, w, c)
    m.fill(xin, x)
    m.op(kind=OP_DWCONV3, name="dw", k=3, stride=1, act=int(act), inp=xin, out=out, weight=wt.numpy(), bias=b.numpy())
    m.run()
    _assert_close_bf16(m.read(out), ref, "dwconv")


def test_dwconv_headmap_accumulate(gpu_device):
    """Attention.pe: depthwise conv on the v channels of the qkv map, added to the attention output."""
    g = torch.Generator().manual_seed(3)
    nh, kd, hd, h, w = 2, 32, 64, 6, 5
    qkv = _rand(g, 2, nh * (2 * kd + hd), h, w)
    prev = _rand(g, 2, nh * hd, h, w)
    wt = torch.randn(nh * hd, 1, 3, 3, generator=g) * 0.3
    b = torch.randn(nh * hd, generator=g) * 0.2
    v = qkv.view(2, nh, 2 * kd + hd, h, w)[:, :, 2 * kd:].reshape(2, nh * hd, h, w)
    ref = prev + F.conv2d(v, wt, b, 1, 1, 1, nh * hd)
    m = Mini(2, gpu_device)
    xin, out = m.buf(h, w, qkv.shape[1]), m.buf(h, w, nh * hd)
    m.fill(xin, qkv)
    m.fill(out, prev)
    m.op(kind=OP_DWCONV3, name="pe", k=3, stride=1, act=0, accumulate=1, grp=hd, grp_stride=2 * kd + hd,
         grp_off=2 * kd, inp=xin, out=out, weight=wt.numpy(), bias=b.numpy())
    m.run()
    _assert_close_bf16(m.read(out), ref, "pe")


@pytest.mark.parametrize("c,h,w", [(128, 20, 20), (64, 5, 7), (256, 12, 20)])
def test_sppf_pool_vs_torch(gpu_device, c, h, w):
    g = torch.Generator().manual_seed(c + h)
    x = _rand(g, 2, c, h, w)
    m = Mini(2, gpu_device)
    sbuf = m.buf(h, w, 4 * c)
    m.fill(sbuf.slice(0, c), x)
    m.op(kind=OP_SPPF_POOL, name="pool", inp=sbuf.slice(0, c), out=sbuf.slice(0, c))
    m.run()
    y1 = F.max_pool2d(x, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    np.testing.assert_array_equal(m.read(sbuf).numpy(), torch.cat([x, y1, y2, y3], 1).numpy())  # max is exact


def test_upsample_and_copy(gpu_device):
    g = torch.Generator().manual_seed(5)
    x = _rand(g, 2, 64, 10, 7)
    m = Mini(2, gpu_device)
    xin, cat = m.buf(10, 7, 64), m.buf(20, 14, 96)
    cp = m.buf(10, 7, 80)
    m.fill(xin, x)
    m.op(kind=OP_UPSAMPLE, name="up", stride=2, inp=xin, out=cat.slice(0, 64))
    m.op(kind=OP_UPSAMPLE, name="cp", stride=1, inp=xin, out=cp.slice(16, 64))
    m.run()
    np.testing.assert_array_equal(m.read(cat.slice(0, 64)).numpy(), F.interpolate(x, scale_factor=2.0).numpy())
    np.testing.assert_array_equal(m.read(cp.slice(16, 64)).numpy(), x.numpy())


@pytest.mark.parametrize("nh,h,w", [(2, 20, 20), (1, 7, 5), (4, 20, 12), (2, 40, 40), (1, 48, 40), (1, 80, 80), (1, 96, 96)])  # 1, 2, 4 and 8 query tiles per wave
def test_attention_vs_torch(gpu_device, nh, h, w):
    """reference nn/modules/block.py:905-912 on the per-head [q|k|v] channel layout."""
    g = torch.Generator().manual_seed(nh * 100 + h)
    kd, hd, B = 32, 64, 2
    N = h * w
    qkv = _rand(g, B, nh * (2 * kd + hd), h, w)
    q, k, v = qkv.view(B, nh, 2 * kd + hd, N).split([kd, kd, hd], dim=2)
    attn = ((q.transpose(-2, -1) @ k) * kd ** -0.5).softmax(-1)
    ref = (v @ attn.transpose(-2, -1)).view(B, nh * hd, h, w)
    m = Mini(B, gpu_device)
    xin, out = m.buf(h, w, qkv.shape[1]), m.buf(h, w, nh * hd)
    m.fill(xin, qkv)
    m.op(kind=OP_ATTENTION, name="attn", heads=nh, key_dim=kd, head_dim=hd, scale=float(kd ** -0.5), inp=xin, out=out)
    m.run()
    got = m.read(out)
    # The kernel rounds the probabilities to bf16 before the P.V MFMA (like any bf16 flash kernel):
    # each term p_j*v_j carries up to 2^-9 relative error, so the bound scales with sum_j p_j*|v_j|,
    # plus the bf16 rounding of the output itself.
    bound = (v.abs() @ attn.transpose(-2, -1)).view(B, nh * hd, h, w)
    err = (got - ref).abs()
    tol = 2.0 ** -8 * bound + 2.0 ** -8 * ref.abs() + 1e-3
    assert not (err > tol).any(), f"attention: max err {float(err.max()):.4g}, worst ratio {float((err / tol).max()):.2f}"


REG_CASES = [
    # Cin, Cout, H, W, act, res, NT     conv3_reg_kernel (filter in registers, input tile in LDS), forced onto small maps
    (64, 64, 20, 20, True, True, 4),
    (64, 64, 20, 20, True, False, 2),
    (64, 64, 7, 9, True, False, 2),
    (32, 32, 40, 40, True, True, 4),
    (64, 32, 13, 17, True, False, 2),
    (32, 64, 24, 40, False, False, 4),
    (64, 64, 80, 80, True, False, 4),
    (64, 64, 33, 150, True, True, 4),   # wide rows: more DMA pieces per wave than k-steps
]


@pytest.mark.parametrize("cin,cout,h,w,act,res,nt", REG_CASES)
def test_conv3_reg_vs_torch_and_direct(gpu_device, monkeypatch, cin, cout, h, w, act, res, nt):
    """The filter-in-registers 3x3 kernel against torch, and against the direct kernel: same k order, same MFMA
    accumulation chain starting from the bias -> bit-identical."""
    g = torch.Generator().manual_seed(cin * 100 + cout + h + nt + 7)
    B = 3
    x = _rand(g, B, cin, h, w)
    wt = _rand(g, cout, cin, 3, 3, scale=1.0 / math.sqrt(cin * 9))
    b = torch.randn(cout, generator=g) * 0.5
    ref = F.conv2d(x, wt, b, 1, 1)
    if act:
        ref = F.silu(ref)
    r = None
    if res:
        r = _rand(g, *ref.shape)
        ref = ref + r
    monkeypatch.setenv("YL_CONV_REG", "1")
    monkeypatch.setenv("YL_CONV_REG_MIN", "1")
    monkeypatch.setenv("YL_CONV_REG_NT", str(nt))
    got = run_conv(gpu_device, x, wt, b, 3, 1, act, r, expect_kernel="conv3x3_reg_mfma")
    _assert_close_bf16(got, ref, f"reg conv {cin}->{cout} NT={nt}")
    monkeypatch.setenv("YL_CONV_REG", "0")
    direct = run_conv(gpu_device, x, wt, b, 3, 1, act, r, expect_kernel="conv3x3_mfma")
    assert torch.equal(got, direct), f"reg vs direct kernel: max diff {float((got - direct).abs().max())}"


CS_CASES = [
    # H, W, act, res, B    conv3_cs_kernel (64 -> 64: output channels split between wave pairs), forced onto small maps
    (20, 20, True, False, 3),
    (20, 20, True, True, 3),
    (7, 9, True, False, 3),          # images shorter than a tile
    (80, 80, True, False, 3),        # the layer it is for (Detect box branch, level 0)
    (33, 150, False, True, 3),       # wide rows: a tile is less than one row; no activation
    (40, 40, True, False, 3),
    (80, 80, True, False, 32),       # BASELINE config 2's shape
    (40, 40, True, True, 20),        # a residual
    (48, 64, True, False, 61),       # a last tile cut short by the tensor's end
    (80, 80, True, False, 40),
    (79, 79, True, False, 8),        # 51 200 positions: the remainder round is a full 128-position tile -> conv3_cs_kernel (no short round)
]


@pytest.mark.parametrize("h,w,act,res,B", CS_CASES)
def test_conv3_cs_vs_torch_and_direct(gpu_device, monkeypatch, h, w, act, res, B):
    """The Cout-split filter-in-registers kernels (csrc/conv3_cs.hip: the tile walk with a balanced short first round,
    conv3_cs1b_kernel; conv3_cs_kernel where the positions fill whole rounds of 128-position tiles) against torch, and against
    conv_kernel: same operands, same k order and accumulation chain from the bias -> bit-identical."""
    g = torch.Generator().manual_seed(6400 + h + w)
    x = _rand(g, B, 64, h, w)
    wt = _rand(g, 64, 64, 3, 3, scale=1.0 / math.sqrt(64 * 9))
    b = torch.randn(64, generator=g) * 0.5
    ref = F.conv2d(x, wt, b, 1, 1)
    if act:
        ref = F.silu(ref)
    r = None
    if res:
        r = _rand(g, *ref.shape)
        ref = ref + r
    monkeypatch.setenv("YL_CONV_CS", "1")
    monkeypatch.setenv("YL_CONV_CS_MIN", "1")
    got = run_conv(gpu_device, x, wt, b, 3, 1, act, r, expect_kernel="conv3x3_cs_mfma")
    _assert_close_bf16(got, ref, f"cs conv {h}x{w}")
    monkeypatch.setenv("YL_CONV_CS", "0")
    monkeypatch.setenv("YL_CONV_REG", "0")
    direct = run_conv(gpu_device, x, wt, b, 3, 1, act, r, expect_kernel="conv3x3_mfma")
    assert torch.equal(got, direct), f"cs vs direct kernel: max diff {float((got - direct).abs().max())}"


def test_view_above_one_gib_fails_loudly(gpu_device):
    """Maximum sizes: buffer offsets are 32 bits with out-of-range sentinels, so a tensor slice above 1 GiB must be
    refused with an error (never wrapped around or silently truncated)."""
    m = Mini(1, gpu_device)
    H = W = 8200  # one plane of 8 channels: 8201 * 8201 * 16 B = 1.08 GB
    xin, out = m.buf(H, W, 8), m.buf(H, W, 16)
    m.op(kind=OP_CONV, name="big", k=1, stride=1, act=1, inp=xin, out=out, res=None,
         weight=np.zeros((16, 8, 1, 1), np.float32), bias=np.zeros(16, np.float32))
    with pytest.raises(RuntimeError, match="1 GiB"):
        m.run()


@pytest.mark.parametrize("B,H,W", [(3, 40, 40), (2, 24, 40), (2, 32, 32), (1, 38, 26), (2, 48, 80)])
def test_conv3_band_vs_conv_kernel(gpu_device, monkeypatch, B, H, W):
    """conv3_band_kernel (csrc/conv3_band.hip: a band of rows per workgroup, input in LDS, filter streamed from L2) on the pair
    Detect runs it on -- 128 -> 64 on an H x W map and 256 -> 64 on the H/2 x W/2 map (head.py:36-40) -- against the same two
    convs on conv_kernel: every bit; and against torch fp32."""
    gen = torch.Generator().manual_seed(H * 100 + W)
    shapes = [(128, H, W), (256, H // 2, W // 2)]
    xs = [bf16_round(torch.randn(B, c, h, w, generator=gen)) for c, h, w in shapes]
    ws = [bf16_round(torch.randn(64, c, 3, 3, generator=gen) / (c * 9) ** 0.5 * 1.5) for c, _, _ in shapes]
    bs = [torch.randn(64, generator=gen) * 0.3 for _ in shapes]
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("YL_CONV_BAND", mode)
        m = Mini(B, gpu_device)
        outs = []
        for (c, h, w), x, wt, b in zip(shapes, xs, ws, bs):
            xin, y = m.buf(h, w, c), m.buf(h, w, 64)
            m.fill(xin, x)
            m.op(kind=OP_CONV, name=f"c{c}", k=3, stride=1, act=1, inp=xin, out=y, group=1,
                 weight=np.ascontiguousarray(wt.numpy()), bias=np.ascontiguousarray(b.numpy()))
            outs.append(y)
        m.run()
        kern = m.eng.op_kernel(0)
        assert kern == ("conv3x3_band_mfma[x2]" if mode == "1" else "conv3x3_mfma[x2]"), kern
        got[mode] = [m.read(y) for y in outs]
    for a, b in zip(got["1"], got["0"]):
        assert torch.equal(a, b), float((a - b).abs().max())
    for y, x, wt, b in zip(got["1"], xs, ws, bs):
        ref = F.silu(F.conv2d(x, wt, b, padding=1))
        assert float((y - ref).abs().max()) <= 0.02 * float(ref.abs().max()) + 0.02


@pytest.mark.parametrize("cin,B,H,W", [(128, 2, 80, 80), (256, 4, 40, 40), (512, 11, 20, 20), (128, 3, 47, 91), (512, 16, 16, 16), (256, 2, 64, 48)])
def test_conv3_band_single_vs_conv_kernel(gpu_device, monkeypatch, cin, B, H, W):
    """conv3_band1_kernel (round 6): ONE deep-K 3x3 conv with 64 output channels per launch -- the first box convs of the wider
    models (YOLO11s: 128 -> 64 @80x80, 256 -> 64 @40x40, 512 -> 64 @20x20; head.py:36-40 cv2[i][0]) -- against conv_kernel: every
    bit; and against torch fp32.  Ragged maps, bands cut short by the image's last rows, one tile per wave quarter (512 @20x20)."""
    gen = torch.Generator().manual_seed(cin + H * 100 + W)
    x = bf16_round(torch.randn(B, cin, H, W, generator=gen))
    wt = bf16_round(torch.randn(64, cin, 3, 3, generator=gen) / (cin * 9) ** 0.5 * 1.5)
    b = torch.randn(64, generator=gen) * 0.3
    monkeypatch.setenv("YL_CONV_BAND", "1")
    got = run_conv(gpu_device, x, wt, b, 3, 1, True, None, expect_kernel="conv3x3_band_mfma")
    monkeypatch.setenv("YL_CONV_BAND", "0")
    direct = run_conv(gpu_device, x, wt, b, 3, 1, True, None, expect_kernel="conv3x3_mfma")
    assert torch.equal(got, direct), float((got - direct).abs().max())
    ref = F.silu(F.conv2d(x, wt, b, padding=1))
    assert float((got - ref).abs().max()) <= 0.02 * float(ref.abs().max()) + 0.02

