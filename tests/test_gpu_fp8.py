"""fp8 (OCP e4m3) conv path (csrc/conv_f8.hip; BASELINE.json config 5: e4m3 weights + activations, fp32 accumulate)
against plain torch fp32 on the SAME quantised operands: the products of e4m3 values are exact in fp32, so the only
differences are the summation order (relative 1e-5 of the accumulated magnitude) and the rounding of the stored output
(bf16: 2^-8 relative; fp8: the neighbouring e4m3 value at most, the large majority identical)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hip_util import F8_MAX, Mini, bf16_round, e4m3_round, quantize_filters_e4m3, run_conv_f8
from yololite_amd.nn.lower import BF16, FP8, OP_QUANT

pytestmark = pytest.mark.gpu


def _ref(xq, in_scale, w, b, k, s, act, res=None, res_scale=0.0):
    wq, sw = quantize_filters_e4m3(w)
    y = F.conv2d(xq.double(), wq.double(), None, s, k // 2).float()
    y = y * (in_scale * sw).view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    if act:
        y = F.silu(y)
    if res is not None:
        y = y + (res * res_scale if res_scale > 0 else res)
    return y


CASES = [
    # B, Cin, Cout, H, W, k, s, kernel
    (2, 128, 64, 20, 20, 3, 1, "conv3x3_f8_mfma"),    # a k-step inside one tap (Cin % 128 == 0), 4 channel tiles
    (2, 48, 48, 13, 17, 3, 1, "conv3x3_f8_mfma"),     # x-scale Bottleneck width: 3 tiles, steps straddle taps, ragged map
    (1, 96, 192, 24, 40, 1, 1, "conv1x1_f8_mfma"),    # 1x1, three channel blocks, K tail (96 = 6 groups of 16)
    (2, 64, 96, 32, 32, 3, 2, "conv3x3s2_f8_mfma"),   # stride 2
    (1, 384, 64, 16, 16, 3, 1, "conv3x3_f8_mfma"),    # 27 k-steps: the filter slab is restaged in chunks
    (3, 16, 32, 9, 9, 3, 1, "conv3x3_f8_mfma"),       # one k-group per tap, 2 tiles
    (1, 128, 192, 12, 20, 3, 1, "conv3x3_f8_mfma"),   # wide layer: six tiles per wave (two channel blocks of 96)
    (2, 256, 384, 16, 16, 3, 2, "conv3x3s2_f8_mfma"), # the same at stride 2, slab in chunks
]


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,s,kern", CASES)
@pytest.mark.parametrize("mode", ["bf16", "bf16+res", "fp8", "fp8+res8"])
def test_conv_f8_vs_torch(gpu_device, B, Cin, Cout, H, W, k, s, kern, mode):
    g = torch.Generator().manual_seed(Cin * 7 + Cout + H + k + s)
    in_scale = 0.037
    xq = e4m3_round(torch.randn(B, Cin, H, W, generator=g) * 30)   # e4m3 values, a few saturate
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.3
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    res, res_scale, out_scale = None, 0.0, 0.0
    if mode == "bf16+res":
        res = bf16_round(torch.randn(B, Cout, Ho, Wo, generator=g))
    if mode == "fp8+res8":
        res, res_scale = e4m3_round(torch.randn(B, Cout, Ho, Wo, generator=g) * 40), 0.02
    if mode.startswith("fp8"):
        out_scale = 0.05
    y = run_conv_f8(gpu_device, xq, in_scale, w, b, k, s, True, res, res_scale, out_scale, expect_kernel=kern)
    ref = _ref(xq, in_scale, w, b, k, s, True, res, res_scale)
    if out_scale > 0:
        want = e4m3_round(ref / out_scale)
        same = (y == want).float().mean().item()
        assert same > 0.99, same                                  # ties at rounding boundaries may fall either way
        mag = torch.maximum(y.abs(), want.abs()).clamp_min(2.0 ** -6)
        step = torch.exp2(torch.floor(torch.log2(mag)) - 3)          # e4m3 spacing at that magnitude
        assert bool(((y - want).abs() <= step).all())               # never further than the neighbouring value
    else:
        err = (y - ref).abs()
        assert bool((err <= 2 ** -7 * ref.abs() + 1e-3).all()), err.max()
        assert (y == bf16_round(ref)).float().mean().item() > 0.97


F8T_CASES = [
    # C, B, H, W     conv3_f8t_kernel (48 -> 48 / 96 -> 96 3x3: input tile in LDS, filter in registers), forced onto small maps too
    (48, 3, 40, 40),
    (48, 2, 33, 150),     # wide rows: a tile is less than two rows
    (48, 5, 7, 9),        # images much smaller than a tile
    (48, 6, 160, 160),    # more than one round of 512 workgroups: full tiles behind the short first round
    (96, 3, 40, 40),      # eight waves: two channel blocks x four columns
    (96, 2, 21, 150),
    (96, 4, 5, 6),
    (96, 2, 160, 160),    # more than one round of 256 workgroups
]


@pytest.mark.parametrize("C,B,H,W", F8T_CASES)
@pytest.mark.parametrize("mode", ["bf16", "bf16+res", "fp8", "fp8+res8"])
@pytest.mark.parametrize("kord", ["1", "0"])
def test_conv3_f8_tile_kernel_vs_torch_and_direct(gpu_device, monkeypatch, C, B, H, W, mode, kord):
    """csrc/conv3_f8t.hip against torch on the same quantised operands (tolerances of test_conv_f8_vs_torch) and against
    conv_f8_kernel: same packed filters, same k-steps in the same order (plane-major and tap-major) -> bit-identical."""
    g = torch.Generator().manual_seed(B * 1000 + H + W + C)
    Cin = Cout = C
    in_scale = 0.037
    xq = e4m3_round(torch.randn(B, Cin, H, W, generator=g) * 30)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.3
    res, res_scale, out_scale = None, 0.0, 0.0
    if mode == "bf16+res":
        res = bf16_round(torch.randn(B, Cout, H, W, generator=g))
    if mode == "fp8+res8":
        res, res_scale = e4m3_round(torch.randn(B, Cout, H, W, generator=g) * 40), 0.02
    if mode.startswith("fp8"):
        out_scale = 0.05
    monkeypatch.setenv("YL_F8_KORD", "1" if kord == "1" else "0")   # plane-major K order from 1 pixel per image up / never
    monkeypatch.setenv("YL_F8_TILE_MIN", "1")
    monkeypatch.setenv("YL_F8_TILE", "2")   # (the default: 48- and 96-channel forms)
    y = run_conv_f8(gpu_device, xq, in_scale, w, b, 3, 1, True, res, res_scale, out_scale, expect_kernel="conv3x3_f8_tile_mfma")
    ref = _ref(xq, in_scale, w, b, 3, 1, True, res, res_scale)
    if out_scale > 0:
        want = e4m3_round(ref / out_scale)
        assert (y == want).float().mean().item() > 0.99
        mag = torch.maximum(y.abs(), want.abs()).clamp_min(2.0 ** -6)
        step = torch.exp2(torch.floor(torch.log2(mag)) - 3)
        assert bool(((y - want).abs() <= step).all())
    else:
        err = (y - ref).abs()
        assert bool((err <= 2 ** -7 * ref.abs() + 1e-3).all()), err.max()
    if C == 96:  # the eight-wave form (one workgroup per CU) beside the default pair of four-wave workgroups
        monkeypatch.setenv("YL_F8_TILE_W8", "1")
        y8 = run_conv_f8(gpu_device, xq, in_scale, w, b, 3, 1, True, res, res_scale, out_scale, expect_kernel="conv3x3_f8_tile_mfma")
        assert torch.equal(y, y8), f"four-wave vs eight-wave tile kernel: max diff {float((y - y8).abs().max())}"
    monkeypatch.setenv("YL_F8_TILE", "0")
    direct = run_conv_f8(gpu_device, xq, in_scale, w, b, 3, 1, True, res, res_scale, out_scale, expect_kernel="conv3x3_f8_mfma")
    assert torch.equal(y, direct), f"tile kernel vs conv_f8_kernel: max diff {float((y - direct).abs().max())}"


def test_conv_f8_no_act_and_identity_filter(gpu_device):
    """A = I with an asymmetric B: a 1x1 conv whose filter is the identity copies its input channel for channel
    (catches a swapped row / column or a permuted k order that random data could hide behind tolerances)."""
    C = 64
    xq = e4m3_round(torch.arange(2 * C * 6 * 10, dtype=torch.float32).reshape(2, C, 6, 10) % 97 - 40)
    w = torch.eye(C).view(C, C, 1, 1) * 3.0
    y = run_conv_f8(gpu_device, xq, 1.0, w, torch.zeros(C), 1, 1, False)
    assert torch.equal(y, bf16_round(xq * 3.0))


def test_quantize_op(gpu_device):
    B, C, H, W = 2, 48, 11, 14
    x = bf16_round(torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(1)) * 20)
    x[0, 0, 0, :4] = torch.tensor([1e4, -1e4, 2 ** -12, 0.0])    # saturation, flush to zero
    m = Mini(B, gpu_device)
    a = m.buf(H, W, C, BF16)
    q = m.buf(H, W, C, FP8)
    m.fill(a, x)
    m.op(kind=OP_QUANT, name="q", inp=a, out=q, out_scale=0.11)
    m.run()
    assert m.eng.op_kernel(0) == "quantize_f8"
    assert torch.equal(m.read(q), e4m3_round(x * (torch.tensor(1.0) / torch.tensor(0.11))))


@pytest.mark.parametrize("scale,h,w,mode", [("n", 96, 128, "3x3"), ("n", 96, 128, "passes"), ("s", 64, 64, "3x3"), ("x", 64, 96, "3x3"),
                                            ("n", 96, 128, "mirror"), ("x", 64, 96, "mirror")])
def test_fp8_network_close_to_bf16(gpu_device, monkeypatch, scale, h, w, mode):
    """Whole network with the dense 3x3 convs in e4m3 (activation scales calibrated on the input itself) against the
    bf16 program of the same weights: level maps within the stated relative L2 error, graph replay bit-identical to
    eager, guards untouched."""
    from yololite_amd.nn import arch
    from yololite_amd.nn.engine import HipEngine
    from yololite_amd.nn.synth import synth_state_dict

    specs, _ = arch.parse_model(arch.yolo11_dict(scale))
    sd = synth_state_dict(scale, 3, cls_bias=[-2.0, -2.0, -2.0], cls_gain=4.0)
    x = torch.randint(0, 256, (2, 3, h, w), generator=torch.Generator().manual_seed(5), dtype=torch.uint8).to(gpu_device)
    e16 = HipEngine.from_model(specs, sd, 2, h, w, gpu_device, keep_logits=True)
    e16.forward(x)
    l16 = [t.clone() for t in e16.raw_levels()]
    # "3x3": the default (dense 3x3 convs, e4m3 copies from the producers' epilogues); "passes": separate quantize passes
    # instead; "mirror": opt-in, 1x1 convs too, tensors stored as e4m3 (faster, outside the stated detection tolerance)
    monkeypatch.setenv("YL_FP8_DUAL", "0" if mode == "passes" else "1")
    monkeypatch.setenv("YL_FP8_MIRROR", "1" if mode == "mirror" else "0")
    monkeypatch.setenv("YL_FP8_1X1_GF", "0")  # mirror mode here: every eligible 1x1 conv
    e8 = HipEngine.from_model(specs, sd, 2, h, w, gpu_device, fp8_calib=x, keep_logits=True)
    kernels = [e8.op_kernel(i) for i in range(len(e8.prog.ops))]
    assert sum("_f8_" in k for k in kernels) >= (35 if mode == "mirror" else 15)
    assert ("quantize_f8" in kernels) == (mode == "passes" or scale != "x")
    assert any("out8" in o for o in e8.prog.ops) == (mode != "passes")
    assert any("conv1x1_f8" in k for k in kernels) == (mode == "mirror")
    y8 = e8.forward(x).clone()
    l8 = [t.clone() for t in e8.raw_levels()]
    errs = [float((a - b).norm() / b.norm()) for a, b in zip(l8, l16)]
    print("fp8 vs bf16 level rel-L2:", scale, errs)
    assert max(errs) < (0.08 if mode == "mirror" else 0.04), errs
    assert torch.equal(e8.forward(x, graph=True), y8)
    for o in e8.prog.ops:
        if o["out"].dtype == FP8:
            assert e8.guards_are_zero(o["out"])


def test_copy_only_producer_is_bit_identical(gpu_device, monkeypatch):
    """yl_op.out8_only on a plain conv (YOLO11x: model.2.cv2, whose only reader is the e4m3 conv model.3): the bf16 store is
    dropped (a zero-byte descriptor), the e4m3 copy and everything downstream are the same bits as with the store; the bf16
    buffer the store would have hit keeps its old contents."""
    from yololite_amd.nn import arch
    from yololite_amd.nn.engine import HipEngine
    from yololite_amd.nn.synth import synth_state_dict

    specs, _ = arch.parse_model(arch.yolo11_dict("x"))
    sd = synth_state_dict("x", 3, cls_bias=[-2.0, -2.0, -2.0], cls_gain=4.0)
    x = torch.randint(0, 256, (2, 3, 64, 96), generator=torch.Generator().manual_seed(5), dtype=torch.uint8).to(gpu_device)
    outs = []
    for only in ("0", "1"):
        monkeypatch.setenv("YL_FP8_OUT8_ONLY", only)
        e8 = HipEngine.from_model(specs, sd, 2, 64, 96, gpu_device, fp8_calib=x, keep_logits=True)
        flagged = [o["name"] for o in e8.prog.ops if o.get("out8_only")]
        assert flagged == (["model.0", "model.2.cv2"] if only == "1" else ["model.0"]), flagged
        y = e8.forward(x).clone()
        assert torch.equal(e8.forward(x, graph=True), y)
        outs.append((y, [t.clone() for t in e8.raw_levels()]))
    assert torch.equal(outs[0][0], outs[1][0])
    assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))


FP8_SCORE_ATOL = 0.12  # stated tolerance of the e4m3 mode: 1.5 x the bf16 band (measured max |dscore| below)


def test_fp8_end_to_end_640_vs_reference_golden(gpu_device, golden):
    """Post-NMS detections of the e4m3 program at 640x640 against the REFERENCE's fp32 detections (golden_v1.npz):
    same gate as the bf16 path (one-to-one IoU >= 0.9, matched classes bit-exact, nothing unexplained) with the score
    band widened to the stated fp8 tolerance."""
    from oracle.parity_gate import parity_gate
    from yololite_amd.nn import arch
    from yololite_amd.nn.engine import HipEngine
    from yololite_amd.nn.synth import synth_state_dict
    from yololite_amd.utils import ops as hops

    sd = synth_state_dict("n", 0, cls_bias=list(golden["net_n_640_bias"]), cls_gain=8.0)
    x = torch.rand(1, 3, 640, 640, generator=torch.Generator().manual_seed(300)).to(gpu_device)
    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    eng = HipEngine.from_model(specs, sd, 1, 640, 640, gpu_device, fp8_calib=x)
    assert sum("_f8_" in eng.op_kernel(i) for i in range(len(eng.prog.ops))) >= 15
    y = eng.forward(x)
    dscore = float((y[:, 4:, ::37].cpu() - torch.from_numpy(golden["net_n_640_ysub"][:, 4:])).abs().max())
    print("fp8 e2e 640: max |dscore| on the golden subsample", dscore)
    assert dscore <= FP8_SCORE_ATOL
    det = hops.non_max_suppression(y, 0.25, 0.7)[0].cpu()
    ref = torch.from_numpy(golden["net_n_640_det0"])
    stats = parity_gate([ref], [det], 0.25, 0.7, band=FP8_SCORE_ATOL, min_matched_frac=0.75, verbose=True,
                        max_explained_share=0.6)  # (the e4m3 mode's stated cap; bf16 gates use 0.2)
    print("fp8 e2e 640 match:", stats)
    assert stats["class_mismatch"] == 0, "class indices of matched boxes must be bit-exact"
    assert stats["ok"], stats


@pytest.mark.parametrize("producer", ["bf16", "fp8"])
def test_out8_copy_from_the_producers_epilogue(gpu_device, producer):
    """yl_op.out8: a conv's epilogue also stores e4m3(y / out8_scale) of a channel range of its bf16 output -- the copy an
    fp8 conv downstream reads, instead of a quantize pass over the tensor.  Compared with quantising the stored bf16
    tensor (identical except where the bf16 rounding moved a value across an e4m3 rounding boundary)."""
    from hip_util import OP_CONV

    B, Cin, Cout, H, W, c0, n8, sc8 = 2, 64, 96, 12, 20, 32, 48, 0.03
    g = torch.Generator().manual_seed(9)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.3
    res = bf16_round(torch.randn(B, Cout, H, W, generator=g))
    m = Mini(B, gpu_device)
    if producer == "bf16":
        x = bf16_round(torch.randn(B, Cin, H, W, generator=g))
        xin = m.buf(H, W, Cin, BF16)
    else:
        x = e4m3_round(torch.randn(B, Cin, H, W, generator=g) * 30)
        xin = m.buf(H, W, Cin, FP8)
    m.fill(xin, x)
    y = m.buf(H, W, Cout, BF16)
    y8 = m.buf(H, W, n8, FP8)
    r = m.buf(H, W, Cout, BF16)
    m.fill(r, res)
    m.op(kind=OP_CONV, name="p", k=3, stride=1, act=1, inp=xin, out=y, res=r, in_scale=0.05, out8=y8, out8_c0=c0, out8_scale=sc8,
         weight=np.ascontiguousarray(w.numpy()), bias=np.ascontiguousarray(b.numpy()))
    m.run()
    assert m.eng.op_kernel(0) == ("conv3x3_mfma" if producer == "bf16" else "conv3x3_f8_mfma")
    yb, q = m.read(y), m.read(y8)
    want = e4m3_round(yb[:, c0: c0 + n8] * (torch.tensor(1.0) / torch.tensor(sc8)))
    assert (q == want).float().mean().item() > 0.97
    mag = torch.maximum(q.abs(), want.abs()).clamp_min(2.0 ** -6)
    assert bool(((q - want).abs() <= torch.exp2(torch.floor(torch.log2(mag)) - 3)).all())
    # and the bf16 output is what it is without the copy
    m2 = Mini(B, gpu_device)
    xin2 = m2.buf(H, W, Cin, xin.dtype)
    m2.fill(xin2, x)
    y2 = m2.buf(H, W, Cout, BF16)
    r2 = m2.buf(H, W, Cout, BF16)
    m2.fill(r2, res)
    m2.op(kind=OP_CONV, name="p", k=3, stride=1, act=1, inp=xin2, out=y2, res=r2, in_scale=0.05,
          weight=np.ascontiguousarray(w.numpy()), bias=np.ascontiguousarray(b.numpy()))
    m2.run()
    assert torch.equal(m2.read(y2), yb)


def test_out8_copy_from_the_gemm_kernel_is_conv_kernels(gpu_device, monkeypatch):
    """conv_gemm_kernel writes the e4m3 copy too (wide 1x1 convs in front of fp8 3x3 convs): bf16 output and copy
    bit-identical to conv_kernel's on the same operands."""
    from hip_util import OP_CONV

    B, Cin, Cout, H, W, c0, n8, sc8 = 3, 256, 384, 12, 20, 192, 192, 0.03
    g = torch.Generator().manual_seed(11)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g) * 0.3
    x = bf16_round(torch.randn(B, Cin, H, W, generator=g))
    got = {}
    for mode, kern in (("1", "conv1x1_gemm_mfma"), ("0", "conv1x1_mfma")):
        monkeypatch.setenv("YL_CONV_GEMM", mode)
        monkeypatch.setenv("YL_CONV_GEMM_BLOCKS", "1")
        m = Mini(B, gpu_device)
        xin = m.buf(H, W, Cin, BF16)
        m.fill(xin, x)
        y = m.buf(H, W, Cout, BF16)
        y8 = m.buf(H, W, n8, FP8)
        m.op(kind=OP_CONV, name="p", k=1, stride=1, act=1, inp=xin, out=y, in_scale=0.05, out8=y8, out8_c0=c0, out8_scale=sc8,
             weight=np.ascontiguousarray(w.numpy()), bias=np.ascontiguousarray(b.numpy()))
        m.run()
        assert m.eng.op_kernel(0) == kern, m.eng.op_kernel(0)
        got[mode] = (m.read(y), m.read(y8))
    assert torch.equal(got["1"][0], got["0"][0]) and torch.equal(got["1"][1], got["0"][1])
    want = e4m3_round(got["1"][0][:, c0: c0 + n8] * (torch.tensor(1.0) / torch.tensor(sc8)))
    assert (got["1"][1] == want).float().mean().item() > 0.97


def test_predict_api_in_fp8_mode(gpu_device, monkeypatch):
    """YL_DTYPE=fp8: YOLOLite.predict builds the e4m3 engine, calibrated on the first batch, behind the unchanged API."""
    from yololite_amd import YOLOLite

    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(2))
    y16 = YOLOLite("yolo11n.yaml")
    r16 = y16.predict(x, device=0, conf=0.001, verbose=False, save=False)
    monkeypatch.setenv("YL_DTYPE", "fp8")
    y8 = YOLOLite("yolo11n.yaml")
    r8 = y8.predict(x, device=0, conf=0.001, verbose=False, save=False)
    eng = next(iter(y8.predictor.model._engines.values()))
    assert any("_f8_" in eng.op_kernel(i) for i in range(len(eng.prog.ops)))
    assert len(r8) == len(r16) == 2
    for a, b in zip(r8, r16):
        assert abs(len(a.boxes) - len(b.boxes)) <= max(10, len(b.boxes) // 3)


def test_fp8_full_size_x1280_batch8_properties(gpu_device):
    """BASELINE.json config 5 at its full size (YOLO11x, 1280x1280, batch 8, e4m3 3x3 convs): permuting the images of the
    batch permutes the records bit for bit, the one-graph hot path equals the eager pass, and the e4m3 program's level maps
    stay within the stated distance of the bf16 program's."""
    from yololite_amd.nn import arch
    from yololite_amd.nn.engine import HipEngine
    from yololite_amd.nn.synth import synth_state_dict

    B, H, W = 8, 1280, 1280
    specs, _ = arch.parse_model(arch.yolo11_dict("x"))
    sd = synth_state_dict("x", 0, cls_bias=[-3.0, -3.0, -3.0], cls_gain=4.0)
    g = torch.Generator().manual_seed(8)
    x = torch.randint(0, 256, (B, 3, H, W), generator=g, dtype=torch.uint8).to(gpu_device)
    e16 = HipEngine.from_model(specs, sd, B, H, W, gpu_device, keep_logits=True)
    e16.forward(x, decode=False)
    l16 = [t.clone() for t in e16.raw_levels()]
    del e16
    torch.cuda.empty_cache()
    eng = HipEngine.from_model(specs, sd, B, H, W, gpu_device, fp8_calib=x, keep_logits=True)
    assert sum("_f8_" in eng.op_kernel(i) for i in range(len(eng.prog.ops))) >= 70
    out, cnt = (t.clone() for t in eng.detect(x, 0.25, 0.7, graph=False))
    errs = [float((a - b).norm() / b.norm()) for a, b in zip(eng.raw_levels(), l16)]
    print("fp8 x 1280 level rel-L2 vs bf16:", errs, "detections per image:", cnt.tolist())
    assert max(errs) < 0.05, errs  # measured 2.4-2.9 % on this input (0.7-0.9 % on the bench's)
    perm = torch.randperm(B, generator=g).to(gpu_device)
    out_p, cnt_p = (t.clone() for t in eng.detect(x[perm].contiguous(), 0.25, 0.7, graph=False))
    assert torch.equal(cnt_p, cnt[perm])
    for i in range(B):
        assert torch.equal(out_p[i, : int(cnt_p[i])], out[perm[i], : int(cnt_p[i])])
    out_g, cnt_g = eng.detect(x, 0.25, 0.7, graph=True)
    assert torch.equal(cnt_g, cnt) and all(torch.equal(out_g[i, : int(cnt[i])], out[i, : int(cnt[i])]) for i in range(B))


def test_fp8_x_1280_vs_fp32_oracle(gpu_device):
    """BASELINE.json config 5's OWN model -- YOLO11x at 1280x1280, e4m3 dense 3x3 convs -- against the fp32 oracle (reference
    arithmetic: nn/modules/block.py:720-739, head.py:95-126, utils/ops.py:138-273) through the FIXED post-NMS gate of bench.py
    (C5_GATE): boxes paired one to one by IoU >= 0.9, classes of paired boxes bit-exact, nothing unexplained, >= 90 % of the
    reference's boxes paired, one-sided share <= 0.35, |det / ref - 1| <= 0.25, |dconf| p99 <= 0.2 and max <= 0.6 (score bands
    stated from the HIP measurement on this head, see bench.py).

    The head carries PLANTED OBJECTS (nn/synth.py plant_object_head, ~96 per image): the per-anchor best score is bimodal, so
    the gate is falsifiable -- a kernel that loses or shifts features loses objects or moves their scores.  Round 5's version of
    this test ran the calibrated random head, whose best scores all sit at the threshold: e4m3 kept 2.2 x the reference's boxes
    there and only a yardstick against this repo's own CPU emulation could pass (review, weak #1).  The emulation is still
    run and printed -- as a diagnostic, not as the criterion."""
    import bench
    from oracle import yolo11_oracle as oracle
    from oracle.parity_gate import parity_gate

    B, S, conf, iou = 2, 1280, 0.25, 0.7
    x = bench.config5_images(B, S).to(gpu_device)
    x8 = (x * 255).to(torch.uint8)
    specs, sd, eng = bench.build_model("x", gpu_device, B, S, S, x, fp8=True, planted=bench.c5_planted(B))
    assert sum("_f8_" in eng.op_kernel(i) for i in range(len(eng.prog.ops))) >= 70
    o, c = eng.detect(x8, conf, iou, max_det=300, graph=False)
    y8 = eng.forward(x8, graph=False).float().cpu()
    torch.cuda.synchronize()
    counts = c.tolist()
    assert max(counts) < 300, f"an image is cut at max_det: the gate would compare truncated records ({counts})"
    torch.set_num_threads(8)
    xin = x8.float().cpu() / 255.0
    with torch.inference_mode():
        y32, _ = oracle.forward(sd, xin, "x")
        ref = oracle.non_max_suppression(y32, conf, iou, max_det=300)
        y8e, _ = oracle.forward(sd, xin, "x", storage="bf16", fp8_scales=eng.fp8_conv_scales())
        det8e = oracle.non_max_suppression(y8e, conf, iou, max_det=300)
    # the head IS bimodal on the reference side: few anchors between 0.05 and 0.6, ~96 objects per image above the threshold
    mx = y32[:, 4:].amax(1)
    mid = int(((mx > 0.05) & (mx < 0.6)).sum())
    over = (mx > conf).sum(1).tolist()
    print("fp32 reference: anchors with a best score in (0.05, 0.6):", mid, "of", mx.numel(), "; over the threshold per image:", over)
    assert mid < 0.01 * mx.numel() and all(60 <= n <= 200 for n in over), (mid, over)
    g = parity_gate(ref, [o[i, : counts[i]].cpu() for i in range(B)], conf, iou, ref_y=y32, got_y=y8, **bench.C5_GATE)
    fmt = parity_gate(ref, det8e, conf, iou, ref_y=y32, got_y=y8e, **bench.C5_GATE)
    keys = ("ref", "det", "matched", "matched_frac", "class_mismatch", "unexplained", "max_dconf", "p99_dconf", "explained_share",
            "count_bias", "near_threshold", "nms_flip", "ok")
    print("fp8 x 1280, HIP e4m3 vs fp32 oracle:                ", {k: g.get(k) for k in keys})
    print("fp8 x 1280, e4m3 emulation vs fp32 (diagnostic):    ", {k: fmt.get(k) for k in keys})
    assert g["ok"], g
    # the same program without e4m3 (bf16 storage) through the same gate with the bf16 score bands: tighter, same head
    from yololite_amd.nn.engine import HipEngine

    e16 = HipEngine.from_model(specs, sd, B, S, S, gpu_device)
    o16, c16 = e16.detect(x8, conf, iou, max_det=300, graph=False)
    y16 = e16.forward(x8, graph=False).float().cpu()
    g16 = parity_gate(ref, [o16[i, : int(c16[i])].cpu() for i in range(B)], conf, iou, ref_y=y32, got_y=y16,
                      **dict(bench.C5_GATE, band=0.1, max_band=0.25, p99_band=0.1))
    print("bf16 x 1280, HIP bf16 vs fp32 oracle:               ", {k: g16.get(k) for k in keys})
    assert g16["ok"], g16


FP8_GAIN1_SCORE_ATOL = 0.02  # SURVEY 8(d)'s own band: measured 0.0056 (e4m3 dense 3x3 convs, YOLO11x) against the reference at a trained head's logit scale


def test_fp8_pre_nms_at_realistic_logit_scale_vs_reference_golden(gpu_device):
    """The e4m3 program against the REFERENCE's own pre-NMS tensor (tests/golden/golden_gain1_v2.npz, x_192x256: YOLO11x, class
    logits at cls_gain = 1): the score band of the e4m3 mode is stated here from the HIP measurement, against reference-made numbers
    -- not against this repo's emulation.  The bf16 program on the same vector is inside SURVEY 8(d)'s 0.02
    (tests/test_gpu_net.py::test_pre_nms_at_realistic_logit_scale_vs_reference_goldens_v2)."""
    import numpy as np
    from pathlib import Path
    from yololite_amd.nn import arch
    from yololite_amd.nn.engine import HipEngine
    from yololite_amd.nn.synth import synth_state_dict

    G = np.load(Path(__file__).resolve().parent / "golden" / "golden_gain1_v2.npz")
    tag, scale = "x_192x256", "x"
    h, w, seed, sub = (int(v) for v in G[tag + "_meta"])
    sd = synth_state_dict(scale, seed, cls_bias=list(G[tag + "_bias"]), cls_gain=1.0)
    x8 = torch.from_numpy(G[tag + "_x8"]).to(gpu_device)
    ref = torch.from_numpy(G[tag + "_y"])
    specs, _ = arch.parse_model(arch.yolo11_dict(scale))
    eng = HipEngine.from_model(specs, sd, 1, h, w, gpu_device, fp8_calib=x8)
    assert sum("_f8_" in eng.op_kernel(i) for i in range(len(eng.prog.ops))) >= 70
    y = eng.forward(x8).cpu()
    d = (y[:, 4:] - ref[:, 4:]).abs()
    print(f"{tag}: HIP e4m3 vs reference golden: max |dscore| {float(d.max()):.4f}, rms {float(d.pow(2).mean().sqrt()):.5f}")
    assert float(d.max()) <= FP8_GAIN1_SCORE_ATOL, float(d.max())
