"""-m gpu: the per-image resident stage kernels (csrc/resident.hip) against the launches they replace (bit for bit) and
against plain torch fp32 of the same block (reference nn/modules/block.py:245-259, 330-343, 731-739)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hip_util import Mini, bf16_round
from yololite_amd.nn import arch
from yololite_amd.nn.engine import HipEngine
from yololite_amd.nn.lower import (OP_ATTENTION, OP_C3K, OP_CONV, OP_DWCONV3, OP_PSA, OP_SPPF, OP_SPPF_POOL, c3k_resident_fits,
                                   psa_resident_fits, sppf_resident_fits)
from yololite_amd.nn.synth import synth_state_dict

pytestmark = pytest.mark.gpu


def _rand_block(gen, c_=64):
    """BN-folded filters of a C3k block of inner width c_ (values bf16-representable, scaled so activations stay O(1))."""
    c = 2 * c_

    def wb(co, ci, k):
        w = bf16_round(torch.randn(co, ci, k, k, generator=gen) * (1.5 / (ci * k * k) ** 0.5))
        return w, torch.randn(co, generator=gen) * 0.2

    return dict(cv12=wb(c, c, 1), m=[wb(c_, c_, 3) for _ in range(4)], cv3=wb(c, c, 1))


def _torch_block(x, blk, storage_bf16=True):
    """fp32 arithmetic, every stored tensor rounded to bf16 where the HIP path stores bf16 (so only summation order differs)."""
    r = bf16_round if storage_bf16 else (lambda t: t)

    def conv(t, wb_, k):
        return r(F.silu(F.conv2d(t, wb_[0], wb_[1], padding=k // 2)))

    u = conv(x, blk["cv12"], 1)
    c_ = u.shape[1] // 2
    a, b = u[:, :c_], u[:, c_:]
    for j in range(2):
        t = conv(a, blk["m"][2 * j], 3)
        a = r(F.silu(F.conv2d(t, blk["m"][2 * j + 1][0], blk["m"][2 * j + 1][1], padding=1)) + a)
    return conv(torch.cat([a, b], 1), blk["cv3"], 1)


def _np(t):
    return np.ascontiguousarray(t.numpy(), dtype=np.float32)


def _run_resident(dev, x, blk):
    B, c, H, W = x.shape
    m = Mini(B, dev)
    tin = m.buf(H, W, 3 * c)          # the block reads a channel slice of a wider buffer, as in C3k2 (block.py:231-235)
    xin = tin.slice(c, c)
    yv = tin.slice(2 * c, c)
    m.fill(xin, x)
    m.op(kind=OP_C3K, name="c3k", act=1, inp=xin, out=yv,
         weight=_np(blk["cv12"][0].reshape(c, -1)), bias=_np(blk["cv12"][1]),
         extra_weight=[_np(w) for w, _ in blk["m"]] + [_np(blk["cv3"][0].reshape(c, -1))],
         extra_bias=[_np(b) for _, b in blk["m"]] + [_np(blk["cv3"][1])])
    m.run()
    assert m.eng.op_kernel(0) == "c3k_resident_mfma"
    return m.read(yv)


def _run_chain(dev, x, blk):
    """The same block as the six conv launches of the unfused program (nn/lower.py::c3k_block)."""
    B, c, H, W = x.shape
    c_ = c // 2
    m = Mini(B, dev)
    xin, u, t, yv = m.buf(H, W, c), m.buf(H, W, c), m.buf(H, W, c_), m.buf(H, W, c)
    m.fill(xin, x)
    a = u.slice(0, c_)
    m.op(kind=OP_CONV, name="cv12", k=1, stride=1, act=1, inp=xin, out=u, weight=_np(blk["cv12"][0]), bias=_np(blk["cv12"][1]))
    for j in range(2):
        m.op(kind=OP_CONV, name=f"m{j}.cv1", k=3, stride=1, act=1, inp=a, out=t, weight=_np(blk["m"][2 * j][0]), bias=_np(blk["m"][2 * j][1]))
        m.op(kind=OP_CONV, name=f"m{j}.cv2", k=3, stride=1, act=1, inp=t, out=a, res=a, weight=_np(blk["m"][2 * j + 1][0]),
             bias=_np(blk["m"][2 * j + 1][1]))
    m.op(kind=OP_CONV, name="cv3", k=1, stride=1, act=1, inp=u, out=yv, weight=_np(blk["cv3"][0]), bias=_np(blk["cv3"][1]))
    m.run()
    return m.read(yv)


@pytest.mark.parametrize("B,H,W", [(2, 20, 20), (3, 12, 20), (1, 10, 10), (2, 5, 7), (2, 16, 16), (1, 2, 3), (2, 12, 12), (2, 16, 20), (1, 19, 19)])
def test_c3k_resident_vs_chain_and_torch(gpu_device, B, H, W):
    """One launch per block == the six launches, every bit; and both within bf16 rounding of torch fp32.  The sizes cover
    every tile-count instantiation (1 .. 7 tiles per wave quarter), ragged last tiles, non-square maps and the largest map that fits."""
    assert c3k_resident_fits(64, H, W)
    gen = torch.Generator().manual_seed(1000 * H + W)
    blk = _rand_block(gen)
    x = bf16_round(torch.randn(B, 128, H, W, generator=gen))
    got = _run_resident(gpu_device, x, blk)
    chain = _run_chain(gpu_device, x, blk)
    assert torch.equal(got, chain), f"max |d| {float((got - chain).abs().max())}"
    ref = _torch_block(x, blk)
    # same operands, fp32 accumulation in another order: a few outputs land on the other side of a bf16 rounding boundary
    err = (got - ref).abs()
    assert float(err.max()) <= 0.05 * float(ref.abs().max()), float(err.max())
    assert float((err > 1e-6).float().mean()) < 0.2


@pytest.mark.parametrize("B,H,W", [(2, 20, 20), (3, 12, 20), (1, 10, 10), (2, 16, 16), (2, 16, 20), (1, 19, 19), (1, 11, 13)])
def test_c3k_resident_two_workgroups_per_image(gpu_device, monkeypatch, B, H, W):
    """YL_C3K_SPLIT=2: each workgroup runs the block on its half of the rows + a halo of four (the rows next to the cut are
    computed with zeros for their missing neighbours and never stored): the same bits as one workgroup per image.  Odd heights,
    windows that overlap by more than the halo."""
    gen = torch.Generator().manual_seed(1000 * H + W + 1)
    blk = _rand_block(gen)
    x = bf16_round(torch.randn(B, 128, H, W, generator=gen))
    monkeypatch.setenv("YL_C3K_SPLIT", "1")
    one = _run_resident(gpu_device, x, blk)
    monkeypatch.setenv("YL_C3K_SPLIT", "2")
    two = _run_resident(gpu_device, x, blk)
    assert torch.equal(one, two), f"max |d| {float((one - two).abs().max())}"


@pytest.mark.parametrize("h,w,b", [(640, 640, 8), (384, 640, 8), (320, 320, 9), (640, 640, 32)])
def test_network_with_resident_c3k_is_bit_identical(gpu_device, monkeypatch, h, w, b):
    """YOLO11n with layers 8 and 22 as ONE launch each against the program with their six launches: level maps and decoded
    predictions, every bit."""
    sd = synth_state_dict("n", 5, cls_bias=[-2.0, -2.0, -2.0], cls_gain=4.0)
    x = torch.randint(0, 256, (b, 3, h, w), generator=torch.Generator().manual_seed(h + 3 * w), dtype=torch.uint8).to(gpu_device)
    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    monkeypatch.setenv("YL_FUSE_C3K", "1")
    eng = HipEngine.from_model(specs, sd, b, h, w, gpu_device)
    names = [eng.op_kernel(i) for i in range(len(eng.prog.ops))]
    assert names.count("c3k_resident_mfma") == 2, names
    y = eng.forward(x).clone()
    lv = [t.clone() for t in eng.head_outputs()]
    monkeypatch.setenv("YL_FUSE_C3K", "0")
    ref = HipEngine.from_model(specs, sd, b, h, w, gpu_device)
    assert "c3k_resident_mfma" not in [ref.op_kernel(i) for i in range(len(ref.prog.ops))]
    assert len(ref.prog.ops) == len(eng.prog.ops) + 10
    assert torch.equal(y, ref.forward(x))
    for p, q in zip(lv, ref.head_outputs()):
        assert torch.equal(p, q)


# ------------------------------------------------------------------------------------------------ PSABlock (C2PSA)
def _rand_psa(gen, c=128):
    def wb(co, ci, k=1, gain=1.0, groups=1):
        w = bf16_round(torch.randn(co, ci // groups, k, k, generator=gen) * (gain / ((ci // groups) * k * k) ** 0.5))
        return w, torch.randn(co, generator=gen) * 0.2

    return dict(qkv=wb(2 * c, c, gain=1.5), pe=wb(c, c, 3, gain=1.0, groups=c), proj=wb(c, c, gain=0.8),
                ffn0=wb(2 * c, c, gain=1.4), ffn1=wb(c, 2 * c, gain=0.8))


def _psa_ops(m, bview, blk, H, W, fused, c=128, nh=2, kd=32, hd=64, u=None, u2=None):
    scale = float(kd ** -0.5)
    if fused:
        m.op(kind=OP_PSA, name="psa", act=0, heads=nh, key_dim=kd, head_dim=hd, scale=scale, inp=u, out=u2,
             weight=_np(blk["qkv"][0].reshape(2 * c, -1)), bias=_np(blk["qkv"][1]),
             extra_weight=[_np(blk["pe"][0]), _np(blk["proj"][0].reshape(c, -1)), _np(blk["ffn0"][0].reshape(2 * c, -1)),
                           _np(blk["ffn1"][0].reshape(c, -1))],
             extra_bias=[_np(blk["pe"][1]), _np(blk["proj"][1]), _np(blk["ffn0"][1]), _np(blk["ffn1"][1])])
        return
    qkv, o, f1 = m.buf(H, W, 2 * c), m.buf(H, W, c), m.buf(H, W, 2 * c)
    m.op(kind=OP_CONV, name="qkv", k=1, stride=1, act=0, inp=bview, out=qkv, weight=_np(blk["qkv"][0]), bias=_np(blk["qkv"][1]))
    m.op(kind=OP_ATTENTION, name="attn", heads=nh, key_dim=kd, head_dim=hd, scale=scale, inp=qkv, out=o)
    m.op(kind=OP_DWCONV3, name="pe", k=3, stride=1, act=0, accumulate=1, grp=hd, grp_stride=2 * kd + hd, grp_off=2 * kd, inp=qkv, out=o,
         weight=_np(blk["pe"][0]), bias=_np(blk["pe"][1]))
    m.op(kind=OP_CONV, name="proj", k=1, stride=1, act=0, inp=o, out=bview, res=bview, weight=_np(blk["proj"][0]), bias=_np(blk["proj"][1]))
    m.op(kind=OP_CONV, name="ffn0", k=1, stride=1, act=1, inp=bview, out=f1, weight=_np(blk["ffn0"][0]), bias=_np(blk["ffn0"][1]))
    m.op(kind=OP_CONV, name="ffn1", k=1, stride=1, act=0, inp=f1, out=bview, res=bview, weight=_np(blk["ffn1"][0]), bias=_np(blk["ffn1"][1]))


def _torch_psa(x, blk, nh=2, kd=32, hd=64):
    """fp32 PSABlock (block.py:919-953, 905-916)."""
    B, C, H, W = x.shape
    N = H * W
    qkv = F.conv2d(x, *blk["qkv"])
    q, k, v = qkv.view(B, nh, 2 * kd + hd, N).split([kd, kd, hd], dim=2)
    attn = ((q.transpose(-2, -1) @ k) * kd ** -0.5).softmax(-1)
    o = (v @ attn.transpose(-2, -1)).view(B, C, H, W) + F.conv2d(v.reshape(B, C, H, W), blk["pe"][0], blk["pe"][1], padding=1, groups=C)
    x = x + F.conv2d(o, *blk["proj"])
    return x + F.conv2d(F.silu(F.conv2d(x, *blk["ffn0"])), *blk["ffn1"])


@pytest.mark.parametrize("B,H,W", [(2, 20, 20), (3, 12, 20), (1, 10, 10), (2, 5, 7), (2, 16, 16), (1, 19, 19), (2, 3, 3)])
def test_psa_resident_vs_chain_and_torch(gpu_device, B, H, W):
    """The PSABlock in one launch == its six launches, every bit; both close to torch fp32.  Sizes: one to four query tiles per
    wave, ragged last tiles, key counts that are not multiples of 32, waves without any tile."""
    assert psa_resident_fits(128, 2, 32, 64, H, W)
    gen = torch.Generator().manual_seed(77 * H + W)
    blk = _rand_psa(gen)
    x = bf16_round(torch.randn(B, 128, H, W, generator=gen))
    xa = bf16_round(torch.randn(B, 128, H, W, generator=gen))
    got = {}
    for fused in (True, False):
        m = Mini(B, gpu_device)
        u, u2 = m.buf(H, W, 256), m.buf(H, W, 256)
        bv = u.slice(128, 128)  # the block works on the second half of C2PSA.cv1's output (block.py:1035-1037)
        m.fill(u.slice(0, 128), xa)
        m.fill(bv, x)
        _psa_ops(m, bv, blk, H, W, fused, u=u, u2=u2)
        m.run()
        if fused:  # in -> out: cat(a, b') in the second buffer, the input untouched
            assert m.eng.op_kernel(0) == "psa_resident_mfma"
            assert torch.equal(m.read(u2.slice(0, 128)), xa) and torch.equal(m.read(bv), x)
            got[fused] = m.read(u2.slice(128, 128))
        else:
            got[fused] = m.read(bv)
    assert torch.equal(got[True], got[False]), f"max |d| {float((got[True] - got[False]).abs().max())}"
    ref = _torch_psa(x, blk)
    err = (got[True] - ref).abs()
    assert float(err.max()) <= 0.06 * float(ref.abs().max()), (float(err.max()), float(ref.abs().max()))


@pytest.mark.parametrize("h,w,b", [(640, 640, 8), (384, 640, 8), (320, 320, 9)])
def test_network_with_resident_psa_is_bit_identical(gpu_device, monkeypatch, h, w, b):
    """YOLO11n with layer 10's PSABlock as ONE launch against the program with its six launches: level maps and decoded
    predictions, every bit."""
    sd = synth_state_dict("n", 6, cls_bias=[-2.0, -2.0, -2.0], cls_gain=4.0)
    x = torch.randint(0, 256, (b, 3, h, w), generator=torch.Generator().manual_seed(h + 5 * w), dtype=torch.uint8).to(gpu_device)
    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    monkeypatch.setenv("YL_FUSE_PSA", "1")
    eng = HipEngine.from_model(specs, sd, b, h, w, gpu_device)
    assert [eng.op_kernel(i) for i in range(len(eng.prog.ops))].count("psa_resident_mfma") == 1
    y = eng.forward(x).clone()
    lv = [t.clone() for t in eng.head_outputs()]
    monkeypatch.setenv("YL_FUSE_PSA", "0")
    monkeypatch.setenv("YL_FUSE_PSA_TAIL", "0")  # (the six launches, not the one-launch tail either)
    ref = HipEngine.from_model(specs, sd, b, h, w, gpu_device)
    assert "psa_resident_mfma" not in [ref.op_kernel(i) for i in range(len(ref.prog.ops))]
    assert len(ref.prog.ops) == len(eng.prog.ops) + 5
    assert torch.equal(y, ref.forward(x))
    for p, q in zip(lv, ref.head_outputs()):
        assert torch.equal(p, q)


# ------------------------------------------------------------------------------------------------ SPPF (+ C2PSA.cv1)
def _rand_sppf(gen):
    def wb(co, ci, gain):
        w = bf16_round(torch.randn(co, ci, 1, 1, generator=gen) * (gain / ci ** 0.5))
        return w, torch.randn(co, generator=gen) * 0.2

    return dict(cv1=wb(128, 256, 1.6), cv2=wb(256, 512, 1.6), tail=wb(256, 256, 1.6))


def _torch_sppf(x, blk, tail):
    """fp32 SPPF (block.py:165-184) [+ the 1x1 conv behind it], stored tensors rounded to bf16 like the HIP path's."""
    def conv(t, wb_):
        return bf16_round(F.silu(F.conv2d(t, *wb_)))

    y = [conv(x, blk["cv1"])]
    for _ in range(3):
        y.append(F.max_pool2d(y[-1], 5, 1, 2))
    s = conv(torch.cat(y, 1), blk["cv2"])
    return conv(s, blk["tail"]) if tail else s


@pytest.mark.parametrize("tail", [True, False])
@pytest.mark.parametrize("B,H,W", [(2, 20, 20), (3, 12, 20), (1, 10, 10), (2, 5, 7), (2, 16, 16), (1, 19, 19), (2, 3, 3), (1, 1, 1), (2, 21, 21), (1, 13, 32)])
def test_sppf_resident_vs_chain_and_torch(gpu_device, B, H, W, tail):
    """SPPF [+ the follow-up 1x1 conv] in one launch == its three [four] launches, every bit; both close to torch fp32.  Sizes: one
    to four cv1 tiles per wave, one and two workgroups per image, ragged last tiles, maps smaller than the 13x13 window, waves
    without any tile."""
    assert sppf_resident_fits(256, 256, 256 if tail else 0, H, W)
    gen = torch.Generator().manual_seed(31 * H + W)
    blk = _rand_sppf(gen)
    x = bf16_round(torch.randn(B, 256, H, W, generator=gen))
    got = {}
    for fused in (True, False):
        m = Mini(B, gpu_device)
        tin = m.buf(H, W, 512)  # the layer reads a channel slice of a wider buffer
        xin = tin.slice(256, 256)
        out = m.buf(H, W, 256)
        m.fill(xin, x)
        if fused:
            m.op(kind=OP_SPPF, name="sppf", act=1, inp=xin, out=out, weight=_np(blk["cv1"][0].reshape(128, -1)), bias=_np(blk["cv1"][1]),
                 extra_weight=[_np(blk["cv2"][0].reshape(256, -1))] + ([_np(blk["tail"][0].reshape(256, -1))] if tail else []),
                 extra_bias=[_np(blk["cv2"][1])] + ([_np(blk["tail"][1])] if tail else []))
        else:
            sbuf = m.buf(H, W, 512)
            mid = m.buf(H, W, 256) if tail else out
            m.op(kind=OP_CONV, name="cv1", k=1, stride=1, act=1, inp=xin, out=sbuf.slice(0, 128), weight=_np(blk["cv1"][0]), bias=_np(blk["cv1"][1]))
            m.op(kind=OP_SPPF_POOL, name="m", inp=sbuf.slice(0, 128), out=sbuf.slice(0, 128))
            m.op(kind=OP_CONV, name="cv2", k=1, stride=1, act=1, inp=sbuf, out=mid, weight=_np(blk["cv2"][0]), bias=_np(blk["cv2"][1]))
            if tail:
                m.op(kind=OP_CONV, name="tail", k=1, stride=1, act=1, inp=mid, out=out, weight=_np(blk["tail"][0]), bias=_np(blk["tail"][1]))
        m.run()
        if fused:
            assert m.eng.op_kernel(0) == "sppf_resident_mfma"
            assert torch.equal(m.read(xin), x)
        got[fused] = m.read(out)
    assert torch.equal(got[True], got[False]), f"max |d| {float((got[True] - got[False]).abs().max())}"
    ref = _torch_sppf(x, blk, tail)
    err = (got[True] - ref).abs()
    assert float(err.max()) <= 0.03 * max(1.0, float(ref.abs().max())), (float(err.max()), float(ref.abs().max()))


@pytest.mark.parametrize("h,w,b", [(640, 640, 8), (384, 640, 8), (320, 320, 9), (640, 640, 32)])
def test_network_with_resident_sppf_is_bit_identical(gpu_device, monkeypatch, h, w, b):
    """YOLO11n with layer 9 and layer 10's first conv as ONE launch against the program with the four launches: level maps and
    decoded predictions, every bit."""
    sd = synth_state_dict("n", 6, cls_bias=[-2.0, -2.0, -2.0], cls_gain=4.0)
    x = torch.randint(0, 256, (b, 3, h, w), generator=torch.Generator().manual_seed(h + 3 * w), dtype=torch.uint8).to(gpu_device)
    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    monkeypatch.setenv("YL_FUSE_SPPF", "1")
    eng = HipEngine.from_model(specs, sd, b, h, w, gpu_device)
    assert [eng.op_kernel(i) for i in range(len(eng.prog.ops))].count("sppf_resident_mfma") == 1
    y = eng.forward(x).clone()
    lv = [t.clone() for t in eng.head_outputs()]
    monkeypatch.setenv("YL_FUSE_SPPF", "0")
    ref = HipEngine.from_model(specs, sd, b, h, w, gpu_device)
    assert "sppf_resident_mfma" not in [ref.op_kernel(i) for i in range(len(ref.prog.ops))]
    assert len(ref.prog.ops) == len(eng.prog.ops) + 3
    assert torch.equal(y, ref.forward(x))
    for p, q in zip(lv, ref.head_outputs()):
        assert torch.equal(p, q)


@pytest.mark.parametrize("h,w,b,one_batch", [(640, 640, 32, True), (640, 640, 3, False), (384, 640, 5, True), (320, 320, 9, True), (96, 160, 2, False),
                                             (224, 352, 7, True), (32, 32, 1, False)])
def test_network_with_psa_tail_is_bit_identical(gpu_device, monkeypatch, h, w, b, one_batch):
    """YOLO11n with the pointwise tail of layer 10 (proj, ffn.0, ffn.1, C2PSA.cv2) as ONE launch over the pixels of the whole batch
    (csrc/psa_tail.hip: the latency profile, and batches too small for the one-launch PSABlock) against the four launches: level
    maps and decoded predictions, every bit; tiles that straddle rows and images, a last workgroup with idle waves."""
    sd = synth_state_dict("n", 12, cls_bias=[-2.0, -2.0, -2.0], cls_gain=4.0)
    x = torch.randint(0, 256, (b, 3, h, w), generator=torch.Generator().manual_seed(h + 5 * w + b), dtype=torch.uint8).to(gpu_device)
    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    monkeypatch.setenv("YL_FUSE_PSA_TAIL", "1")
    eng = HipEngine.from_model(specs, sd, b, h, w, gpu_device, one_batch=one_batch)
    names = [eng.op_kernel(i) for i in range(len(eng.prog.ops))]
    assert names.count("psa_tail_mfma") == 1 and "psa_resident_mfma" not in names
    y = eng.forward(x).clone()
    lv = [t.clone() for t in eng.head_outputs()]
    out, cnt = eng.detect(x, 0.25, 0.7, graph=True)
    out, cnt = out.clone(), cnt.clone()
    monkeypatch.setenv("YL_FUSE_PSA_TAIL", "0")
    ref = HipEngine.from_model(specs, sd, b, h, w, gpu_device, one_batch=one_batch)
    assert "psa_tail_mfma" not in [ref.op_kernel(i) for i in range(len(ref.prog.ops))]
    assert len(ref.prog.ops) == len(eng.prog.ops) + 3
    assert torch.equal(y, ref.forward(x))
    for p, q in zip(lv, ref.head_outputs()):
        assert torch.equal(p, q)
    o2, c2 = ref.detect(x, 0.25, 0.7, graph=False)
    assert torch.equal(cnt, c2) and all(torch.equal(out[i, :n], o2[i, :n]) for i, n in enumerate(cnt.tolist()))


def test_latency_and_throughput_profiles_agree_on_random_shapes(gpu_device):
    """The two profiles of the program (nn/lower.py lower(one_batch=...): other launches, same arithmetic) on a dozen random
    (map size, batch, head, threshold) cases: counts and detections bit for bit, eager against captured graph against its replay
    (the fused stem clears the NMS candidate counters of every replay), level maps bit for bit between the profiles."""
    import random

    rng = random.Random(4)
    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    for it in range(12):
        h, w, b = 32 * rng.randint(1, 20), 32 * rng.randint(1, 20), rng.choice([1, 2, 3, 8, 9, 16, 17])
        b = max(1, min(b, 640 * 640 * 16 // (h * w)))
        sd = synth_state_dict("n", 100 + it, cls_bias=[rng.choice([-6.0, -4.5, -3.5, -2.0])] * 3, cls_gain=rng.choice([4.0, 6.0]))
        x = torch.randint(0, 256, (b, 3, h, w), generator=torch.Generator().manual_seed(it), dtype=torch.uint8).to(gpu_device)
        conf = rng.choice([0.25, 0.1, 0.5])
        res = []
        for one_batch in (False, True):
            eng = HipEngine.from_model(specs, sd, b, h, w, gpu_device, one_batch=one_batch)
            o1, c1 = eng.detect(x, conf, 0.7, graph=False)
            o1, c1 = o1.clone(), c1.clone()
            eng.detect(x, conf, 0.7, graph=True)
            o3, c3 = eng.detect(x, conf, 0.7, graph=True)
            assert torch.equal(c1, c3), (it, h, w, b, one_batch)
            assert all(torch.equal(o1[i, :n], o3[i, :n]) for i, n in enumerate(c1.tolist())), (it, h, w, b, one_batch)
            res.append((o1, c1, [t.clone() for t in eng.head_outputs()]))
            del eng
        assert torch.equal(res[0][1], res[1][1]), (it, h, w, b)
        assert all(torch.equal(res[0][0][i, :n], res[1][0][i, :n]) for i, n in enumerate(res[0][1].tolist())), (it, h, w, b)
        assert all(torch.equal(p, q) for p, q in zip(res[0][2], res[1][2])), (it, h, w, b)


@pytest.mark.parametrize("side", [1, 2])
def test_side_lane_graphs_bit_identical_and_rebindable(gpu_device, monkeypatch, side):
    """lower(one_batch=True) with YL_ONE_BATCH_SIDE: the first levels' Detect branches on side lane 1, the program run as ONE GRAPH
    PER SEGMENT on a stream per lane (csrc/engine.hip detect_lanes; a single captured graph's branches are serialised by this
    runtime).  Same detections bit for bit as the program without lanes: eager, segment graphs, their replay, and the replay
    on a FRESH input tensor (the plan rewrites the stem node's input address); the single-handle form captures the lanes as branches of one graph."""
    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    b, h, w = 4, 320, 352
    sd = synth_state_dict("n", 7, cls_bias=[-3.5] * 3, cls_gain=6.0)
    xs = [torch.randint(0, 256, (b, 3, h, w), generator=torch.Generator().manual_seed(s), dtype=torch.uint8).to(gpu_device) for s in (1, 2, 3)]
    ref = HipEngine.from_model(specs, sd, b, h, w, gpu_device, one_batch=True)
    assert not any(o.get("lane", 0) for o in ref.prog.ops)
    want = []
    for x in xs:
        o, c = ref.detect(x, 0.25, 0.7, graph=False)
        want.append((o.clone(), c.clone()))
    assert int(want[0][1].sum()) > 0
    monkeypatch.setenv("YL_ONE_BATCH_SIDE", str(side))
    eng = HipEngine.from_model(specs, sd, b, h, w, gpu_device, one_batch=True)
    lanes = [o.get("lane", 0) for o in eng.prog.ops]
    assert sum(lanes) == 4 * side and lanes[0] == 0
    for mode in (dict(graph=False), dict(graph=True), dict(graph=True), dict(graph=True, stage="rebind")):
        for x, (o0, c0) in zip(xs, want):
            o, c = eng.detect(x, 0.25, 0.7, **mode)
            torch.cuda.synchronize()
            assert torch.equal(c, c0), mode
            assert all(torch.equal(o[i, :n], o0[i, :n]) for i, n in enumerate(c0.tolist())), mode
    # the single-handle form: the lanes as branches of one captured graph (serialised by the runtime, same results)
    o, c, hd = eng.prepare_detect(xs[1], 0.25, 0.7, max_det=300, slot=1)
    for _ in range(2):
        c.zero_()
        eng.launch_many([hd], [torch.cuda.current_stream()])
        torch.cuda.synchronize()
        assert torch.equal(c, want[1][1])
        assert all(torch.equal(o[i, :n], want[1][0][i, :n]) for i, n in enumerate(want[1][1].tolist()))
