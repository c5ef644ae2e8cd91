"""CPU: lowering of the YOLO11 graph to the engine program (no GPU needed)."""
import pytest

from yololite_amd.nn import arch
from yololite_amd.nn.lower import (OP_BNECK2, OP_C3K2S, OP_ATTENTION, OP_CONV, OP_DWCONV3, OP_PSA_TAIL, OP_SPPF_POOL, OP_STEM, OP_UPSAMPLE, lower,
                                   plane_pixels, view_bytes)
from yololite_amd.nn.synth import synth_state_dict


def _prog(scale, b=2, h=640, w=640):
    specs, _ = arch.parse_model(arch.yolo11_dict(scale))
    return lower(specs, synth_state_dict(scale, 0), b, h, w)


def _conv_gflop(prog):
    f = 0.0
    for o in prog.ops:
        if o["kind"] == OP_BNECK2:  # both 3x3 convs of a Bottleneck in one op
            for w in (o["weight"], o["post_weight"]):
                f += 2.0 * o["out"].H * o["out"].W * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3]
        if o["kind"] == OP_C3K2S:  # Bottleneck.cv1, Bottleneck.cv2 and C3k2.cv2 in one op
            for w in (o["weight"], o["post_weight"], o["stem_weight"]):
                f += 2.0 * o["out"].H * o["out"].W * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3]
        if o["kind"] == OP_PSA_TAIL:  # proj, ffn.0, ffn.1 and C2PSA.cv2 in one op (all 1x1)
            for w in [o["weight"]] + list(o["extra_weight"]):
                f += 2.0 * o["out"].H * o["out"].W * w.shape[0] * w.shape[1]
        if o["kind"] in (OP_CONV, OP_DWCONV3, OP_STEM):
            w = o["weight"]
            f += 2.0 * o["out"].H * o["out"].W * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3]
            if "post_weight" in o:  # a 1x1 conv fused into this conv's epilogue
                f += 2.0 * o["out"].H * o["out"].W * o["post_weight"].shape[0] * o["post_weight"].shape[1]
            if "stem_weight" in o and o["kind"] == OP_CONV:  # the network's first layer fused in front of this conv
                w0 = o["stem_weight"]
                f += 2.0 * o["inp"].H * o["inp"].W * w0.shape[0] * w0.shape[1] * w0.shape[2] * w0.shape[3]
            if "stem_weight" in o and o["kind"] == OP_DWCONV3:  # a further 1x1 behind depthwise + pointwise
                f += 2.0 * o["out"].H * o["out"].W * o["stem_weight"].shape[0] * o["stem_weight"].shape[1]
    return f / 1e9


@pytest.mark.parametrize("scale,gflop,nconv", [("n", 6.4811, 88), ("s", 21.4669, 88), ("x", 194.9038, 174)])
def test_algorithmic_flops_match_survey(scale, gflop, nconv):
    """2*MAC over all nn.Conv2d calls per image at 640x640 (DFL conv excluded
    here: it is 16*4*8400 MACs = 0.0011 GFLOP, folded into the decode kernel)."""
    p = _prog(scale, 1)
    got = _conv_gflop(p) + 2 * 16 * 4 * p.anchors / 1e9
    assert abs(got - gflop) < 2e-3 * gflop
    # C3k cv1+cv2 pairs are merged into one launch: count reference Conv2d calls
    calls = sum((0 if o["kind"] == OP_PSA_TAIL else 2 if o.get("name", "").endswith("cv1+cv2") else 1) + len(o.get("fused_names", []))
                for o in p.ops if o["kind"] in (OP_CONV, OP_DWCONV3, OP_STEM, OP_C3K2S, OP_BNECK2, OP_PSA_TAIL))
    assert calls + 1 == nconv  # + the DFL conv


def test_views_inside_workspace_and_aligned():
    p = _prog("n", 3, 96, 128)
    assert p.anchors == 12 * 16 + 6 * 8 + 3 * 4
    for o in p.ops:
        for key in ("inp", "out", "res", "inp2"):
            v = o.get(key)
            if v is None:
                continue
            assert v.offset % 256 == 0 and v.coff + v.C <= v.pitch
            assert v.offset + view_bytes(p.batch, v) <= p.ws_bytes
            if v.dtype == 0:  # bf16 views are plane ranges: everything in units of 8 channels
                assert v.C % 8 == 0 and v.pitch % 8 == 0 and v.coff % 8 == 0
    kinds = [o["kind"] for o in p.ops]
    assert kinds.count(OP_SPPF_POOL) == 1 and kinds.count(OP_ATTENTION) == 1
    # both nn.Upsample layers are read through by the 1x1 conv behind their Concat (yl_op.in2)
    assert kinds.count(OP_UPSAMPLE) == 0 and sum("inp2" in o for o in p.ops) == 2
    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    q = lower(specs, synth_state_dict("n", 0), 3, 96, 128, fuse_post=False)
    assert [o["kind"] for o in q.ops].count(OP_UPSAMPLE) == 2 and not any("inp2" in o or "post_weight" in o for o in q.ops)


def test_plane_geometry_known_answers():
    """pixel(b, y, x) = g0 + b*(H+1)*(W+1) + y*(W+1) + x must leave room for every 3x3 tap."""
    for B, H, W in [(1, 2, 2), (3, 20, 20), (32, 80, 80), (2, 3, 4)]:
        pp = plane_pixels(B, H, W)
        g0, w1, img = W + 2, W + 1, (H + 1) * (W + 1)
        first = g0 - w1 - 1                                   # tap (-1, -1) of pixel (0, 0, 0)
        last = g0 + (B - 1) * img + (H - 1) * w1 + (W - 1) + w1 + 1  # tap (+1, +1) of the last pixel
        assert first == 0 and last < pp and pp % 16 == 0
    assert plane_pixels(1, 2, 2) == 16 and plane_pixels(32, 80, 80) == (82 + 32 * 81 * 81 + 1 + 15) // 16 * 16


@pytest.mark.parametrize("scale", ["n", "x"])
def test_buffers_never_overlap(scale):
    """No recycling: the guard pixels of a buffer are zeroed once and must never be overwritten."""
    p = _prog(scale, 2, 96, 128)
    spans = {}
    for o in p.ops:
        for key in ("inp", "out", "res"):
            v = o.get(key)
            if v is not None:
                spans[v.offset] = max(spans.get(v.offset, 0), view_bytes(p.batch, v))
    for L in p.levels:  # a level = the nc8 class logits (a planar buffer) + the decoded boxes, fp32 (cx, cy, w, h) per anchor
        assert L["offset"] == -1 and L["offset_cls"] in spans and L["nc8"] % 8 == 0 and L["nc8"] >= p.nc
        assert L["offset_xywh"] >= 0 and L["offset_xywh"] not in spans
        spans[L["offset_xywh"]] = p.batch * L["H"] * L["W"] * 16
    dfl = [o for o in p.ops if o.get("dfl_level")]
    assert sorted(o["dfl_level"] for o in dfl) == [1, 2, 3] and all(o["out"].C == 64 and not o["dfl_keep"] for o in dfl)
    # (the rounds 1-5 form: 64 bf16 DFL logits per anchor, stored and decoded by their readers)
    specs, _ = arch.parse_model(arch.yolo11_dict(scale))
    q = lower(specs, synth_state_dict(scale, 0), 2, 96, 128, dfl_epilogue=False)
    assert all(L["offset"] >= 0 and L["offset_xywh"] == -1 for L in q.levels) and not any(o.get("dfl_level") for o in q.ops)
    k = lower(specs, synth_state_dict(scale, 0), 2, 96, 128, keep_logits=True)
    assert all(L["offset"] >= 0 and L["offset_xywh"] >= 0 for L in k.levels) and all(o["dfl_keep"] for o in k.ops if o.get("dfl_level"))
    t = sorted(spans.items())
    for (o1, s1), (o2, _) in zip(t, t[1:]):
        assert o1 + s1 <= o2
    assert t[-1][0] + t[-1][1] <= p.ws_bytes


def _prog_noreuse(scale, b=1, h=640, w=640):
    return _prog(scale, b, h, w)


def test_concat_members_are_slices_of_one_buffer():
    """Concat never copies: layer 6 and the upsampled layer 10 share layer 12's buffer."""
    p = _prog_noreuse("n", 1)
    v6, v11, v12 = p.layer_views[6], p.layer_views[11], p.layer_views[12]
    assert v6.offset == v12.offset == v11.offset
    assert (v11.coff, v11.C, v6.coff, v6.C, v12.C) == (0, 256, 256, 128, 384)
    assert not any(o.get("name", "").endswith(".copy") for o in p.ops)


@pytest.mark.parametrize("scale,with_1x1", [("n", False), ("x", False), ("n", True), ("s", True), ("x", True)])
def test_fp8_pass_structure(scale, with_1x1, monkeypatch):
    """lower.apply_fp8_mirrors / apply_fp8: every fp8 conv reads e4m3 planes that were written earlier (by a quantize op,
    by an fp8 conv, or by a producer's epilogue copy) with the scale the reader assumes; nothing but fp8 convs reads an
    e4m3 view; algorithmic FLOPs are unchanged."""
    from yololite_amd.nn.lower import FP8, OP_QUANT, fp8_candidate

    monkeypatch.setenv("YL_FP8_MIRROR", "1" if with_1x1 else "0")
    monkeypatch.setenv("YL_FP8_1X1_GF", "0")
    specs, _ = arch.parse_model(arch.yolo11_dict(scale))
    sd = synth_state_dict(scale, 0)
    p = lower(specs, sd, 2, 64, 96)
    cands = [o for o in p.ops if fp8_candidate(o, allow_1x1=with_1x1)]
    scales = {o["name"]: 0.01 * (1 + i) for i, o in enumerate(cands)}
    p8 = lower(specs, sd, 2, 64, 96, fp8_scales=scales)
    assert abs(_conv_gflop(p8) - _conv_gflop(p)) < 1e-9
    written = {}

    def wrote(v, sc):
        assert v.coff % 16 == 0 and v.C % 16 == 0 and sc > 0
        for pl in range(v.coff // 16, (v.coff + v.C) // 16):
            written[(v.offset, pl)] = sc

    n8 = 0
    for o in p8.ops:
        if o["kind"] == OP_CONV and o["inp"].dtype == FP8:
            n8 += 1
            v = o["inp"]
            for pl in range(v.coff // 16, (v.coff + v.C) // 16):
                assert written.get((v.offset, pl)) == o["in_scale"], (o["name"], pl)
            assert o["name"] in scales
        else:
            for key in ("inp", "res", "inp2"):
                assert o.get(key) is None or o[key].dtype != FP8, o.get("name")
        if o["out"].dtype == FP8:
            assert o["kind"] in (OP_QUANT, OP_CONV)
            wrote(o["out"], o["out_scale"])
        if "out8" in o:  # the producer's epilogue stores the e4m3 copy of a channel range of its bf16 output
            v = o["out8"]
            assert v.dtype == FP8 and o["out"].dtype != FP8 and o["out8_c0"] % 8 == 0 and o["out8_c0"] + v.C <= o["out"].C
            assert (v.H, v.W) == (o["out"].H, o["out"].W)
            wrote(v, o["out8_scale"])
    if scale == "x":  # every e4m3 tensor comes out of a producer's epilogue: no quantize pass left, the stem stores fp8 only
        assert not any(o["kind"] == OP_QUANT for o in p8.ops) and p8.ops[0].get("out8_only") == 1
    # a copy-only producer: the copy covers the whole output and NO op of the program reads the bf16 planes it would have written
    only = [o for o in p8.ops if o.get("out8_only")]
    for o in only:
        assert o["out8_c0"] == 0 and o["out8"].C == o["out"].C and o.get("res") is None
        mine = {(o["out"].offset, pl) for pl in range(o["out"].coff // 8, (o["out"].coff + o["out"].C) // 8)}
        for r in p8.ops:
            for key in ("inp", "res", "inp2"):
                v = r.get(key)
                if v is not None and v.dtype != FP8:
                    assert not ({(v.offset, pl) for pl in range(v.coff // 8, (v.coff + v.C) // 8)} & mine), (o["name"], r["name"], key)
    if scale == "x" and not with_1x1:  # the stem and model.2.cv2 (read by model.3 alone)
        assert [o["name"] for o in only] == ["model.0", "model.2.cv2"], [o["name"] for o in only]
    if with_1x1:
        assert n8 >= {"n": 35, "s": 50, "x": 130}[scale]
        assert sum(o["kind"] == OP_CONV and o["out"].dtype == FP8 for o in p8.ops) >= {"n": 15, "s": 25, "x": 80}[scale]
    else:
        assert n8 == len(cands) and n8 >= (19 if scale == "n" else 70)
        assert sum(o["kind"] == OP_CONV and o["out"].dtype == FP8 for o in p8.ops) >= (6 if scale == "n" else 30)
    assert p8.ws_bytes > p.ws_bytes


def test_detect_levels_are_grouped_at_the_end_of_the_program():
    """nn/lower.py::group_detect_levels: the head moves behind the neck, the levels' copies of a conv stand side by side
    and carry one yl_op.group id; every op still reads tensors that an earlier op (or the input) wrote."""
    import os

    from yololite_amd.nn import arch
    from yololite_amd.nn.lower import lower
    from yololite_amd.nn.synth import synth_state_dict

    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    sd = synth_state_dict("n", 0)
    for mode, groups in (("2", {1: 2, 2: 3, 3: 3}), ("1", {1: 3, 2: 3, 3: 3}), ("3", {1: 5, 2: 6}), ("4", {1: 5, 2: 3, 3: 3}), ("0", {})):
        os.environ["YL_GROUP_LEVELS"] = mode
        try:
            p = lower(specs, sd, 2, 128, 128)
        finally:
            os.environ.pop("YL_GROUP_LEVELS", None)
        names = [o["name"] for o in p.ops]
        head = [i for i, n in enumerate(names) if n.startswith("model.23.")]
        assert len(head) == 12
        if mode != "0":
            assert head == list(range(len(names) - 12, len(names)))  # the whole head is the tail of the program
        seen = {}
        for o in p.ops:
            g = o.get("group", 0)
            if g:
                seen[g] = seen.get(g, 0) + 1
        assert seen == groups, (mode, seen)
        # members of a group are consecutive
        for g in seen:
            idx = [i for i, o in enumerate(p.ops) if o.get("group", 0) == g]
            assert idx == list(range(idx[0], idx[0] + len(idx)))
        # dataflow of the moved ops: a head op's input planes were written by an earlier op
        written = set()
        for o in p.ops:
            if o["name"].startswith("model.23."):
                v = o["inp"]
                for c in range(v.coff // 8, (v.coff + v.C) // 8):
                    assert (v.offset, c) in written, (mode, o["name"])
            v = o["out"]
            for c in range(v.coff // 8, (v.coff + v.C + 7) // 8):
                written.add((v.offset, c))


def test_c3k_resident_limits():
    """Which C3k blocks the per-image resident kernel takes (mirror of c3k_resident_fits in csrc/resident.hip)."""
    from yololite_amd.nn.lower import c3k_resident_fits

    assert c3k_resident_fits(64, 20, 20) and c3k_resident_fits(64, 12, 20) and c3k_resident_fits(64, 19, 19)
    assert not c3k_resident_fits(64, 21, 21) and not c3k_resident_fits(64, 40, 40)   # LDS
    assert not c3k_resident_fits(32, 20, 20) and not c3k_resident_fits(128, 10, 10)  # inner width 64 only


def test_sppf_resident_limits():
    """Which SPPF layers the resident kernel takes (mirror of sppf_resident_fits in csrc/sppf_resident.hip)."""
    from yololite_amd.nn.lower import sppf_resident_fits

    assert sppf_resident_fits(256, 256, 256, 20, 20) and sppf_resident_fits(256, 256, 0, 12, 20) and sppf_resident_fits(256, 256, 256, 21, 21)
    assert not sppf_resident_fits(256, 256, 256, 22, 22) and not sppf_resident_fits(256, 256, 0, 40, 40)  # LDS: 16 planes + the filter ring
    assert not sppf_resident_fits(512, 512, 512, 20, 20) and not sppf_resident_fits(256, 256, 128, 20, 20)  # YOLO11n's widths only


def test_lowering_uses_resident_c3k_from_batch_8(monkeypatch):
    from yololite_amd.nn import arch
    from yololite_amd.nn.lower import OP_C3K, OP_PSA, OP_PSA_TAIL, OP_SPPF, lower, psa_tail_fits
    from yololite_amd.nn.synth import synth_state_dict

    monkeypatch.delenv("YL_FUSE_C3K", raising=False)
    monkeypatch.delenv("YL_FUSE_SPPF", raising=False)
    monkeypatch.delenv("YL_FUSE_PSA_TAIL", raising=False)
    assert psa_tail_fits(128, 256, 1) and not psa_tail_fits(256, 512, 1) and not psa_tail_fits(128, 256, 2)
    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    sd = synth_state_dict("n", 0)
    big = lower(specs, sd, 32, 640, 640)
    assert [o["name"] for o in big.ops if o["kind"] == OP_C3K] == ["model.8.m.0", "model.22.m.0"]
    assert [o["name"] for o in big.ops if o["kind"] == OP_PSA] == ["model.10.m.0"]
    sppf = [o for o in big.ops if o["kind"] == OP_SPPF]  # layer 9 with C2PSA.cv1 behind it: its own output is never stored
    assert [o["name"] for o in sppf] == ["model.9"] and len(sppf[0]["extra_weight"]) == 2
    assert sppf[0]["fused_names"] == ["model.9.cv1", "model.9.m", "model.9.cv2", "model.10.cv1"]
    assert sppf[0]["out"] == [o for o in big.ops if o["kind"] == OP_PSA][0]["inp"]
    assert not [o for o in big.ops if o["name"] == "model.10.cv1"]
    small = lower(specs, sd, 1, 640, 640)
    # (without the one-launch PSABlock the layer is qkv, attention, pe and ONE launch for its pointwise tail -- proj, ffn.0, ffn.1 and
    # C2PSA.cv2: four ops where the resident form has two, PSABlock + cv2)
    assert not [o for o in small.ops if o["kind"] in (OP_C3K, OP_PSA, OP_SPPF)] and len(small.ops) == len(big.ops) + 15
    monkeypatch.setenv("YL_FUSE_PSA", "0")  # without the resident PSABlock the SPPF launch still carries C2PSA.cv1
    nopsa = lower(specs, sd, 32, 640, 640)
    assert [len(o["extra_weight"]) for o in nopsa.ops if o["kind"] == OP_SPPF] == [2] and len(nopsa.ops) == len(big.ops) + 2
    tail = [o for o in nopsa.ops if o["kind"] == OP_PSA_TAIL]
    assert [o["name"] for o in tail] == ["model.10.m.0.tail"]
    assert tail[0]["fused_names"] == ["model.10.m.0.attn.proj", "model.10.m.0.ffn.0", "model.10.m.0.ffn.1", "model.10.cv2"]
    assert tail[0]["res"].C == 256 and tail[0]["inp"].C == 128 and tail[0]["out"].C == 256
    assert not [o for o in nopsa.ops if o["name"] in ("model.10.cv2", "model.10.m.0.ffn.0")]
    monkeypatch.setenv("YL_FUSE_PSA_TAIL", "0")
    assert len(lower(specs, sd, 32, 640, 640).ops) == len(big.ops) + 5
    monkeypatch.delenv("YL_FUSE_PSA_TAIL")
    monkeypatch.delenv("YL_FUSE_PSA")
    lat = lower(specs, sd, 32, 640, 640, one_batch=True)  # the latency profile: resident C3k with two workgroups per image, SPPF / PSA unfused
    assert [o["split"] for o in lat.ops if o["kind"] == OP_C3K] == [2, 2] and all(o["split"] == 0 for o in big.ops if o["kind"] == OP_C3K)
    assert not [o for o in lat.ops if o["kind"] in (OP_PSA, OP_SPPF)] and len(lat.ops) == len(big.ops) + 3 + 2
    assert len([o for o in lat.ops if o["kind"] == OP_PSA_TAIL]) == 1
    s_scale = lower(arch.parse_model(arch.yolo11_dict("s"))[0], synth_state_dict("s", 0), 32, 640, 640)
    assert not [o for o in s_scale.ops if o["kind"] == OP_C3K]   # inner width 128: the maps do not fit


def test_side_lane_head_placement(monkeypatch):
    """YL_ONE_BATCH_SIDE=n (opt-in): the Detect branches of the first n levels follow the layer that feeds them on lane 1, the
    other levels' copies stay behind the network (grouped when there are two of them); op 0 stays on the main lane."""
    from yololite_amd.nn import arch
    from yololite_amd.nn.lower import lower
    from yololite_amd.nn.synth import synth_state_dict

    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    sd = synth_state_dict("n", 0, cls_bias=[0.0, 0.0, 0.0])
    base = lower(specs, sd, 32, 640, 640, one_batch=True)
    assert not any(o.get("lane", 0) for o in base.ops)
    for side in (1, 2):
        monkeypatch.setenv("YL_ONE_BATCH_SIDE", str(side))
        prog = lower(specs, sd, 32, 640, 640, one_batch=True)
        assert sorted(o["name"] for o in prog.ops) == sorted(o["name"] for o in base.ops)
        names = [o["name"] for o in prog.ops]
        for lvl, feed in ((0, 16), (1, 19))[:side]:
            first = names.index(f"model.23.cv3.{lvl}.0.0+pw")
            assert names[first - 1].startswith(f"model.{feed}.") and not names[first + 4].startswith("model.23.")
            assert [o.get("lane", 0) for o in prog.ops[first:first + 4]] == [1, 1, 1, 1]
        assert prog.ops[0].get("lane", 0) == 0 and sum(o.get("lane", 0) for o in prog.ops) == 4 * side
        tail = [o for o in prog.ops if o["name"].startswith("model.23.") and not o.get("lane", 0)]
        assert len(tail) == 4 * (3 - side) and names[-1].startswith("model.23.")
        assert (len({o.get("group", 0) for o in tail} - {0}) == 2) == (side == 1)
    # a throughput program never gets lanes
    assert not any(o.get("lane", 0) for o in lower(specs, sd, 32, 640, 640).ops)


def test_latency_profile_keeps_the_fused_stem_and_c3k2_outputs_cacheable(monkeypatch):
    """lower(one_batch=True): the fused stem conv and the narrow C3k2 blocks carry `cacheable` (yl_op.reserved3 bit 8: no
    non-temporal hint on their stores -- their readers are the next launches of a caller that runs one batch at a time); the
    throughput profile marks nothing, and YL_LAT_CACHEABLE=0 turns the marking off."""
    from yololite_amd.nn.lower import OP_C3K2S, OP_CONV

    specs, _ = arch.parse_model(arch.yolo11_dict("n"))
    sd = synth_state_dict("n", 0)
    thr = lower(specs, sd, 32, 640, 640)
    assert not any(o.get("cacheable") for o in thr.ops)
    lat = lower(specs, sd, 32, 640, 640, one_batch=True)
    marked = [o for o in lat.ops if o.get("cacheable")]
    assert marked and all(o["kind"] == OP_C3K2S or (o["kind"] == OP_CONV and "stem_weight" in o) for o in marked)
    assert sum(o["kind"] == OP_C3K2S for o in marked) == sum(o["kind"] == OP_C3K2S for o in lat.ops) == 3  # layers 2, 4, 16
    assert sum(o["kind"] == OP_CONV for o in marked) == 1                        # the conv with the network's first layer fused in front
    monkeypatch.setenv("YL_LAT_CACHEABLE", "0")
    assert not any(o.get("cacheable") for o in lower(specs, sd, 32, 640, 640, one_batch=True).ops)
