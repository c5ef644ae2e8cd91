"""The post-NMS parity gate itself (oracle/parity_gate.py, test infrastructure): every explanation category on hand-built
cases, and the rule that explanations need evidence."""
import torch

from oracle.parity_gate import format_yardstick_ok, parity_gate

CONF, IOU, BAND = 0.25, 0.7, 0.08


def _b(x1, y1, x2, y2, s, c):
    return [float(x1), float(y1), float(x2), float(y2), float(s), float(c)]


def test_identical_detections_pass():
    d = [torch.tensor([_b(10, 10, 50, 50, 0.9, 3), _b(100, 100, 180, 160, 0.6, 1)])]
    g = parity_gate(d, d, CONF, IOU, BAND)
    assert g["ok"] and g["matched"] == 2 and g["unexplained"] == 0 and g["max_dconf"] == 0.0


def test_score_drift_beyond_band_fails_and_class_flip_of_a_pair_fails():
    ref = [torch.tensor([_b(10, 10, 50, 50, 0.9, 3)])]
    assert not parity_gate(ref, [torch.tensor([_b(10, 10, 50, 50, 0.7, 3)])], CONF, IOU, BAND)["ok"]  # |dconf| 0.2 > band
    g = parity_gate(ref, [torch.tensor([_b(10, 10, 50, 50, 0.9, 4)])], CONF, IOU, BAND)
    assert g["class_mismatch"] == 1 and not g["ok"]


def test_one_sided_boxes_need_a_stated_mechanism():
    ref = [torch.tensor([_b(10, 10, 50, 50, 0.9, 3)])]
    # near the threshold: explained
    g = parity_gate(ref, [torch.tensor([_b(10, 10, 50, 50, 0.9, 3), _b(200, 200, 240, 240, 0.30, 2)])], CONF, IOU, BAND,
                    max_explained_share=1.0)
    assert g["near_threshold"] == 1 and g["unexplained"] == 0 and g["ok"]
    # a confident extra box with nothing near it: unexplained
    g = parity_gate(ref, [torch.tensor([_b(10, 10, 50, 50, 0.9, 3), _b(200, 200, 240, 240, 0.8, 2)])], CONF, IOU, BAND)
    assert g["unexplained"] == 1 and not g["ok"]
    # the other side holds a same-class box that overlaps it near the NMS threshold (its suppressor over there): nms_flip
    ref2 = [torch.tensor([_b(10, 10, 50, 50, 0.9, 3)])]
    det2 = [torch.tensor([_b(10, 10, 50, 50, 0.9, 3), _b(10, 18, 50, 58, 0.6, 3)])]  # IoU 0.667 with the kept box
    g = parity_gate(ref2, det2, CONF, IOU, BAND, max_explained_share=1.0)
    assert g["nms_flip"] == 1 and g["ok"]


def test_explained_share_is_capped():
    ref = [torch.tensor([_b(10, 10, 50, 50, 0.9, 3)])]
    det = [torch.tensor([_b(10, 10, 50, 50, 0.9, 3)] + [_b(100 + 50 * k, 100, 140 + 50 * k, 140, 0.28, 1) for k in range(3)])]
    g = parity_gate(ref, det, CONF, IOU, BAND)
    assert g["unexplained"] == 0 and g["explained_share"] == 0.75 and not g["ok"]


def test_argmax_flip_needs_the_other_sides_pre_nms_tensor():
    """A near-tied arg-max that went the other way: the anchor exists over there with another class (and lost that class's
    NMS).  Explained only with the pre-NMS tensor as evidence."""
    nc, A = 3, 2
    y = torch.zeros(1, 4 + nc, A)
    y[0, :4, 0] = torch.tensor([50.0, 50.0, 20.0, 20.0])
    y[0, 4:, 0] = torch.tensor([0.60, 0.62, 0.10])       # reference: class 1 by 0.02
    y[0, :4, 1] = torch.tensor([50.0, 52.0, 20.0, 20.0])
    y[0, 4:, 1] = torch.tensor([0.05, 0.90, 0.05])       # its class-1 suppressor
    ref = [torch.tensor([_b(40, 42, 60, 62, 0.9, 1)])]
    det = [torch.tensor([_b(40, 42, 60, 62, 0.9, 1), _b(40, 40, 60, 60, 0.61, 0)])]  # the other side: class 0 won, nothing suppresses it
    g = parity_gate(ref, det, CONF, IOU, BAND, ref_y=y, max_explained_share=1.0)
    assert g["argmax_flip"] == 1 and g["unexplained"] == 0 and g["ok"]
    assert parity_gate(ref, det, CONF, IOU, BAND, max_explained_share=1.0)["unexplained"] == 1  # no tensor, no excuse
    y[0, 4, 0] = 0.40                                                                  # not a near tie: no excuse either
    assert parity_gate(ref, det, CONF, IOU, BAND, ref_y=y, max_explained_share=1.0)["unexplained"] == 1


def test_format_yardstick_compares_with_the_formats_own_error():
    got = dict(class_mismatch=0, unexplained=0, max_dconf=0.06, matched_frac=0.93, matched=90, ref=100, det=100, explained_share=0.1)
    fmt = dict(unexplained=0, max_dconf=0.07, matched_frac=0.95, explained_share=0.1)
    assert format_yardstick_ok(got, fmt, BAND)["ok"]
    assert not format_yardstick_ok(dict(got, max_dconf=0.12), dict(fmt, max_dconf=0.03), 0.2)["ok"]  # 1.5 x the format's + 0.01


def test_fixed_gate_keywords_p99_count_bias_and_max_band():
    """Round 6, the fixed gate of the e4m3 configuration: ``max_band`` bounds the maximum |dconf| without widening what counts as
    explained (``band`` keeps sizing the explanations), ``p99_band`` bounds the 99th percentile, ``max_count_bias`` |det / ref - 1|."""
    boxes = [_b(10 + 60 * i, 10, 50 + 60 * i, 50, 0.9, 1) for i in range(10)]
    ref = [torch.tensor(boxes)]
    drift = [list(b) for b in boxes]
    drift[0][4] = 0.55                                     # one outlier: |dconf| 0.35
    det = [torch.tensor(drift)]
    assert not parity_gate(ref, det, CONF, IOU, 0.1)["ok"]                                   # max > band
    g = parity_gate(ref, det, CONF, IOU, 0.1, max_band=0.4, p99_band=0.4)
    assert g["ok"] and g["max_dconf"] == 0.35 and g["max_band"] == 0.4
    drift[1][4] = 0.6                                      # a second one: with 10 pairs the 99th percentile is the 9th smallest
    det2 = [torch.tensor(drift)]
    assert parity_gate(ref, det2, CONF, IOU, 0.1, max_band=0.4, p99_band=0.4)["ok"]
    assert not parity_gate(ref, det2, CONF, IOU, 0.1, max_band=0.4, p99_band=0.2)["ok"]
    # a confident one-sided box stays unexplained however wide max_band is: only `band` sizes "near the threshold"
    extra = [torch.tensor(boxes + [_b(700, 10, 740, 50, 0.5, 2)])]
    g = parity_gate(ref, extra, CONF, IOU, 0.1, max_band=0.9, max_explained_share=1.0)
    assert g["unexplained"] == 1 and not g["ok"]
    g = parity_gate(ref, extra, CONF, IOU, 0.3, max_explained_share=1.0)                      # (band 0.3: 0.5 <= 0.25 + 0.3)
    assert g["near_threshold"] == 1 and g["ok"]
    # count bias: 10 reference boxes, 14 detections of which 4 are near the threshold (explained) -> bias 0.4
    many = [torch.tensor(boxes + [_b(700 + 60 * i, 10, 740 + 60 * i, 50, 0.27, 2) for i in range(4)])]
    assert parity_gate(ref, many, CONF, IOU, 0.1, max_explained_share=1.0)["ok"]
    g = parity_gate(ref, many, CONF, IOU, 0.1, max_explained_share=1.0, max_count_bias=0.25)
    assert abs(g["count_bias"] - 0.4) < 1e-6 and not g["ok"]
