"""Hand-derived known-answer tests for non_max_suppression (reference utils/ops.py:138-273 around
torchvision.ops.nms, ops.py:265).  The expected rows below were worked out BY HAND from the reference's statements
(the derivation is next to each case) -- they do not come from oracle.nms_greedy or from any implementation here --
so they pin the oracle (tests/test_nms_kat.py, CPU) and the HIP kernel (same file, -m gpu) independently.

All coordinates are chosen exactly representable in fp32 (or their rounding is part of the derivation); IoU thresholds
0.5 / 0.25 are exact in binary so `iou > thr` has one meaning in float and in double (torchvision's CPU kernel compares
a float IoU with a double threshold, its CUDA kernel with a float one: they differ only for thresholds like 0.7 that
are not exactly representable, which is why no case sits on such a threshold).
"""
import numpy as np

NC = 80


def _pred(rows, A=8):
    """rows: (cx, cy, w, h, cls, score) -> (1, 4 + NC, A) fp32; unused anchors score 0."""
    p = np.zeros((1, 4 + NC, A), np.float32)
    for a, (cx, cy, w, h, c, s) in enumerate(rows):
        p[0, :4, a] = (cx, cy, w, h)
        p[0, 4 + c, a] = s
    return p


def cases():
    out = []
    # 1. IoU exactly == threshold keeps (ops.py:265 -> torchvision: suppress when iou > thr, STRICT).
    #    A = xyxy (0,0,16,16) area 256; B = (0,0,16,8) area 128; inter 128; union 256; IoU = 0.5 exactly.
    #    thr 0.5 -> 0.5 > 0.5 false -> both kept, score order A, B.  C = (0,0,16,9) area 144, inter(A,C) = 144,
    #    union 256 -> 0.5625 > 0.5 -> C (lowest score) suppressed by A.
    rows = [(8, 8, 16, 16, 3, 0.9), (8, 4, 16, 8, 3, 0.8), (8, 4.5, 16, 9, 3, 0.7)]
    exp = [[0, 0, 16, 16, 0.9, 3], [0, 0, 16, 8, 0.8, 3]]
    out.append(("iou_equal_threshold_keeps", _pred(rows), dict(conf_thres=0.25, iou_thres=0.5), exp))

    # 2. conf filter is STRICT (ops.py:203 `> conf_thres`, :244): a score exactly 0.25 is dropped, 0.25 + 2^-20 kept.
    rows = [(50, 50, 10, 10, 1, 0.25), (100, 100, 10, 10, 1, 0.25 + 2.0 ** -20)]
    exp = [[95, 95, 105, 105, np.float32(0.25 + 2.0 ** -20), 1]]
    out.append(("conf_strict", _pred(rows), dict(conf_thres=0.25, iou_thres=0.5), exp))

    # 3. class offset (ops.py:258,264: boxes + cls * 7680): identical boxes of DIFFERENT classes never suppress each other
    #    (offset 7680 * (5 - 4) moves them 7680 px apart) unless agnostic (offset 0 -> IoU 1 > thr -> lower score dropped).
    rows = [(20, 20, 8, 8, 4, 0.6), (20, 20, 8, 8, 5, 0.5)]
    out.append(("class_offset_separates", _pred(rows), dict(conf_thres=0.25, iou_thres=0.5),
                [[16, 16, 24, 24, 0.6, 4], [16, 16, 24, 24, 0.5, 5]]))
    out.append(("agnostic_merges", _pred(rows), dict(conf_thres=0.25, iou_thres=0.5, agnostic=True),
                [[16, 16, 24, 24, 0.6, 4]]))

    # 4. class-79 offset quantisation (ops.py:264 is an fp32 add): offset 79 * 7680 = 606720; fp32 spacing there is
    #    2^-4 = 0.0625.  A = (0,0,16,16), B = (0,0,16,8.02):
    #      class 0 (offset 0): inter = 16 * 8.02 = 128.32, union = 256 + 128.32 - 128.32 = 256, IoU = 0.50125 > 0.5
    #        -> B suppressed.
    #      class 79: y2(B) + 606720 = 606728.02 rounds to 606728.0 (nearest multiple of 0.0625; .02 < .03125), every other
    #        coordinate is integral -> shifted B = shifted (0,0,16,8): inter 128, union 256, IoU = 0.5, not > 0.5
    #        -> B KEPT.  Returned rows carry the UNshifted boxes (ops.py:266 indexes x): y2(B) = fl32(8.02).
    #    (cy = 4.01, h = 8.02: y1 = fl(4.01) - fl(8.02)/2 = 0 exactly, y2 = 2 fl(4.01) = fl(8.02).)
    y2 = float(np.float32(4.01) + np.float32(8.02) / np.float32(2))
    for cls, exp in ((0, [[0, 0, 16, 16, 0.9, 0]]), (79, [[0, 0, 16, 16, 0.9, 79], [0, 0, 16, y2, 0.8, 79]])):
        rows = [(8, 8, 16, 16, cls, 0.9), (8, 4.01, 16, 8.02, cls, 0.8)]
        out.append((f"offset_quantisation_cls{cls}", _pred(rows), dict(conf_thres=0.25, iou_thres=0.5), exp))

    # 5. chain: suppression is by KEPT boxes only (greedy, score order).  A keeps, B (IoU(A,B) = 0.6 > 0.5) is dropped,
    #    C overlaps only B strongly (IoU(B,C) = 0.6) but A weakly (IoU(A,C) = 0.25, not > 0.5) -> C kept although
    #    B would have suppressed it.  A = (0,0,10,8), B = (0,2,10,10): inter 60, union 100 -> 0.6;
    #    C = (0,4,10,12): inter(A,C) = 40, union 120 -> 1/3; use thr 0.5: 1/3 not > 0.5 -> keep.
    rows = [(5, 4, 10, 8, 7, 0.9), (5, 6, 10, 8, 7, 0.8), (5, 8, 10, 8, 7, 0.7)]
    out.append(("greedy_chain", _pred(rows), dict(conf_thres=0.25, iou_thres=0.5),
                [[0, 0, 10, 8, 0.9, 7], [0, 4, 10, 12, 0.7, 7]]))

    # 6. max_det truncation AFTER nms (ops.py:266 `i[:max_det]`), score-descending: 5 disjoint boxes, max_det 3.
    rows = [(20 + 40 * i, 20, 10, 10, 2, s) for i, s in enumerate((0.5, 0.9, 0.3, 0.7, 0.6))]
    exp = [[55, 15, 65, 25, 0.9, 2], [135, 15, 145, 25, 0.7, 2], [175, 15, 185, 25, 0.6, 2]]
    out.append(("max_det_after_nms", _pred(rows), dict(conf_thres=0.25, iou_thres=0.5, max_det=3), exp))

    # 7. classes filter (ops.py:247-248) applies BEFORE nms: a high-score box of a filtered-out class cannot suppress.
    rows = [(8, 8, 16, 16, 9, 0.95), (8, 8, 16, 15, 11, 0.5)]
    out.append(("classes_filter_before_nms", _pred(rows), dict(conf_thres=0.25, iou_thres=0.5, classes=[11], agnostic=True),
                [[0, 0.5, 16, 15.5, 0.5, 11]]))

    # 8. best class = FIRST maximum (ops.py:243 `cls.max(1)`): one anchor with equal scores in classes 12 and 40 -> 12.
    p = _pred([(30, 30, 10, 10, 40, 0.6)])
    p[0, 4 + 12, 0] = 0.6
    out.append(("first_max_class", p, dict(conf_thres=0.25, iou_thres=0.5), [[25, 25, 35, 35, 0.6, 12]]))

    # 9. equal scores: torchvision's sort order among exactly equal scores is implementation-defined; the
    #    convention here (oracle and HIP kernel alike, documented in include/yl_amd.h) is lower anchor index first.
    #    Two overlapping boxes (IoU 0.5625 > 0.5) with the same score: anchor 0 wins.  NOT a reference fact.
    rows = [(8, 8, 16, 16, 3, 0.75), (8, 4.5, 16, 9, 3, 0.75)]
    out.append(("equal_score_convention_lower_anchor_first", _pred(rows), dict(conf_thres=0.25, iou_thres=0.5),
                [[0, 0, 16, 16, 0.75, 3]]))
    # 10. zero-area boxes (no minimum size in ops.py:138-273; torchvision: IoU = inter / (area_i + area_j - inter)).  A = (0,0,16,16);
    #     Z1 = Z2 = the vertical segment x = 8, y in [0, 16] (w = 0: x1 = x2 = 8).  IoU(A, Z) = 0 / (256 + 0 - 0) = 0, not > 0.5;
    #     IoU(Z1, Z2) = 0 / (0 + 0 - 0) = NaN, and NaN > 0.5 is false: nothing is suppressed, all three come back in score order.
    rows = [(8, 8, 16, 16, 3, 0.9), (8, 8, 0, 16, 3, 0.8), (8, 8, 0, 16, 3, 0.7)]
    out.append(("zero_area_boxes_suppress_nothing", _pred(rows), dict(conf_thres=0.25, iou_thres=0.5),
                [[0, 0, 16, 16, 0.9, 3], [8, 0, 8, 16, 0.8, 3], [8, 0, 8, 16, 0.7, 3]]))

    # 11. exactly equal scores ACROSS classes: the same box at two anchors, classes 5 and 2, score 0.75 each.  Class-aware: the
    #     offsets 5 * 7680 and 2 * 7680 separate them -> both kept; their order among equal scores is the convention of case 9
    #     (lower anchor first; NOT a reference fact -- a caller of the reference must not rely on it either).  Agnostic: IoU 1 > 0.5
    #     -> the first in that order suppresses the other.
    rows = [(20, 20, 8, 8, 5, 0.75), (20, 20, 8, 8, 2, 0.75)]
    out.append(("equal_scores_across_classes", _pred(rows), dict(conf_thres=0.25, iou_thres=0.5),
                [[16, 16, 24, 24, 0.75, 5], [16, 16, 24, 24, 0.75, 2]]))
    out.append(("equal_scores_across_classes_agnostic", _pred(rows), dict(conf_thres=0.25, iou_thres=0.5, agnostic=True),
                [[16, 16, 24, 24, 0.75, 5]]))

    # 12. IoU exactly equal to the threshold ACROSS the class offset (agnostic off): class 1 at both anchors, offset 7680 exactly
    #     representable next to these coordinates (7680 + 16 etc. are integers): the shifted boxes keep IoU = 0.5 exactly -> kept.
    rows = [(8, 8, 16, 16, 1, 0.9), (8, 4, 16, 8, 1, 0.8)]
    out.append(("iou_equal_threshold_with_class_offset", _pred(rows), dict(conf_thres=0.25, iou_thres=0.5),
                [[0, 0, 16, 16, 0.9, 1], [0, 0, 16, 8, 0.8, 1]]))
    return out
