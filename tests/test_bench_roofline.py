"""bench.roofline_from_profile on the committed per-op tables (CPU): the block reproduces by hand from the committed per-launch
table, and families that tie with the dominant one are reported beside it."""
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _roof(tag, cfg):
    import bench

    rows = json.loads((ROOT / "profiles" / f"{tag}_per_op.json").read_text())
    return bench.roofline_from_profile(rows, cfg["batch"], cfg), rows


def test_roofline_block_reproduces_from_the_committed_rows():
    (roof, kernels), rows = _roof("r06", {"scale": "n", "imgsz": 640, "batch": 32, "dtype": "bf16"})
    fam = [r for r in rows if r["kernel"] == roof["kernel"]]
    assert len(fam) == roof["launches_per_step"] == len(roof["instances"])
    ms = sum(r["ms"] for r in fam)
    work = sum(r["flops"] if roof["bound"] == "mfma" else r["bytes"] for r in fam)
    frac = work / (ms * 1e-3) / (1e12 if roof["bound"] == "mfma" else 1e9) / roof["peak"]
    assert abs(frac - roof["frac"]) < 1e-3
    # kernel-only figure: the instances' work over their committed rocprofv3 durations
    us = sum(i["us_rocprof"] for i in roof["instances"])
    w = sum(i["GFLOP"] * 1e9 if roof["bound"] == "mfma" else i["algorithmic_bytes"] for i in roof["instances"])
    assert abs(w / (us * 1e-6) / (1e12 if roof["bound"] == "mfma" else 1e9) / roof["peak"] - roof["frac_rocprof"]) < 1e-3
    assert roof["traffic"] == round(sum(i["traffic"] for i in roof["instances"]) / len(roof["instances"]))
    assert set(kernels) == {r["kernel"] for r in rows}


def test_families_that_tie_with_the_dominant_one_ride_along():
    (roof, kernels), _ = _roof("r06", {"scale": "n", "imgsz": 640, "batch": 32, "dtype": "bf16"})
    lead = kernels[roof["kernel"]]["ms"]
    near = {k for k, v in kernels.items() if k != roof["kernel"] and v["ms"] >= 0.95 * lead}
    assert {a["kernel"] for a in roof.get("also", [])} == near
    # YOLO11n: the fused C3k2 launches and the stride-2 convs are within 5 % of each other -- whichever leads, both are in the line
    assert {"c3k2_fused_mfma", "conv3x3s2_mfma"} <= {roof["kernel"]} | near
    for a in roof.get("also", []):
        assert a["instances"] and "frac_rocprof" in a and a["frac"] > 0
