#!/usr/bin/env python3
"""Measures (on the GPU box) how far the HIP network is from the oracle in its two modes -- fp32 (the reference-pinned
restatement) and bf16-storage (same arithmetic, tensors rounded where the HIP path stores bf16) -- so that the test
tolerances can be stated from measurements.  Writes profiles/<tag>_parity_probe.json."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from yololite_amd import _lib  # noqa: E402

_lib.apply_runtime_knobs()
from oracle import yolo11_oracle as oracle  # noqa: E402
from oracle.parity_gate import parity_gate  # noqa: E402
from yololite_amd.nn import arch  # noqa: E402
from yololite_amd.nn.engine import HipEngine  # noqa: E402
from yololite_amd.nn.synth import synth_state_dict  # noqa: E402
from yololite_amd.utils import ops as hops  # noqa: E402


def strides_per_anchor(h, w):
    return torch.tensor([float(s) for s in (8, 16, 32) for _ in range((h // s) * (w // s))])


def main(tag="r02"):
    dev = torch.device("cuda:0")
    g = np.load(ROOT / "tests" / "golden" / "golden_v1.npz")
    torch.set_num_threads(8)
    out = {}
    cases = [("n", 64, 64, 0, "net_n_64x64"), ("n", 96, 128, 1, "net_n_96x128"), ("s", 64, 64, 2, "net_s_64x64"),
             ("m", 64, 64, 3, "net_m_64x64"), ("x", 64, 64, 4, "net_x_64x64"), ("n", 640, 640, 0, "net_n_640")]
    for scale, h, w, seed, key in cases:
        sd = synth_state_dict(scale, seed, cls_bias=list(g[key + "_bias"]), cls_gain=8.0)
        if key == "net_n_640":
            x = torch.rand(4, 3, 640, 640, generator=torch.Generator().manual_seed(300))
        else:
            x = torch.from_numpy(g[key + "_x"])
        specs, _ = arch.parse_model(arch.yolo11_dict(scale))
        eng = HipEngine.from_model(specs, sd, x.shape[0], h, w, dev, keep_logits=True)
        y = eng.forward(x.to(dev)).cpu()
        lv = [t.cpu() for t in eng.raw_levels()]
        det = [d.cpu() for d in hops.non_max_suppression(y.to(dev), 0.25, 0.7)]
        st = strides_per_anchor(h, w)
        rec = {}
        for mode in ("fp32", "bf16"):
            with torch.inference_mode():
                yr, raw = oracle.forward(sd, x, scale, storage=mode)
            ds = (y[:, 4:] - yr[:, 4:]).abs()
            ref = oracle.non_max_suppression(yr, 0.25, 0.7)
            rec[mode] = dict(score_max=round(float(ds.max()), 5), score_rms=round(float(ds.pow(2).mean().sqrt()), 6),
                             box_max_strides=round(float(((y[:, :4] - yr[:, :4]).abs() / st).max()), 4),
                             level_rel_l2=[round(float((a - b).norm() / b.norm()), 5) for a, b in zip(lv, raw)],
                             cls_logit_max=[round(float((a[:, 64:] - b[:, 64:]).abs().max()), 4) for a, b in zip(lv, raw)],
                             gate=parity_gate(ref, det, 0.25, 0.7, band=0.08, ref_y=yr))
        out[key] = rec
        print(key, json.dumps(rec), flush=True)
        del eng
    p = ROOT / "profiles" / f"{tag}_parity_probe.json"
    p.write_text(json.dumps(out, indent=1))
    print("wrote", p)


if __name__ == "__main__":
    main(*sys.argv[1:])
