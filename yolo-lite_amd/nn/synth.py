"""Deterministic synthetic weights for YOLO11 (no checkpoints are available offline).

Every tensor is a pure function of (key, shape, seed): the same call gives the same bytes in the
build container (where the reference is imported to make golden vectors) and on the GPU box.
BN statistics are randomised so that Conv+BN folding is exercised.  Keys and shapes follow the
reference state dict (``model.<i>...{conv.weight,bn.*}``, ``model.23.dfl.conv.weight``).
"""
from __future__ import annotations

import hashlib
import math
from typing import Dict, Optional, Sequence

import torch

from .arch import ConvUnit, conv_units, parse_model, yolo11_dict


def _gen(key: str, seed: int) -> torch.Generator:
    h = hashlib.sha256(f"{seed}:{key}".encode()).digest()
    g = torch.Generator(device="cpu")
    g.manual_seed(int.from_bytes(h[:8], "little") & 0x7FFFFFFFFFFFFFFF)
    return g


def _randn(key, seed, shape):
    return torch.randn(shape, generator=_gen(key, seed), dtype=torch.float32)


def _rand(key, seed, shape, lo, hi):
    return torch.rand(shape, generator=_gen(key, seed), dtype=torch.float32) * (hi - lo) + lo


def synth_state_dict(scale: str = "n", seed: int = 0, nc: int = 80,
                     cls_bias: Optional[Sequence[float]] = None, cls_gain: float = 8.0,
                     box_gain: float = 6.0, model_dict: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """Random-init fp32 state dict of the YOLO11 architecture.

    ``cls_bias``: optional per-level bias for the final class 1x1 convs (calibration, see
    :func:`calibrate_cls_bias`); default is the reference ``Detect.bias_init`` value
    ``log(5 / nc / (640 / stride)**2)`` (reference ``head.py:128-139``).
    ``cls_gain`` / ``box_gain`` multiply the final class / box conv weights (spread the logits so
    that scores and DFL distributions vary across anchors).  Gains of the BN'd convs are chosen so
    activations stay O(1) through every scale n..x: sqrt(2/fan_in) before SiLU, 0.8/sqrt(fan_in)
    for linear convs and 0.5/sqrt(fan_in) for convs whose output is added to a residual.
    """
    d = model_dict or yolo11_dict(scale, nc)
    specs, _ = parse_model(d)
    sd: Dict[str, torch.Tensor] = {}
    det = specs[-1]
    strides = (8.0, 16.0, 32.0)
    for u in conv_units(specs):
        fan_in = (u.c1 // u.g) * u.k * u.k
        if u.bn:
            n = u.name
            is_res = (".m." in n and n.endswith(".cv2")) or n.endswith("attn.proj") or n.endswith("ffn.1")
            std = (0.5 if is_res else (math.sqrt(2.0) if u.act else 0.8)) / math.sqrt(fan_in)
            sd[u.name + ".conv.weight"] = _randn(u.name + ".conv.weight", seed, (u.c2, u.c1 // u.g, u.k, u.k)) * std
            sd[u.name + ".bn.weight"] = _rand(u.name + ".bn.weight", seed, (u.c2,), 0.6, 1.4)
            sd[u.name + ".bn.bias"] = _randn(u.name + ".bn.bias", seed, (u.c2,)) * 0.2
            sd[u.name + ".bn.running_mean"] = _randn(u.name + ".bn.running_mean", seed, (u.c2,)) * 0.2
            sd[u.name + ".bn.running_var"] = _rand(u.name + ".bn.running_var", seed, (u.c2,), 0.5, 1.5)
            sd[u.name + ".bn.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        else:
            std = math.sqrt(1.0 / fan_in)
            w = _randn(u.name + ".weight", seed, (u.c2, u.c1, u.k, u.k)) * std
            is_cls = ".cv3." in u.name
            lvl = int(u.name.split(".")[3])
            if is_cls:
                w = w * cls_gain
                b0 = cls_bias[lvl] if cls_bias is not None else math.log(5 / det.nc / (640 / strides[lvl]) ** 2)
                b = torch.full((u.c2,), float(b0), dtype=torch.float32)
            else:
                w = w * box_gain
                b = torch.full((u.c2,), 1.0, dtype=torch.float32)
            sd[u.name + ".weight"] = w
            sd[u.name + ".bias"] = b
    sd[f"model.{det.i}.dfl.conv.weight"] = torch.arange(16, dtype=torch.float32).view(1, 16, 1, 1)
    return sd


def calibrate_cls_bias(level_max_logits: Sequence[torch.Tensor], target_frac: float = 0.015,
                       conf: float = 0.25) -> list:
    """Per-level class bias such that about ``target_frac`` of anchors clear ``conf``.

    ``level_max_logits[l]`` holds, for level ``l``, the per-anchor maximum class logit computed with
    a zero class bias.  The (1 - target_frac) quantile over all levels is mapped to logit(conf).
    """
    allv = torch.cat([t.flatten().float().cpu() for t in level_max_logits])
    q = torch.quantile(allv, 1.0 - target_frac).item()
    shift = math.log(conf / (1.0 - conf)) - q
    return [shift for _ in level_max_logits]



def synth_images(batch: int, h: int, w: int, seed: int = 0) -> torch.Tensor:
    """Deterministic synthetic images in [0, 1], (B, 3, H, W) fp32, with SPATIAL STRUCTURE: smooth colour fields and blocks at
    several scales plus a texture whose amplitude varies from region to region.  White noise (``torch.rand``) looks the same
    everywhere, so a random-init network answers with nearly the same class logits at every anchor and "1.5 % of the anchors
    above the threshold" is a razor-thin slice of a very narrow distribution -- the box count then swings by tens of
    percent with the last bit of rounding.  With structure the per-anchor maxima spread out, as they do on photographs."""
    import torch.nn.functional as F

    g = _gen(f"images:{batch}x{h}x{w}", seed)

    def field(cell: int, mode: str, ch: int = 3) -> torch.Tensor:
        hh, ww = max(1, -(-h // cell)) + 1, max(1, -(-w // cell)) + 1
        f = torch.rand((batch, ch, hh, ww), generator=g) * 2.0 - 1.0
        kw = dict(mode=mode) if mode == "nearest" else dict(mode=mode, align_corners=False)
        return F.interpolate(f, size=(hh * cell, ww * cell), **kw)[:, :, :h, :w]

    x = 0.5 + 0.30 * field(256, "bilinear") + 0.22 * field(64, "nearest") + 0.14 * field(16, "nearest") + 0.08 * field(4, "nearest")
    amp = field(48, "nearest", 1).abs()  # texture strength per region: some flat, some busy
    x = x + 0.25 * amp * (torch.rand((batch, 3, h, w), generator=g) * 2.0 - 1.0)
    return x.clamp_(0.0, 1.0)


def plant_object_head(sd: Dict[str, torch.Tensor], feats: Sequence[torch.Tensor], per_image: int = 96, alpha: float = 12.0,
                      margin: float = 3.0, ridge: float = 0.05, ridge_samples: float = 0.25, seed: int = 0,
                      det_index: int = 23) -> Dict[str, torch.Tensor]:
    """A synthetic class head with OBJECTS in it: a copy of ``sd`` whose last class convs (``model.<det>.cv3.<l>.2``, reference
    head.py:47 ``nn.Conv2d(c3, nc, 1)``) are matched filters for ~``per_image`` planted anchors per image.

    ``feats[l]``: (B, c3, H_l, W_l) fp32, the INPUT of level l's last class conv on the images the head is built for (random-init
    features of a fixed synthetic batch).  Planted anchors are drawn per image and level (in proportion to the level's anchors, at
    least two pixels apart) and dealt to the classes round-robin.  Random-init features of noise images are strongly correlated
    from anchor to anchor (a few dominant directions), so the filters are WHITENED matched filters: with mu, S = mean and
    covariance of the level's features over all anchors and R = S + r^2 * mean(diag S) * I, r = max(ridge, ridge_samples * sqrt(c3 / anchors)),
        w_c = alpha * sum_k R^-1 d_k / (d_k^T R^-1 d_k),  d_k = p_k - mu,      b_c = margin - alpha - w_c . mu
    A planted anchor answers with logit ~``margin`` (score 0.95 for 3), an unrelated anchor with ``margin - alpha`` plus cross-talk
    of std ~alpha * sqrt(prototypes per class / c3): the per-anchor best score is BIMODAL, as a trained detector's is.  With the
    plain random head (``calibrate_cls_bias``) white-noise images put every anchor's best score within a fraction of a logit of
    the threshold, so the box COUNT measured the last bit of the number format and a post-NMS parity gate could not be failed by a
    wrong kernel (round-5 review, weak #1).  ``ridge`` bounds the amplification of feature noise (the storage format's rounding)
    along low-variance directions.  Deterministic in (sd, feats, seed)."""
    out = dict(sd)
    g = _gen("plant", seed)
    B = feats[0].shape[0]
    A = sum(f.shape[2] * f.shape[3] for f in feats)
    k_class = 0
    for l, f in enumerate(feats):
        f = f.double().cpu()
        _, c3, H, W = f.shape
        key = f"model.{det_index}.cv3.{l}.2"
        nc = sd[key + ".weight"].shape[0]
        flat = f.permute(1, 0, 2, 3).reshape(c3, -1)          # (c3, B * H * W)
        mu = flat.mean(1)
        dev = flat - mu[:, None]
        S = dev @ dev.T / max(1, dev.shape[1] - 1)
        # (a level with few anchors per feature dimension -- P5 of a small batch -- gets a stronger ridge: its sample covariance
        # is poorly conditioned and the filter would amplify the format's rounding noise along directions it merely under-sampled)
        rl = max(ridge, ridge_samples * math.sqrt(c3 / max(1, dev.shape[1])))
        R = S + (rl ** 2) * S.diagonal().mean() * torch.eye(c3, dtype=torch.float64)
        Rinv = torch.linalg.inv(R)
        n_l = max(1, int(round(per_image * H * W / A)))
        ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        lattice = ((ys % 3 == 1) & (xs % 3 == 1)).flatten().nonzero().flatten() if H >= 3 and W >= 3 else torch.arange(H * W)
        w = torch.zeros(nc, c3, dtype=torch.float64)
        for b in range(B):
            pick = lattice[torch.randperm(lattice.numel(), generator=g)[: min(n_l, lattice.numel())]]
            for p in pick.tolist():
                d = f[b, :, p // W, p % W] - mu
                rd = Rinv @ d
                w[k_class % nc] += alpha * rd / (d @ rd).clamp(min=1e-30)
                k_class += 1
        out[key + ".weight"] = w.float().view(nc, c3, 1, 1).contiguous()
        out[key + ".bias"] = ((margin - alpha) - w @ mu).float()
    return out
