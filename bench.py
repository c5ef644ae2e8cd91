#!/usr/bin/env python3
"""bench.py -- images/sec end-to-end (network + NMS), YOLO11n 640x640 batch 32 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic images: HIP network forward (from the uint8 input to
the Detect level maps) + fused Detect decode + GPU NMS, replayed as ONE hipGraph.

N = 1: the inputs are resident in HBM; ``value`` = steps pipelined over ``--streams`` HIP streams (one workspace slot
each); ``value_sync_per_iteration`` = the same step with the stream synchronised after every iteration (SURVEY 8(d)'s
literal definition); ``api_images_per_sec`` = the same batches through ``YOLOLite.predict``.

N > 1 (BASELINE.json config 4): rank 0 holds the uint8 GLOBAL batch (N x 32 images) in its HBM; every
step it scatters the 32-image slices over RCCL (issued ``--streams`` steps ahead, hidden behind the batches in
flight), every rank runs the one-graph hot path on its slice, and rank 0 gathers the packed ``(count, 300 x 6)`` records
of every step: ``value`` = all ranks' images / max-over-ranks time of that loop.  ``value_resident`` = the replica figure
without any per-step collective (every rank's shards resident in its own HBM, one gather at the end), timed separately.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _argv_gpus(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def self_launch(n):
    """``python bench.py --gpus N`` with no external launcher: this process -- which has made no GPU call and never
    will -- starts N fresh children of this file, one rank per GPU (yololite_amd/launch.py launch_ranks: RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* as ``torch.distributed.run`` would set them; never an exec from a process that touched the GPU),
    relays rank 0's single JSON line on stdout, every rank's stderr on stderr (prefixed), and exits non-zero -- ending the other
    ranks -- as soon as any rank fails."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_yl_launch", str(ROOT / "yolo-lite_amd" / "launch.py"))  # (not via the package:
    mod = importlib.util.module_from_spec(spec)                                                              # nothing else is imported)
    spec.loader.exec_module(mod)
    rc, lines = mod.launch_ranks(n, [sys.executable, str(Path(__file__).resolve())] + sys.argv[1:])
    record = [ln for ln in lines if ln.lstrip().startswith("{")]
    if rc == 0 and len(record) != 1:
        sys.stderr.write(f"[bench] expected ONE JSON line from rank 0, got {len(record)}\n")
        rc = 1
    if rc == 0:
        sys.stdout.write(record[0])
        sys.stdout.flush()
    else:
        sys.stderr.write(f"[bench] --gpus {n} FAILED (status {rc}); no record printed\n")
    raise SystemExit(rc)


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _argv_gpus(sys.argv) > 1:
    self_launch(_argv_gpus(sys.argv))

from yololite_amd import _lib as yl_lib  # noqa: E402

# bench.py is its own launcher: the HIP runtime knobs the hot path depends on are set here, explicitly, before the
# first GPU call of the process (yololite_amd/_lib.py documents them); the JSON line records whether that was in time
# and whether two graph branches really overlap on this runtime ("runtime").
# Hardware queues: exactly one per batch in flight (fewer streams than queues is as bad as more -- tools/queue_sweep.sh on
# one MI355X, queues / streams -> images/s).  Single-GPU path: 8 / 8 -> 43.6-43.9 k, 16 / 16 -> 46.5-47.1 k, 20 / 20 ->
# 46.9-47.1 k (and 42.7-43.0 k against 40.0-40.7 k in a 20-step run, which 20 batches in flight cover in one round), 20 / 16
# -> 37.6 k, 16 / 8 -> 34.5 k, 24 / 24 -> 43.6 k, 32 / 32 -> 32.7 k.  A rank of the sharded path (RCCL brings streams of
# its own; world size 1 rehearsal): 8 / 7 -> 47.0 k, 8 / 8 -> 32.3 k, 16 / 7 -> 39.6 k, 16 / 14 -> 32.6 k, 20 / 18 -> 35.0 k.
SHARDED_ENV = int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("YL_BENCH_FORCE_SHARDED", "0") == "1"
SINGLE_IN_FLIGHT = 20
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8" if SHARDED_ENV else str(SINGLE_IN_FLIGHT))
KNOBS = yl_lib.apply_runtime_knobs()
# stdout carries exactly ONE line, the JSON record: libraries that print there (RCCL's version banner does, at communicator
# creation) are sent to stderr at the descriptor level; the record is written to the saved descriptor at the end.
_REAL_STDOUT = os.dup(1)
os.dup2(2, 1)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP8_TFLOPS = 5000.0     # dense e4m3 MFMA peak (MI355X_MICROARCH.md); the sparse figure is never used
PEAK_HBM_GBS = 8000.0       # HBM3E spec (same table)
# SURVEY.md section 8(d), per image at 640x640 (both scale with the image area): conv GFLOP = 2 * MAC over all nn.Conv2d
# calls; activation elements in + out of every layer (M), stored as bf16
CONV_GFLOP_AT_640 = {"n": 6.4811, "s": 21.4669, "x": 194.9038}
ACT_MELEM_AT_640 = {"n": 40.83, "s": 74.89, "x": 312.05}


def build_model(scale, device, batch, h, w, x_calib, target_frac=0.015, fp8=False, planted=0):
    """Random-init YOLO11 with the class bias calibrated on-device so ~1.5 % of anchors pass conf 0.25.
    fp8: the dense 3x3 convs run in e4m3 (weights + activations, fp32 accumulate), activation scales calibrated on
    ``x_calib`` (BASELINE.json config 5).
    planted > 0: instead of the calibrated random head, a head with that many OBJECTS per image planted in ``x_calib``
    (nn/synth.py plant_object_head: whitened matched filters on the features the last class convs see): the per-anchor best
    score is bimodal, as a trained detector's is, so the post-NMS parity gate measures the kernels and not the last bit of the
    number format at a threshold every anchor sits on (round-5 review, weak #1).  The objects exist in ``x_calib``'s images only."""
    from yololite_amd.nn import arch
    from yololite_amd.nn.engine import HipEngine
    from yololite_amd.nn.synth import calibrate_cls_bias, plant_object_head, synth_state_dict

    specs, _ = arch.parse_model(arch.yolo11_dict(scale))
    sd0 = synth_state_dict(scale, 0, cls_bias=[0.0, 0.0, 0.0])
    if planted:
        e0 = HipEngine.from_model(specs, sd0, batch, h, w, device, keep_layers=True)  # (unfused: the last class convs' inputs exist)
        e0.forward(x_calib, decode=False)
        torch.cuda.synchronize()
        det_i = specs[-1].i
        feats = []
        for lvl in range(3):
            op = next(o for o in e0.prog.ops if o.get("name", "") == f"model.{det_i}.cv3.{lvl}.2")
            feats.append(e0.view_tensor(op["inp"]).cpu())
        del e0
        torch.cuda.empty_cache()
        kw = dict(planted) if isinstance(planted, dict) else dict(per_image=int(planted))  # (a dict: plant_object_head's keywords, probes)
        sd = plant_object_head(sd0, feats, det_index=det_i, **kw)
        del feats
        return specs, sd, HipEngine.from_model(specs, sd, batch, h, w, device, fp8_calib=x_calib if fp8 else None)
    eng = HipEngine.from_model(specs, sd0, batch, h, w, device)
    eng.forward(x_calib)
    torch.cuda.synchronize()
    mx = [r.amax(1).float().cpu() for r in eng.raw_cls_levels()]
    bias = calibrate_cls_bias(mx, target_frac)
    del eng
    sd = synth_state_dict(scale, 0, cls_bias=bias)
    torch.cuda.empty_cache()
    return specs, sd, HipEngine.from_model(specs, sd, batch, h, w, device, fp8_calib=x_calib if fp8 else None)


def config5_target_frac(imgsz: int, per_image: float = 0.0) -> float:
    """Fraction of anchors the synthetic head lets over conf 0.25: 1.5 % at 640x640 (126 of 8 400 anchors, ~100 detections
    after NMS: the headline configuration).  Larger images keep the candidate COUNT per image, not the fraction
    (``per_image``, default C5_CANDIDATES_PER_IMAGE): at 1280x1280 (33 600 anchors) 1.5 % of white-noise anchors meant 500
    scattered candidates and every image cut at max_det = 300 (round-3 review)."""
    anchors = sum((imgsz // s) ** 2 for s in (8, 16, 32))
    if anchors <= 8400:
        return 0.015
    return min(0.015, (per_image or C5_CANDIDATES_PER_IMAGE) / anchors)


# BASELINE.json config 5 (YOLO11x, 1280x1280): candidates per image over conf 0.25 -- the headline configuration's own count, so
# that an image keeps ~60-140 boxes after NMS and none is cut at max_det
C5_CANDIDATES_PER_IMAGE = float(os.environ.get("YL_C5_PER_IMAGE", "126"))


# config 5's synthetic head: objects planted per image (build_model(planted=...)) and the fixed post-NMS gate its e4m3 program must
# pass against the fp32 oracle (parity_gate keywords; tests/test_gpu_fp8.py asserts the same)
# (objects per image = C5_PLANTED_TOTAL / batch, at most 96: a class filter is the sum of its prototypes' matched filters, and the
# cross-talk on unrelated anchors grows with sqrt(prototypes per class / feature dimensions) -- ~190 prototypes per level over 80
# classes keep the background four logits under the threshold; 8 x 96 of them put 650 background anchors per image over it)
C5_PLANTED_TOTAL = int(os.environ.get("YL_C5_PLANTED_TOTAL", "192"))


def c5_planted(batch: int) -> int:
    return max(8, min(96, C5_PLANTED_TOTAL // max(1, batch)))
C5_GATE = dict(band=0.2, max_band=0.6, p99_band=0.2, min_matched_frac=0.9, max_explained_share=0.35, max_count_bias=0.25)


def config5_images(batch: int, imgsz: int, seed: int = 0):
    """Inputs of config 5: white noise, like the headline run's.  Measured alternative (tools/synth_input_probe.py): images
    with spatial structure (nn/synth.py::synth_images) spread YOLO11x's per-anchor class maxima from 0.3 to 1.7 logit between
    the 98.5 % and 99.6 % quantiles -- the box COUNT then agrees between fp32, bf16 and e4m3 (35 / 36 / 43 boxes) where white
    noise gives 60 / 93 / 135 -- but the wider logits also show e4m3's real error (max |dconf| 0.15-0.19, half of the boxes
    paired): white noise is the gentler input for the scores and the harsher one for the count.  Kept: white noise, with the
    count bias reported and the e4m3 FORMAT's own count bias (CPU emulation) beside it."""
    return torch.rand(batch, 3, imgsz, imgsz, generator=torch.Generator().manual_seed(seed))


def cpu_baseline(sd, scale, h, w, conf, iou, sample_b=32, iters=3, eng=None, dev=None, max_det=300, band=0.08):
    """The oracle (fp32 eager PyTorch restatement of the reference path, unfused Conv->BN->SiLU +
    python-loop NMS) timed on this host's cores: threads = min(8, cores-1) as the reference sets
    (utils/__init__.py:32, torch_utils.py:168-169).  kind = "port".  Its last outputs double as the parity gate of the
    measured HIP path (oracle/parity_gate.py): the HIP detections against the fp32 restatement's with the stated bf16
    FORMAT tolerance, and -- the yardstick -- the same gate applied to the oracle's own bf16-storage mode (fp32
    arithmetic, tensors rounded where the HIP path stores bf16; no kernel involved): the HIP path may be as far from
    the reference as the format itself is, and no farther (``format_yardstick``)."""
    from oracle import yolo11_oracle as oracle
    from oracle.parity_gate import format_yardstick_ok, parity_gate

    cores = os.cpu_count() or 1
    threads = min(8, max(1, cores - 1))
    torch.set_num_threads(threads)
    x = torch.rand(sample_b, 3, h, w, generator=torch.Generator().manual_seed(0))
    with torch.inference_mode():
        y, _ = oracle.forward(sd, x, scale)  # warm-up
        oracle.non_max_suppression(y, conf, iou)
        t0 = time.time()
        for _ in range(iters):
            y, _ = oracle.forward(sd, x, scale)
            oracle.non_max_suppression(y, conf, iou)
        dt = time.time() - t0
        ref = oracle.non_max_suppression(y, conf, iou, max_det=max_det)
    out = {"value": round(sample_b * iters / dt, 2), "unit": "images/sec", "cores": threads, "kind": "port",
           "host_cores": cores,
           "sample": f"{iters} iterations of batch {sample_b} at {h}x{w} after 1 warm-up, fp32 oracle forward + NMS"}
    if eng is not None:  # the checker's last outputs double as the parity gate of the measured HIP path
        xb = torch.zeros(eng.batch, 3, h, w)
        xb[:sample_b] = x
        o, c = eng.detect(xb.to(dev), conf, iou, max_det=max_det, graph=False)
        torch.cuda.synchronize(dev)
        got = [o[i, : int(c[i])].cpu() for i in range(sample_b)]
        out["parity_vs_oracle"] = parity_gate(ref, got, conf, iou, band=band, ref_y=y)
        with torch.inference_mode():
            y16, _ = oracle.forward(sd, x, scale, storage="bf16")
            det16 = oracle.non_max_suppression(y16, conf, iou, max_det=max_det)
        out["format_error_bf16_storage"] = parity_gate(ref, det16, conf, iou, band=band, ref_y=y)
        out["format_yardstick"] = format_yardstick_ok(out["parity_vs_oracle"], out["format_error_bf16_storage"], band)
    return out


def short_config(scale, batch, imgsz, dtype, streams, steps, dev, conf=0.25, iou=0.7, max_det=300, stream_pool=None):
    """A short pipelined run of another BASELINE.json configuration (3: YOLO11s bf16 batch 64; 5: YOLO11x e4m3 1280x1280
    batch 8) inside the default bench, so that the driver's own run carries a number for them.  Same step, same timing
    bracket as `value`; fewer steps.  For fp8 the e4m3 program's detections are gated against the bf16 program's."""
    x = config5_images(batch, imgsz).to(dev)
    tfrac = config5_target_frac(imgsz)  # (0.015 at 640x640: the headline configuration's own calibration)
    # config 5 (e4m3): the head with planted objects (C5_PLANTED per image; round-5 review: the calibrated random head puts every
    # anchor's best score at the threshold, where a post-NMS gate measures the format's last bit and cannot fail a wrong kernel)
    planted = c5_planted(batch) if dtype == "fp8" else 0
    specs, sd, eng = build_model(scale, dev, batch, imgsz, imgsz, x, target_frac=tfrac, fp8=dtype == "fp8", planted=planted)
    x8 = (x * 255).to(torch.uint8)
    del x
    S = streams
    # the main run's streams when given: new ones would share the 20 hardware queues with the 40 streams this process
    # already owns (main loop + the API predictor's) unevenly (YOLO11s: 23.9 instead of 25.1 k images/s)
    st = list(stream_pool[:S]) if stream_pool is not None and len(stream_pool) >= S else [torch.cuda.Stream(device=dev) for _ in range(S)]
    if planted:
        # the planted objects live in x8's images: the other slots hold the same images in another order (separate buffers in HBM, the
        # same NMS load in every slot; fresh noise images would carry no object at all)
        xs = [x8] + [x8[torch.randperm(batch, generator=torch.Generator().manual_seed(2000 + j)).to(dev)].contiguous() for j in range(1, S)]
    else:
        xs = [x8] + [torch.randint(0, 256, (batch, 3, imgsz, imgsz), generator=torch.Generator().manual_seed(2000 + j),
                                   dtype=torch.uint8).to(dev) for j in range(1, S)]

    def step(k):
        with torch.cuda.stream(st[k % S]):
            return eng.detect(xs[k % S], conf, iou, max_det=max_det, graph=True, slot=k % S)

    for k in range(2 * S):
        step(k)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps):
        out, cnt = step(k)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    res = {"workload": f"YOLO11{scale} {dtype}, {imgsz}x{imgsz} batch={batch}, {S} batches in flight", "value": round(batch * steps / dt, 1),
           "unit": "images/sec", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
           "mean_detections_per_image": round(cnt.float().mean().item(), 1), "images_cut_at_max_det": int((cnt >= max_det).sum().item()),
           "candidate_fraction_target": round(tfrac, 5)}
    try:  # the best 3x3 conv of this configuration against its MFMA roof (in-sequence HIP events, one batch in flight)
        rows = eng.profile_detect(x8, conf, iou, max_det=max_det, iters=10)
        best, tot_fl = None, 0.0
        for r in rows:
            tot_fl += r["flops"]
            if r["kernel"].startswith("conv3x3") and r["ms"] > 0:
                f = r["flops"] / (r["ms"] * 1e-3) / 1e12 / (PEAK_FP8_TFLOPS if "_f8_" in r["kernel"] else PEAK_BF16_TFLOPS)
                if best is None or f > best[1]:
                    best = (r["name"], f, r["ms"], r["kernel"])
        if best:
            res["best_conv3x3_layer"] = {"layer": best[0], "kernel": best[3], "mfma_frac": round(best[1], 4), "us": round(best[2] * 1e3, 2)}
        res["e2e_tflops"] = round(tot_fl / (dt / steps) / 1e12, 1)
    except Exception as e:  # (a diagnostic: never in the way of the number)
        res["best_conv3x3_layer"] = {"error": repr(e)}
    if dtype == "fp8":
        from oracle.parity_gate import parity_gate
        from yololite_amd.nn.engine import HipEngine

        gate_max_det = 3000  # (lifted: the cut at max_det = 300 must not create one-sided boxes of its own)
        o8, c8 = eng.detect(x8, conf, iou, max_det=gate_max_det, graph=False)
        e16 = HipEngine.from_model(specs, sd, batch, imgsz, imgsz, dev)
        o16, c16 = e16.detect(x8, conf, iou, max_det=gate_max_det, graph=False)
        torch.cuda.synchronize(dev)
        # THE FIXED GATE of config 5 (C5_GATE, round 6): boxes paired by IoU >= 0.9, classes of paired boxes bit-exact, nothing
        # unexplained, >= 90 % of the reference's boxes paired, one-sided share <= 0.35, |det / ref - 1| <= 0.25, |dconf| p99 <= 0.2
        # and max <= 0.6 (the two score bands are stated from the HIP measurement on this head -- 1280x1280, 2 and 8 images: p99
        # 0.11-0.16, max 0.19-0.57, the 0.57 at ONE of 207 boxes; the bf16 program: p99 0.06, max 0.07-0.15 -- a whitened matched
        # filter of gain 12 turns the e4m3 rounding of a few feature directions into logit errors of up to 2.7; the CPU emulation
        # of the same roundings has tails of the same size at other anchors: 0.17-0.25)
        # (both programs' pre-NMS tensors go in: a paired box whose classes differ because two classes tie within the band at that
        # anchor is counted as a near-tie, not as a class mismatch -- SURVEY 8(d))
        y16_all = e16.forward(x8, graph=False).float().cpu()
        y8_all = eng.forward(x8, graph=False).float().cpu()
        g = parity_gate([o16[i, : int(c16[i])].cpu() for i in range(batch)], [o8[i, : int(c8[i])].cpu() for i in range(batch)],
                        conf, iou, ref_y=y16_all, got_y=y8_all, **C5_GATE)
        del y16_all
        keys = ("ref", "det", "matched", "matched_frac", "class_mismatch", "unexplained", "max_dconf", "p99_dconf", "explained_share",
                "max_explained_share", "count_bias", "max_count_bias", "ok")
        res["head"] = f"planted objects, {planted} per image (nn/synth.py plant_object_head)"
        res["fp8_vs_bf16"] = {k: g.get(k) for k in keys}
        # ... and against the REFERENCE arithmetic: the fp32 oracle on the first two images (one host forward each, ~2 s), the
        # stated e4m3 tolerance (score band 0.12 = 1.5 x the bf16 band), >= 85 % of the reference's boxes paired, classes of
        # paired boxes exact, nothing unexplained.  max_det is lifted so that the cut at 300 does not create one-sided
        # boxes of its own.  Round 4: the head is calibrated to ~130 candidates per image (config5_target_frac; round 3's 1.5 %
        # of 33 600 anchors cut every image at max_det), and the one-sided share is capped at 0.35 (was 0.6).
        from oracle import yolo11_oracle as oracle

        n_or = min(2, batch)
        o8g, c8g = o8, c8
        torch.set_num_threads(min(8, max(1, (os.cpu_count() or 2) - 1)))
        t_or = time.time()
        with torch.inference_mode():
            y32, _ = oracle.forward(sd, x8[:n_or].float().cpu() / 255.0, scale)
            ref = oracle.non_max_suppression(y32, conf, iou, max_det=gate_max_det)
        y8 = y8_all[:n_or]  # the e4m3 program's own pre-NMS tensor (near-tied arg-maxes)
        g2 = parity_gate(ref, [o8g[i, : int(c8g[i])].cpu() for i in range(n_or)], conf, iou, ref_y=y32, got_y=y8, **C5_GATE)
        res["fp8_vs_fp32_oracle"] = dict({k: g2.get(k) for k in keys}, images=n_or, oracle_seconds=round(time.time() - t_or, 1),
                                         near_threshold=g2["near_threshold"], nms_flip=g2["nms_flip"])
        # the FORMAT's own distance from fp32: the oracle with the same convs emulated in e4m3 on the CPU (no kernel involved);
        # the HIP e4m3 program may be as far from the reference as that, and no farther (format_yardstick_ok)
        from oracle.parity_gate import format_yardstick_ok

        with torch.inference_mode():
            y8e, _ = oracle.forward(sd, x8[:n_or].float().cpu() / 255.0, scale, storage="bf16", fp8_scales=eng.fp8_conv_scales())
            det8e = oracle.non_max_suppression(y8e, conf, iou, max_det=gate_max_det)
        g3 = parity_gate(ref, det8e, conf, iou, ref_y=y32, got_y=y8e, **C5_GATE)
        res["format_error_e4m3"] = dict({k: g3.get(k) for k in keys}, note="DIAGNOSTIC: the CPU emulation of the same roundings through the same gate")
        res["format_yardstick_e4m3"] = dict(format_yardstick_ok(g2, g3, C5_GATE["max_band"]), note="DIAGNOSTIC, not the pass criterion (that is fp8_vs_fp32_oracle.ok)")
        del e16
    else:
        # bf16 configurations: the HIP detections of the first two images against the fp32 oracle (one host forward, a few
        # seconds), the same gate and band as the main configuration's cpu_baseline leg
        try:
            from oracle import yolo11_oracle as oracle
            from oracle.parity_gate import parity_gate

            n_or = min(2, batch)
            og, cg = eng.detect(x8, conf, iou, max_det=3000, graph=False)
            torch.cuda.synchronize(dev)
            torch.set_num_threads(min(8, max(1, (os.cpu_count() or 2) - 1)))
            t_or = time.time()
            with torch.inference_mode():
                y32, _ = oracle.forward(sd, x8[:n_or].float().cpu() / 255.0, scale)
                ref = oracle.non_max_suppression(y32, conf, iou, max_det=3000)
            y_hip = eng.forward(x8, graph=False)[:n_or].float().cpu()  # the HIP path's own pre-NMS tensor (near-tied arg-maxes)
            g3 = parity_gate(ref, [og[i, : int(cg[i])].cpu() for i in range(n_or)], conf, iou, band=0.08, ref_y=y32, got_y=y_hip)
            res["vs_fp32_oracle"] = dict({k: g3[k] for k in ("ref", "det", "matched", "matched_frac", "class_mismatch", "unexplained",
                                                              "max_dconf", "explained_share", "max_explained_share", "count_bias", "argmax_flip", "ok")},
                                         images=n_or, oracle_seconds=round(time.time() - t_or, 1))
        except Exception as e:
            res["vs_fp32_oracle"] = {"error": repr(e)}
    del eng
    torch.cuda.empty_cache()
    return res


def per_launch_profile(cfg):
    """The newest committed per-launch table (profiles/*_per_launch.json, tools/summarize_profile.py: rocprofv3 kernel-trace duration
    and PMC FETCH_SIZE / WRITE_SIZE bytes of every launch of a step, matched to the per-op table by position) whose ``_config``
    equals the configuration being benched; None when there is none."""
    best = None
    for f in sorted((ROOT / "profiles").glob("*_per_launch.json")):
        try:
            t = json.loads(f.read_text())
        except ValueError:
            continue
        if t.get("_config") == cfg and t.get("rows"):
            best = (f.name, t)
    return best


def roofline_from_profile(rows, batch, cfg):
    """Aggregate per-op HIP-event timings by kernel; dominant kernel = largest total time."""
    agg = {}
    for r in rows:
        a = agg.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
        a["ms"] += r["ms"]
        a["flops"] += r["flops"]
        a["bytes"] += r["bytes"]
        a["launches"] += 1
    plp = per_launch_profile(cfg)

    def block(name, a):
        tf = a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else 0.0
        gbs = a["bytes"] / (a["ms"] * 1e-3) / 1e9 if a["ms"] > 0 else 0.0
        peak_tf = PEAK_FP8_TFLOPS if "_f8_" in name else PEAK_BF16_TFLOPS  # e4m3 kernels are priced against the fp8 peak
        mfma_frac, hbm_frac = tf / peak_tf, gbs / PEAK_HBM_GBS
        bound = "mfma" if mfma_frac >= hbm_frac else "hbm"
        roof = {"kernel": name, "bound": bound,
                "achieved": round(tf if bound == "mfma" else gbs, 2),
                "peak": peak_tf if bound == "mfma" else PEAK_HBM_GBS,
                "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                "frac": round(max(mfma_frac, hbm_frac), 4), "traffic": None,
                "launches_per_step": a["launches"], "avg_launch_us": round(a["ms"] * 1e3 / a["launches"], 2),
                "mfma_frac": round(mfma_frac, 4), "hbm_frac": round(hbm_frac, 4),
                "algorithmic_bytes_per_launch": round(a["bytes"] / a["launches"])}
        # `frac` divides by the IN-SEQUENCE HIP-event time of a launch (measured live in this run; it includes the ~2.6 us dispatch gap
        # in front of a dependent launch); the committed rocprofv3 kernel trace gives the kernel-only duration: both are stated
        roof["time_source"] = "in-sequence HIP events of this run (dispatch gap included)"
        # Traffic and the kernel-only duration come from the committed per-launch table: THE SAME LAUNCHES (matched by op name) whose
        # algorithmic work is divided above -- round 5 divided the stride-2 convs' bytes by the duration / PMC bytes of a template family
        # that also held the stride-1 convs.  Every instance is listed; frac_rocprof must reproduce from those rows by hand.
        if plp:
            fname, tab = plp
            by_name = {r["name"]: r for r in tab["rows"]}
            inst, miss = [], []
            for r in rows:
                if r["kernel"] != name:
                    continue
                p = by_name.get(r["name"])
                if p is None or "us_rocprof" not in p:
                    miss.append(r["name"])
                    continue
                work = r["flops"] if bound == "mfma" else r["bytes"]
                tr = (p.get("read_bytes", 0) + p.get("write_bytes", 0)) if "read_bytes" in p and "write_bytes" in p else None
                inst.append({"op": r["name"], "us_in_sequence": round(r["ms"] * 1e3, 2), "us_rocprof": p["us_rocprof"],
                             "algorithmic_bytes": round(r["bytes"]), "GFLOP": round(r["flops"] / 1e9, 3), "traffic": tr,
                             "frac_in_sequence": round(work / (r["ms"] * 1e-3) / (1e12 if bound == "mfma" else 1e9) / roof["peak"], 4),
                             "frac_rocprof": round(work / (p["us_rocprof"] * 1e-6) / (1e12 if bound == "mfma" else 1e9) / roof["peak"], 4)})
            if inst and not miss:
                us_sum = sum(i["us_rocprof"] for i in inst)
                work_sum = sum((i["GFLOP"] * 1e9) if bound == "mfma" else i["algorithmic_bytes"] for i in inst)
                roof["instances"] = inst
                roof["avg_launch_us_rocprof"] = round(us_sum / len(inst), 2)
                roof["frac_rocprof"] = round(work_sum / (us_sum * 1e-6) / (1e12 if bound == "mfma" else 1e9) / roof["peak"], 4)
                roof["frac_rocprof_note"] = ("sum of the instances' algorithmic work / sum of their committed rocprofv3 kernel-trace durations "
                                             f"(profiles/{fname}, kernel only, one batch in flight)")
                if all(i["traffic"] is not None for i in inst):
                    roof["traffic"] = round(sum(i["traffic"] for i in inst) / len(inst))
                    roof["traffic_unit"] = "HBM bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE), mean over the instances listed"
                    roof["traffic_source"] = f"profiles/{fname}: {tab.get('_source', '')}"
                    roof["traffic_over_algorithmic"] = round(roof["traffic"] / max(1, roof["algorithmic_bytes_per_launch"]), 3)
            elif miss:
                roof["traffic_note"] = f"profiles/{fname} has no row for {miss}: the program changed since that profile was taken"
        return roof

    ranked = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
    roof = block(*ranked[0])
    # Families within 5 % of the dominant one's time (YOLO11n: the three fused C3k2 launches and the four stride-2 convs are 122 us each,
    # and which of them leads changes from run to run): their blocks ride along, so the line never depends on a coin toss
    near = [kv for kv in ranked[1:] if kv[1]["ms"] >= 0.95 * ranked[0][1]["ms"]]
    if near:
        roof["also"] = [block(*kv) for kv in near]
        roof["also_note"] = "families whose in-sequence time is within 5 % of the dominant family's: same block, same sources"
    # best single 3x3 layer against the MFMA roof (the north-star target names the fused 3x3 conv)
    best = None
    for r in rows:
        if r["kernel"].startswith("conv3x3") and r["ms"] > 0:
            f = r["flops"] / (r["ms"] * 1e-3) / 1e12 / (PEAK_FP8_TFLOPS if "_f8_" in r["kernel"] else PEAK_BF16_TFLOPS)
            if best is None or f > best[1]:
                best = (r["name"], f, r["ms"])
    if best:
        roof["best_conv3x3_layer"] = {"layer": best[0], "mfma_frac": round(best[1], 4), "us": round(best[2] * 1e3, 2)}
    kernels = {k: {"ms": round(v["ms"], 4), "launches": v["launches"],
                   "TFLOPs": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2),
                   "GBs": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)} for k, v in agg.items()}
    return roof, kernels


def main():
    if os.environ.get("YL_BENCH_WATCHDOG"):  # development: dump every thread's stack if the run takes this many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["YL_BENCH_WATCHDOG"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500,
                    help="timed steps; with several batches in flight the first and last few steps run on a partly empty GPU, so short "
                         "runs read 2-3 %% low (50 steps: 42.7 k images/s, 1000 steps: 43.9 k on the same box)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--scale", default="n")
    ap.add_argument("--streams", type=int, default=0,
                    help="batches in flight: consecutive steps are issued round-robin on this many HIP streams "
                         "(one workspace slot each) so one batch's large kernels overlap another's small ones; 0 = 20, one per "
                         "hardware queue (GPU_MAX_HW_QUEUES = 20, set at the top of this file); sharded path: 8 queues, and 6, 7 and 8 batches "
                         "in flight are tried for min(--steps, 80) untimed steps each (a batch stream that has to share a queue with RCCL's "
                         "stream loses a third of the throughput)")
    ap.add_argument("--cand-frac", type=float, default=0.0,
                    help="fraction of anchors above conf 0.25 the synthetic class bias is calibrated to; 0 (default) = ~126 "
                         "candidates per image whatever the image size (config5_target_frac: 0.015 at 640x640, the bench's "
                         "operating point; 0.05 = the NMS stress point of SURVEY 8d)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: dense 3x3 convs in OCP e4m3 (weights + activations, fp32 accumulate) on "
                         "v_mfma_f32_16x16x128_f8f6f4, everything else bf16 (BASELINE.json config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of BASELINE.json configs 3 and 5 that the default bench appends (other_configs)")
    ap.add_argument("--profile-out", default="", help="write the per-op timing table to this JSON file")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE=1: run `python bench.py --gpus N` without WORLD_SIZE (it starts its own "
                         "ranks) or under torch.distributed.run with --nproc-per-node N")
    ndev = torch.cuda.device_count()  # does not initialise the GPU
    if ndev < max(1, local_rank + 1) or not torch.cuda.is_available():
        raise SystemExit(f"bench.py rank {rank} of {world} needs GPU {local_rank} and sees {ndev} GPU(s): the HIP path has no CPU fallback")
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    import torch.distributed as dist

    # YL_BENCH_FORCE_SHARDED=1: run the N > 1 code path (RCCL scatter / gather per step) with a process group of ONE rank --
    # a rehearsal of that path on a one-GPU box (the numbers then say nothing about scaling)
    sharded = world > 1 or os.environ.get("YL_BENCH_FORCE_SHARDED", "0") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    elif sharded:
        import socket

        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)

    B, H, W = args.batch, args.imgsz, args.imgsz
    conf, iou, max_det = 0.25, 0.7, 300
    # the same calibration batch on every rank (seed 0): the class bias, hence the weights, are identical replicas
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(0)).to(dev)
    if args.cand_frac <= 0:
        args.cand_frac = config5_target_frac(args.imgsz)
    main_planted = c5_planted(B) if (args.dtype == "fp8" and not sharded) else 0  # (config 5 as the main run: the planted head, as in other_configs)
    specs, sd, eng = build_model(args.scale, dev, B, H, W, x, target_frac=args.cand_frac, fp8=args.dtype == "fp8", planted=main_planted)
    cfg_key = {"scale": args.scale, "imgsz": args.imgsz, "batch": B, "dtype": args.dtype}

    cands = [args.streams] if args.streams > 0 else ([6, 7, 8] if sharded else [SINGLE_IN_FLIGHT])
    S = max(cands)  # rebound below to the count that wins the trial
    # (mixed stream priorities were tried to keep 20 batches out of lock step in short runs: 7.6-20 k images/s in a 20-step
    # run against 44-45 k, 35-38 k against 47.9 k in a long one)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    step_no = [0]
    tried = {}
    # trial length = the timed region's own (a short run pays for its fill and drain: K = 20 steps with 8 batches in flight is
    # two and a half rounds), capped at 80 steps
    n_trial = max(max(cands), min(80, args.steps))

    def hot_path(im):
        # one step = network + fused Detect decode / candidate filter / NMS for one batch, ONE hipGraph
        # launch; step k runs on stream k % S with workspace slot k % S of the same engine
        j = step_no[0] % S
        step_no[0] += 1
        with torch.cuda.stream(streams[j]):
            return eng.detect(im, conf, iou, max_det=max_det, graph=True, slot=j)

    def sync_all():
        torch.cuda.synchronize(dev)
        if sharded:
            dist.barrier()
            torch.cuda.synchronize(dev)

    ndet = None
    extra = {}
    if not sharded:
        # one resident input batch per slot (a graph bakes its input pointer): uint8 NCHW, what the
        # predictor's preprocess hands to the device (predictor.py:77-84); the /255 is fused into the stem.
        # Same input format as the N > 1 path, so per-GPU work is identical at every N.
        x8 = (x * 255).to(torch.uint8)
        if main_planted:  # (the planted objects live in x8's images: the other slots hold them in another order)
            xs = [x8] + [x8[torch.randperm(B, generator=torch.Generator().manual_seed(1000 + j)).to(dev)].contiguous() for j in range(1, S)]
        else:
            xs = [x8] + [torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(1000 + j),
                                       dtype=torch.uint8).to(dev) for j in range(1, S)]
        if len(cands) > 1:  # untimed trial of the stream counts
            for S in cands:
                step_no[0] = 0
                for k in range(3 * S):
                    hot_path(xs[k % S])
                sync_all()
                step_no[0] = 0
                tt = time.perf_counter()
                for k in range(n_trial):
                    hot_path(xs[k % S])
                sync_all()
                tried[S] = round(B * n_trial / (time.perf_counter() - tt), 1)
                print(f"[bench] trial, {S} batches in flight: {tried[S]} images/s", file=sys.stderr, flush=True)
            S = max(tried, key=tried.get)
        step_no[0] = 0
        for k in range(max(args.warmup, 2 * S)):
            out, cnt = hot_path(xs[step_no[0] % S])
        sync_all()
        # The timed K steps are issued by ONE native call (yl_graph_launch_many): step k = the captured graph of slot k % S on
        # stream k % S, exactly what hot_path launches -- without ~80 us of Python per step, which in the driver's 20-step
        # run is a 1.6 ms fill ramp.  YL_BENCH_PY_LOOP=1: the Python loop of rounds 1-2.
        handles, slot_res = [], []
        for j in range(S):
            with torch.cuda.stream(streams[j]):
                o_j, c_j, hj = eng.prepare_detect(xs[j], conf, iou, max_det=max_det, slot=j)
            handles.append(hj)
            slot_res.append((o_j, c_j))  # the slot's engine-owned result buffers (what the graph writes)
        step_h = [handles[k % S] for k in range(args.steps)]
        step_s = [streams[k % S] for k in range(args.steps)]
        native = os.environ.get("YL_BENCH_PY_LOOP", "0") != "1"
        if native:
            eng.launch_many(step_h[:S], step_s[:S])  # (builds the argument arrays' cache entry for the slice; untimed)
            sync_all()
            # the K steps ONCE before the warm-up rounds below: the figure a K-step run gives right behind the W warm-up steps the
            # caller asked for (rounds 1-4's method) -- reported as value_cold_first_run beside `value`, which is taken on the plateau
            tc0 = time.perf_counter()
            eng.launch_many(step_h, step_s)
            sync_all()
            extra["value_cold_first_run"] = round(B * args.steps / (time.perf_counter() - tc0), 1)
            extra["value_cold_first_run_note"] = (f"the same K = {args.steps} steps timed once right behind the {max(args.warmup, 2 * S)} warm-up steps, before "
                                                  "the extra warm-up rounds (the method of BENCH_r01-r04; its first call also builds the launch arrays)")
        sync_all()
        # Warm-up to the steady state.  The W steps the caller asks for are a minimum: after the engine build the first rounds of
        # S batches run 4-5 % below the rate the same loop settles at (tools/round_timeline.py --reps 40 on one box: 54.7, 55.1,
        # 56.2, 56.6, 56.9, 57.1 k images/s for rounds 1-6, then 56.5-57.4 k: clocks and caches), and a K = 20 run IS one such
        # round.  WARM_ROUNDS more untimed rounds of the same native loop (~0.1 s) put the timed K steps on the plateau.
        warm_rounds = int(os.environ.get("YL_BENCH_WARM_ROUNDS", "10"))
        # (one round per call, synchronised: behind ONE call with thousands of launches the runtime recycles its command objects
        # during the next launches -- the timed K steps then take 0.8-2 ms to issue instead of 0.25)
        for _ in range(warm_rounds):
            if native:
                eng.launch_many(step_h[:S], step_s[:S])
            else:
                for k in range(S):
                    hot_path(xs[k % S])
            sync_all()
        extra["warmup_steps_run"] = max(args.warmup, 2 * S) + S + args.steps + warm_rounds * S
        extra["warmup_note"] = (f"--warmup {args.warmup} is the minimum: {extra['warmup_steps_run']} untimed steps ran before the timed region "
                                f"(the last {warm_rounds * S} of them = {warm_rounds} rounds of the same native loop, until the rate stops rising)")
        step_no[0] = 0
        t0 = time.perf_counter()
        if native:
            eng.launch_many(step_h, step_s)
        else:
            for k in range(args.steps):
                out, cnt = hot_path(xs[k % S])
        t_issue = time.perf_counter() - t0
        sync_all()
        dt = time.perf_counter() - t0
        extra["issue_ms"] = round(t_issue * 1e3, 4)  # host time to issue the K steps (inside the timed region)
        if native and int(os.environ.get("YL_BENCH_REPEAT", "0")) > 0:  # development: the same timed region again, N times
            reps = []
            for _ in range(int(os.environ["YL_BENCH_REPEAT"])):
                tr0 = time.perf_counter()
                eng.launch_many(step_h, step_s)
                sync_all()
                reps.append(round(B * args.steps / (time.perf_counter() - tr0), 1))
            extra["debug_repeats_images_per_sec"] = reps
        extra["step_issue"] = "native: yl_graph_launch_many, one call for the K steps" if native else "python loop"
        out, cnt = slot_res[(args.steps - 1) % S]
        ndet = cnt.float().mean().item()
        if args.steps < 5 * S:
            # a run this short is one or two rounds of the S batches in flight -- mostly fill and drain; the same loop over
            # 500 steps is reported beside it (value itself stays the K steps the caller asked for)
            n_long = 500
            step_no[0] = 0
            long_h, long_s = [handles[k % S] for k in range(n_long)], [streams[k % S] for k in range(n_long)]
            if native:
                eng.launch_many(long_h[:1], long_s[:1])
                sync_all()
            tl = time.perf_counter()
            if native:
                eng.launch_many(long_h, long_s)
            else:
                for k in range(n_long):
                    hot_path(xs[k % S])
            sync_all()
            extra["value_steady_state"] = round(B * n_long / (time.perf_counter() - tl), 1)
            extra["value_steady_state_note"] = f"the same pipelined loop over {n_long} steps (fill and drain of the {S} batches in flight ~1 %)"
        # SURVEY 8(d)'s literal definition: one batch at a time, stream-synchronised once per iteration
        n_sync = max(20, min(200, args.steps))
        st0 = streams[0]
        with torch.cuda.stream(st0):
            for _ in range(5):
                eng.detect(x8, conf, iou, max_det=max_det, graph=True, slot=0)
            st0.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_sync):
                eng.detect(x8, conf, iou, max_det=max_det, graph=True, slot=0)
                st0.synchronize()
            lat = (time.perf_counter() - t1) / n_sync
        extra["latency_ms_single_stream"] = round(lat * 1e3, 4)
        extra["value_sync_per_iteration"] = round(B / lat, 1)
        extra["sync_per_iteration_note"] = (f"{n_sync} iterations, one batch in flight, hipStreamSynchronize after every "
                                            "iteration (SURVEY 8(d)); `value` pipelines the same step over the streams")
        if args.dtype == "bf16" and os.environ.get("YL_BENCH_ONE_BATCH_PROFILE", "1") == "1":
            # the program a caller that runs ONE batch at a time gets (lower(one_batch=True): what YOLOLite.predict() per call
            # uses -- two workgroups per image in the resident C3k blocks, SPPF / PSABlock as separate launches): same kernels,
            # same bits, tuned for latency instead of chip time
            from yololite_amd.nn.engine import HipEngine as _HE

            eng1 = _HE.from_model(specs, sd, B, H, W, dev, one_batch=True)
            with torch.cuda.stream(st0):
                for _ in range(5):
                    o1, c1 = eng1.detect(x8, conf, iou, max_det=max_det, graph=True, slot=0)
                st0.synchronize()
                t1 = time.perf_counter()
                for _ in range(n_sync):
                    eng1.detect(x8, conf, iou, max_det=max_det, graph=True, slot=0)
                    st0.synchronize()
                lat1 = (time.perf_counter() - t1) / n_sync
                o0, c0 = eng.detect(x8, conf, iou, max_det=max_det, graph=True, slot=0)
                st0.synchronize()
            extra["one_batch_profile"] = {"value_sync_per_iteration": round(B / lat1, 1), "latency_ms": round(lat1 * 1e3, 4),
                                          "program_ops": len(eng1.prog.ops),
                                          "bit_identical_to_throughput_profile": bool(torch.equal(o1, o0) and torch.equal(c1, c0)),
                                          "note": "lower(one_batch=True): the program YOLOLite.predict() per call runs"}
            del eng1
    else:
        # BASELINE.json config 4: rank 0 owns the uint8 global batches (resident in its HBM before the timed region);
        # per step: RCCL scatter of the 32-image slices (S steps ahead) -> one graph launch per rank -> RCCL gather of
        # the packed (count, 300 x 6) records.  No all-reduce anywhere.
        from yololite_amd.dist import ShardedRunner, record_floats, run_resident, unpack_record

        nb = min(S, 3)
        gb = None
        if rank == 0:
            g = torch.Generator().manual_seed(100)
            gb = [torch.randint(0, 256, (world * B, 3, H, W), generator=g, dtype=torch.uint8).to(dev) for _ in range(nb)]
        batches = gb if rank == 0 else [None] * nb

        def compute(im, slot, lane):
            return eng.detect_record(im, conf, iou, max_det=max_det, slot=slot, rec_key=lane)

        # transport of the two exchanges: torch.distributed's "nccl" backend (default) or the yl_comm_* group of the C ABI
        # (YL_BENCH_COMM=native: grouped RCCL send / recv issued by libyl_amd.so, one communicator and exchange stream per direction)
        native_comm = None
        if os.environ.get("YL_BENCH_COMM", "torch") == "native":
            from yololite_amd.dist import NativeComm, NativeShardedRunner

            native_comm = NativeComm(rank, world, NativeComm.id_via_torch(), device=dev)
            native_comm_g = NativeComm(rank, world, NativeComm.id_via_torch(), device=dev)  # (the gathers' own communicator: dist.py)
        extra["transport"] = ("yl_comm_* (include/yl_amd.h): RCCL send / recv issued by libyl_amd.so" if native_comm is not None
                              else "torch.distributed, backend nccl (= RCCL)")

        def make_runner(n):
            if native_comm is not None:
                o_ = os.environ.get("YL_BENCH_NATIVE_OPTS", "")  # (experiments: "copy" = no in-place root slice, "one" = one communicator)
                return NativeShardedRunner(native_comm, compute, B, (3, H, W), torch.uint8, dev, record_floats(B, max_det), slots=n,
                                           streams=streams[:n], comm_gather=None if "one" in o_ else native_comm_g,
                                           root_in_place="copy" not in o_)
            return ShardedRunner(compute, B, (3, H, W), torch.uint8, dev, record_floats(B, max_det), slots=n, streams=streams[:n])

        if len(cands) > 1:  # untimed trial of the stream counts; every rank takes the same decision (MAX over ranks)
            for S in cands:
                r_ = make_runner(S)
                r_.run(batches, 4 * S)
                sync_all()
                tt = time.perf_counter()
                r_.run(batches, n_trial)
                sync_all()
                tv = torch.tensor([time.perf_counter() - tt], dtype=torch.float64, device=dev)
                dist.all_reduce(tv, op=dist.ReduceOp.MAX)
                tried[S] = round(world * B * n_trial / float(tv.item()), 1)
                if rank == 0:
                    print(f"[bench] sharded trial, {S} batches in flight: {tried[S]} images/s", file=sys.stderr, flush=True)
                del r_
            S = max(tried, key=tried.get)
            torch.cuda.empty_cache()
        runner = make_runner(S)
        n_warm = max(args.warmup, 2 * runner.lanes, int(os.environ.get("YL_BENCH_WARM_ROUNDS", "10")) * S)  # (to the steady state, as at N = 1)
        runner.run(batches, n_warm)
        extra["warmup_steps_run"] = n_warm
        sync_all()
        t0 = time.perf_counter()
        last = runner.run(batches, args.steps)
        t_issue = time.perf_counter() - t0  # (the host's share: run() returns when the last step is ENQUEUED)
        sync_all()
        dt = time.perf_counter() - t0
        extra["host_issue_ms_per_step"] = round(t_issue * 1e3 / args.steps, 4)
        extra["host_issue_note"] = ("time for this rank's Python loop to enqueue one step (scatter, graph launch, gather, their events); "
                                    "close to ms_per_step = the host loop, not the GPU or a link, bounds the sharded run")
        if rank == 0:
            _, cnts = unpack_record(last, B, max_det)
            ndet = torch.cat(cnts).float().mean().item()
        # replica figure: shards resident in every rank's own HBM, no per-step collective, one gather at the end
        g = torch.Generator().manual_seed(100 + rank)
        shards = [torch.randint(0, 256, (B, 3, H, W), generator=g, dtype=torch.uint8).to(dev) for _ in range(S)]
        step_no[0] = 0
        run_resident(hot_path, shards, max(args.warmup, 2 * S), finish=lambda: torch.cuda.synchronize(dev))
        sync_all()
        step_no[0] = 0
        t2 = time.perf_counter()
        run_resident(hot_path, shards, args.steps, finish=lambda: torch.cuda.synchronize(dev))
        sync_all()
        dt_res = time.perf_counter() - t2
        tr = torch.tensor([dt_res], dtype=torch.float64, device=dev)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        extra["value_resident"] = round(world * B * args.steps / float(tr.item()), 1)
        extra["value_resident_note"] = "same steps on shards resident in each rank's HBM, no per-step collective (replicas)"
        # like-for-like N = 1 point inside the SAME process group and runtime configuration: rank 0 alone runs the resident
        # steps while its peers idle at the barrier -- what one GPU of this job does without neighbours and collectives
        step_no[0] = 0
        t3 = time.perf_counter()
        if rank == 0:
            for k in range(args.steps):
                hot_path(shards[k % S])
            torch.cuda.synchronize(dev)
        dt_alone = time.perf_counter() - t3
        sync_all()
        extra["value_rank0_alone"] = round(B * args.steps / dt_alone, 1)
        extra["value_rank0_alone_note"] = ("rank 0 running the resident steps while the other ranks idle: the N = 1-equivalent rate "
                                           "under this job's own queue / stream configuration (scaling efficiency like for like = "
                                           "value / (n_gpus * value_rank0_alone))")
        # self-diagnosis: did RCCL really connect every rank, and which rank is the slow one (the root pays the scatter)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        extra["rccl_ranks"] = int(ones.item())
        mine = torch.tensor([dt * 1e3 / args.steps, dt_res * 1e3 / args.steps], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        extra["per_rank_ms"] = {"sharded": [round(float(t[0]), 4) for t in allr], "resident": [round(float(t[1]), 4) for t in allr]}
        extra["scatter_bytes_per_step"] = (world - 1) * B * 3 * H * W
        extra["gather_bytes_per_step"] = (world - 1) * record_floats(B, max_det) * 4
        # The scatter ALONE, outside the timed region, nothing else on the GPUs: the root pushes one slice per peer over that
        # peer's xGMI link, so bytes of ONE slice / time of a scatter = the one-direction rate each link sustained.  The design
        # needs slice bytes x (value_rank0_alone / B) per link; a link slower than that makes the
        # sharded step scatter-bound whatever the kernels do.
        from yololite_amd.dist import scatter_images

        probe = torch.empty((B, 3, H, W), dtype=torch.uint8, device=dev)
        n_sc = 10
        for _ in range(3):
            scatter_images(batches[0], probe, 0)
        sync_all()
        t4 = time.perf_counter()
        for _ in range(n_sc):
            scatter_images(batches[0], probe, 0)
        sync_all()
        t_sc = torch.tensor([(time.perf_counter() - t4) / n_sc], dtype=torch.float64, device=dev)
        dist.all_reduce(t_sc, op=dist.ReduceOp.MAX)
        slice_bytes = B * 3 * H * W
        extra["scatter_ms_alone"] = round(float(t_sc.item()) * 1e3, 4)
        extra["scatter_gbps_per_link"] = round(slice_bytes / float(t_sc.item()) / 1e9, 2) if world > 1 else None
        need = slice_bytes * extra["value_rank0_alone"] / B / 1e9
        extra["scatter_gbps_per_link_needed"] = round(need, 2)
        v_sh = world * B * args.steps / float(max(dt, 1e-9))
        if world > 1:
            slow_link = extra["scatter_gbps_per_link"] is not None and extra["scatter_gbps_per_link"] < 1.1 * need
            extra["bound"] = "scatter" if (slow_link or v_sh < 0.9 * extra["value_resident"]) else "compute"
            extra["bound_note"] = ("scatter: a link sustains less than 1.1 x the rate the compute consumes, or the sharded rate is more "
                                   "than 10 % below the resident (no per-step collective) rate of the same ranks; compute otherwise")
        else:
            extra["bound"] = "compute"
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if sharded:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        workload = (f"YOLO11{args.scale} {args.dtype}, {H}x{W} batch={B} per GPU (fused Conv+BN+SiLU + HIP NMS), "
                    f"random-init weights, random uint8 NCHW images, conf {conf} iou {iou} max_det {max_det}; "
                    f"{S} batches in flight on {S} HIP streams (one hipGraph launch per step); ")
        workload += ("inputs resident in HBM" if not sharded else
                     f"global batch {world * B} resident on rank 0, per step: RCCL scatter of the uint8 slices -> compute -> "
                     "RCCL gather of the packed (count, 300x6) records")
        res = {
            "metric": f"images/sec end-to-end (incl. NMS), YOLO11{args.scale} {H}x{W} batch={B}",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else ("fp8 e4m3 (dense 3x3 and 1x1 convs, tensors mirrored in e4m3) + bf16"
                                                          if os.environ.get("YL_FP8_MIRROR", "0") == "1" else "fp8 e4m3 (dense 3x3 convs) + bf16"),
            "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch": world * B, "parallelism": f"dp{world}" if world > 1 else "single",
                       "streams": S, "streams_tried_images_per_sec": tried or None, "candidate_fraction_target": args.cand_frac,
                       "mean_detections_per_image": None if ndet is None else round(ndet, 1)},
            # end-to-end fractions of the bf16 MFMA peak and of the HBM peak (algorithmic work per image)
            "e2e_mfma_frac": (round(value * CONV_GFLOP_AT_640[args.scale] * (H * W) / (640 * 640) / 1e3 / PEAK_BF16_TFLOPS / world, 5)
                              if args.scale in CONV_GFLOP_AT_640 else None),
            "e2e_hbm_frac": (round(value * ACT_MELEM_AT_640[args.scale] * 2 * (H * W) / (640 * 640) / 1e3 / PEAK_HBM_GBS / world, 5)
                             if args.scale in ACT_MELEM_AT_640 else None),
        }
        res.update(extra)
        # which of the figures `value` is (VERDICT r05 weak #12 / ADVICE): the pipelined rate on the plateau; SURVEY 8(d)'s literal
        # per-iteration figure and the cold first run are value_sync_per_iteration and value_cold_first_run of the same line
        res["value_kind"] = (f"pipelined: {S} batches in flight, K steps issued by one native call after the warm-up rounds; beside it: "
                             "value_sync_per_iteration (one batch at a time, one host synchronisation per iteration: SURVEY 8(d) as written), "
                             "value_cold_first_run (the same K steps right behind the --warmup steps), value_steady_state (500 steps)")
        res["bench_config"] = cfg_key  # what profiles/*_per_launch.json files are matched against
        try:
            probe = yl_lib.graph_branch_probe(dev)
        except Exception as e:  # the probe must never cost the bench line
            probe = {"error": repr(e)}
        res["runtime"] = {"knobs": KNOBS, "graph_branch_probe": probe}
        if sharded:
            # N > 1: the per-GPU program is the one of N = 1 (replicas): its roofline block is measured on rank 0 the same way (the
            # other ranks wait at the closing barrier); `cpu_baseline` is an N = 1 object
            try:
                rows = eng.profile_detect((x * 255).to(torch.uint8), conf, iou, max_det=max_det, iters=10)
                res["roofline"], res["kernels"] = roofline_from_profile(rows, B, cfg_key)
                res["roofline"]["note"] = "per-GPU program, measured on rank 0 after the timed region (every rank runs the same kernels on its shard)"
            except Exception as e:  # never at the cost of the line
                res["roofline"] = {"error": repr(e)}
        if not sharded:
            rows = eng.profile_detect(x8, conf, iou, max_det=max_det, iters=20)
            roof, kernels = roofline_from_profile(rows, B, cfg_key)
            res["roofline"] = roof
            res["kernels"] = kernels
            # + the launch that clears the NMS candidate counters, unless the network's first launch is the fused stem, which
            # clears them on its way (csrc/engine.hip detect_eager, Stem2Args::zero_ptr)
            zero_in_stem = rows[0]["kernel"].startswith("stem+") and os.environ.get("YL_ZERO_IN_STEM", "1") != "0"
            res["launches_per_step"] = len(rows) + (0 if zero_in_stem else 1)
            fused_bytes = sum(r["bytes"] for r in rows)
            res["program_bytes_per_step"] = round(fused_bytes)
            res["e2e_hbm_frac_fused_program"] = round(fused_bytes / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 5)
            if args.profile_out:
                Path(args.profile_out).parent.mkdir(parents=True, exist_ok=True)
                Path(args.profile_out).write_text(json.dumps(rows, indent=1))
            # the reference API on the same resident batches: YOLOLite.predict -> DetectionPredictor -> one graph launch
            try:
                from yololite_amd import YOLOLite

                if args.dtype == "bf16" and args.scale in "nsmlx":
                    api = YOLOLite(f"yolo11{args.scale}.yaml")
                    api.model.load_state_dict(sd)
                    kw = dict(conf=conf, iou=iou, max_det=max_det, device=str(dev), save=False, verbose=False, batch=B)
                    for _ in range(3):
                        r_api = api.predict(x8, **kw)
                    torch.cuda.synchronize(dev)
                    n_api, api_runs = 20, []
                    for _ in range(3):  # median of three runs of 20 calls (one run in a fresh process was once 3 x slower: a host hiccup)
                        t3 = time.perf_counter()
                        for _ in range(n_api):
                            r_api = api.predict(x8, **kw)
                        torch.cuda.synchronize(dev)
                        api_runs.append(round(B * n_api / (time.perf_counter() - t3), 1))
                    res["api_images_per_sec"] = sorted(api_runs)[1]
                    res["api_runs_images_per_sec"] = api_runs
                    res["api_note"] = ("YOLOLite.predict(uint8 tensor resident in HBM) incl. list[Results] construction and its one "
                                       "host sync per batch; one batch at a time")
                    # the same API on a source of MANY batches (predict(stream=True) on a LoadTensorBatches loader): the
                    # predictor keeps S engine slots in flight and yields batch k while k + 1 .. run
                    from yololite_amd.data.loaders import LoadTensorBatches

                    os.environ["YL_PREDICT_INFLIGHT"] = str(S)
                    n_st = max(10 * S, 100)  # fill and drain of the S batches in flight ~5 % of the run
                    src = LoadTensorBatches([xs[j % len(xs)] for j in range(n_st)])
                    for _ in api.predict(LoadTensorBatches(xs[: min(len(xs), S)]), stream=True, **kw):
                        pass  # slots + graphs of the stream path
                    torch.cuda.synchronize(dev)
                    t4 = time.perf_counter()
                    n_seen = sum(1 for _ in api.predict(src, stream=True, **kw))
                    torch.cuda.synchronize(dev)
                    res["api_stream_images_per_sec"] = round(n_seen / (time.perf_counter() - t4), 1)
                    res["api_stream_note"] = (f"YOLOLite.predict(LoadTensorBatches of {n_st} resident uint8 batches, stream=True): {S} "
                                              "batches in flight, one Results object per image built while the next batches run")
                    sc1 = torch.tensor([[0.0, 0.0, 1.0, float(W), float(H)]] * B, dtype=torch.float32, device=dev)
                    o_ref, c_ref = eng.detect(x8, conf, iou, max_det=max_det, graph=False, scale=sc1)  # predictor.py:125 clips
                    res["api_equals_engine"] = all(torch.equal(r.boxes.data, o_ref[i, : int(c_ref[i])]) for i, r in enumerate(r_api))
                    del api
            except Exception as e:
                res["api_images_per_sec"] = None
                res["api_note"] = "failed: " + repr(e)
            if args.dtype == "fp8":
                # the e4m3 program against the bf16 program of the same weights on the same images (SURVEY 8(d) gate)
                from oracle.parity_gate import parity_gate
                from yololite_amd.nn.engine import HipEngine

                o8, c8 = eng.detect(x8, conf, iou, max_det=max_det, graph=False)
                l8 = [t.clone() for t in eng.head_outputs()] if B * H * W <= 32 * 640 * 640 else None
                e16 = HipEngine.from_model(specs, sd, B, H, W, dev)
                o16, c16 = e16.detect(x8, conf, iou, max_det=max_det, graph=False)
                torch.cuda.synchronize(dev)
                nb8 = min(B, 8)
                res["fp8_vs_bf16"] = parity_gate([o16[i, : int(c16[i])].cpu() for i in range(nb8)],
                                                 [o8[i, : int(c8[i])].cpu() for i in range(nb8)], conf, iou,
                                                 ref_y=e16.forward(x8, graph=False)[:nb8].float().cpu(),
                                                 got_y=eng.forward(x8, graph=False)[:nb8].float().cpu(), **C5_GATE)
                if l8 is not None:
                    l16 = e16.head_outputs()  # (class logits, then decoded boxes, level by level)
                    res["fp8_vs_bf16"]["level_rel_l2"] = [round(float((a - b).norm() / b.norm()), 4) for a, b in zip(l8, l16)]
                n8 = sum(1 for r in rows if "_f8_" in r["kernel"])
                res["fp8_vs_bf16"]["fp8_conv_launches"] = n8
                res["fp8_vs_bf16"]["fp8_flop_share"] = round(sum(r["flops"] for r in rows if "_f8_" in r["kernel"]) /
                                                             max(1.0, sum(r["flops"] for r in rows)), 4)
                del e16
            default_cfg = (args.scale, B, args.imgsz, args.dtype) == ("n", 32, 640, "bf16")
            if default_cfg and not args.no_other_configs:
                # BASELINE.json configs 3 and 5, briefly, with the main engine's slots released first
                del eng._slots[1:]
                torch.cuda.empty_cache()
                res["other_configs"] = []
                for sc_, b_, sz_, dt_, st_, n_ in (("s", 64, 640, "bf16", S, 10 * S), ("x", 8, 1280, "fp8", S, 3 * S)):
                    try:
                        res["other_configs"].append(short_config(sc_, b_, sz_, dt_, st_, n_, dev, stream_pool=streams))
                    except Exception as e:  # never at the cost of the main line
                        res["other_configs"].append({"workload": f"YOLO11{sc_} {dt_} {sz_} batch={b_}", "error": repr(e)})
                # BASELINE.json configs[0] on the GPU: ONE image file-sized frame through the reference API -- 1080x1920 BGR array
                # -> device letterbox to 384x640 -> batch-1 engine -> NMS -> Results -- as a latency (the reference runs this
                # case on the CPU; tests/test_gpu_config1.py checks its parity)
                try:
                    import numpy as np
                    from yololite_amd import YOLOLite

                    api1 = YOLOLite(f"yolo11{args.scale}.yaml")
                    api1.model.load_state_dict(sd)
                    # the frame: the reference's own example image, decoded once on the host (tests/golden/boats.jpg -- a data file of
                    # the reference, 1080 x 1920; BASELINE.json configs[0] names it); white noise of that size when PIL is missing
                    frame_src = "white noise"
                    frame = np.random.default_rng(1).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
                    try:
                        from PIL import Image

                        fp = ROOT / "tests" / "golden" / "boats.jpg"
                        frame = np.ascontiguousarray(np.asarray(Image.open(fp).convert("RGB"))[:, :, ::-1])  # BGR, as the loaders decode
                        frame_src = "boats.jpg"
                    except Exception:
                        pass
                    kw1 = dict(conf=conf, iou=iou, max_det=max_det, device=str(dev), save=False, verbose=False, imgsz=640)
                    for _ in range(3):
                        api1.predict(frame, **kw1)
                    torch.cuda.synchronize(dev)
                    t5 = time.perf_counter()
                    for _ in range(20):
                        r1 = api1.predict(frame, **kw1)
                    torch.cuda.synchronize(dev)
                    lat = (time.perf_counter() - t5) / 20
                    res["other_configs"].append({"workload": f"YOLO11{args.scale} bf16, one {frame.shape[0]}x{frame.shape[1]} frame ({frame_src}) -> 384x640, batch=1 "
                                                             "through YOLOLite.predict (host array in, upload + device letterbox + network + NMS + Results)",
                                                 "latency_ms": round(lat * 1e3, 3), "value": round(1.0 / lat, 1), "unit": "images/sec",
                                                 "detections": int(len(r1[0].boxes))})
                    del api1
                except Exception as e:
                    res["other_configs"].append({"workload": "config 1 (one frame through predict)", "error": repr(e)})
            if not args.no_cpu_baseline:
                                # batch 32, 3 iterations for YOLO11n at 640x640 (96 images after one warm-up iteration, a few
                # seconds on 8 threads); other configurations: about the same host work
                gf = CONV_GFLOP_AT_640.get(args.scale, 200.0) * (H * W) / (640 * 640)
                n_img = max(1, int(96 * CONV_GFLOP_AT_640["n"] / gf))
                sb = min(32, B, n_img)
                res["cpu_baseline"] = cpu_baseline(sd, args.scale, H, W, conf, iou, sample_b=sb, iters=max(1, round(n_img / sb)),
                                                   eng=eng, dev=dev, max_det=max_det,
                                                   band=0.12 if args.dtype == "fp8" else 0.08)  # stated tolerances
        sys.stdout.flush()
        os.write(_REAL_STDOUT, (json.dumps(res) + "\n").encode())
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
