"""DetectionPredictor: preprocess -> inference -> postprocess with the reference's hooks
(``yololite/engine/predictor.py:21-246``).

Differences from the reference, all deliberate and documented in DESIGN.md:
* ``inference`` runs the HIP engine (bf16 activations, fp32 accumulate/decode), not ``nn.Module`` s.
* ``postprocess`` keeps results on the device and fuses ``scale_boxes`` + clip into the NMS kernel's
  output stage; one host sync (the per-image counts) per batch instead of three ``synchronize`` s.
* for tensor sources the uint8 host copy of the input (``convert_torch2numpy_batch``,
  predictor.py:120-121) is only made when ``keep_orig_img=True``; ``Results.orig_shape`` is always set.
* saving / plotting are result sinks outside the accelerated path, drawn with PIL instead of cv2
  (``save``, ``save_crop``, ``save_txt`` work for image sources; ``show`` needs a display and raises).
"""
from __future__ import annotations

import os
import threading
from pathlib import Path
from typing import List

import numpy as np
import torch

from ..cfg import DEFAULT_CFG, get_cfg, get_save_dir
from ..data import load_inference_source
from ..data.augment import LetterBox
from ..nn.autobackend import AutoBackend
from ..utils import LOGGER, ops
from ..utils.checks import check_imgsz
from ..utils.torch_utils import select_device, smart_inference_mode
from .results import Boxes, Results


class DetectionPredictor:
    def __init__(self, cfg=DEFAULT_CFG, overrides=None, keep_orig_img: bool = False):
        self.args = get_cfg(cfg, overrides)
        self._save_dir = None
        if self.args.conf is None:
            self.args.conf = 0.25
        self.done_warmup = False
        self.model = None
        self.data = self.args.data
        self.imgsz = None
        self.device = None
        self.dataset = None
        self.source_type = None
        self.seen = 0
        self.batch = None
        self.results = None
        self.txt_path = None
        self.keep_orig_img = keep_orig_img
        self.fuse_postprocess = True  # set False to run the reference's three separate stages
        self.last_preds = None
        self._lock = threading.Lock()

    @property
    def save_dir(self) -> Path:
        if self._save_dir is None:
            self._save_dir = get_save_dir(self.args)
        return self._save_dir

    # ------------------------------------------------------------------ hooks
    def preprocess(self, im):
        """BCHW tensor passthrough, or list of HWC BGR uint8 -> letterbox -> RGB CHW /255
        (reference predictor.py:67-85)."""
        not_tensor = not isinstance(im, torch.Tensor)
        if not_tensor and self.device is not None and torch.device(self.device).type == "cuda" and \
                type(self).pre_transform is DetectionPredictor.pre_transform:
            return self._device_letterbox(im)  # letterbox + RGB + CHW on the GPU, uint8 batch for the stem
        if not_tensor:
            im = np.stack(self.pre_transform(im))
            im = im[..., ::-1].transpose((0, 3, 1, 2))
            im = torch.from_numpy(np.ascontiguousarray(im))
        im = im.to(self.device)
        if not_tensor:
            return im  # uint8: the engine's input kernel divides by 255 (predictor.py:82-84)
        return im if im.dtype in (torch.float16, torch.float32, torch.uint8) else im.float()

    def _device_letterbox(self, im: List[np.ndarray]):
        from ..data.augment import DeviceLetterBox

        same_shapes = len({x.shape for x in im}) == 1
        key = (tuple(self.imgsz) if isinstance(self.imgsz, (list, tuple)) else self.imgsz, bool(same_shapes and self.model.pt))
        cache = self.__dict__.setdefault("_dlb", {})
        if key not in cache:
            cache[key] = DeviceLetterBox(self.imgsz, auto=key[1], stride=self.model.stride, device=self.device)
        return cache[key](im)

    def pre_transform(self, im: List[np.ndarray]):
        same_shapes = len({x.shape for x in im}) == 1
        letterbox = LetterBox(self.imgsz, auto=same_shapes and self.model.pt, stride=self.model.stride)
        return [letterbox(image=x) for x in im]

    def inference(self, im, *args, **kwargs):
        if self.args.visualize or self.args.augment or self.args.embed:
            raise NotImplementedError("visualize / augment / embed are outside the accelerated predict path")
        return self.model(im, *args, **kwargs)

    def postprocess(self, preds, img, orig_imgs):
        """NMS + rescale to the original image + Results (reference predictor.py:110-127)."""
        pred = preds[0] if isinstance(preds, (list, tuple)) else preds
        self.last_preds = pred
        B = pred.shape[0]
        if isinstance(orig_imgs, torch.Tensor):
            shapes = [tuple(orig_imgs.shape[2:])] * B
            imgs = ops.convert_torch2numpy_batch(orig_imgs) if self.keep_orig_img else [None] * B
        else:
            shapes = [tuple(o.shape[:2]) for o in orig_imgs]
            imgs = orig_imgs
        sc = []
        for h0, w0 in shapes:
            gain, pad = ops.scale_boxes_params(tuple(img.shape[2:]), (h0, w0))
            sc.append([pad[0], pad[1], gain, w0, h0])
        scale = torch.tensor(sc, dtype=torch.float32, device=pred.device)
        out, cnt = ops.nms_padded(pred, self.args.conf, self.args.iou, classes=self.args.classes,
                                  agnostic=self.args.agnostic_nms, max_det=self.args.max_det, scale=scale)
        counts = cnt.tolist()
        return [Results(imgs[i], path=self.batch[0][i], names=self.model.names, boxes=out[i, : counts[i]],
                        orig_shape=shapes[i]) for i in range(B)]

    # ------------------------------------------------------------------ fused fast path
    def _can_fuse(self) -> bool:
        """The fused path is used when the hooks are not overridden and nobody asked for ``last_preds``."""
        cls = type(self)
        return (self.fuse_postprocess and cls.inference is DetectionPredictor.inference
                and cls.postprocess is DetectionPredictor.postprocess
                and not (self.args.visualize or self.args.augment or self.args.embed))

    def _shapes_and_scale(self, img, orig_imgs, B, device):
        if isinstance(orig_imgs, torch.Tensor):
            shapes = [tuple(orig_imgs.shape[2:])] * B
            imgs = ops.convert_torch2numpy_batch(orig_imgs) if self.keep_orig_img else [None] * B
        else:
            shapes = [tuple(o.shape[:2]) for o in orig_imgs]
            imgs = orig_imgs
        # the (pad, gain, original size) table of scale_boxes: one small device tensor per distinct shape tuple, cached
        # (building and uploading it cost more host time than launching the whole network graph)
        key = (tuple(img.shape[2:]), tuple(shapes), str(device))
        cache = self.__dict__.setdefault("_scale_cache", {})
        scale = cache.get(key)
        if scale is None:
            sc = []
            for h0, w0 in shapes:
                gain, pad = ops.scale_boxes_params(tuple(img.shape[2:]), (h0, w0))
                sc.append([pad[0], pad[1], gain, w0, h0])
            if len(cache) > 64:
                cache.clear()
            scale = cache[key] = torch.tensor(sc, dtype=torch.float32, device=device)
        return shapes, imgs, scale

    def _detect_fused(self, im, orig_imgs):
        """inference + postprocess in one pass: the (B, 4+nc, A) tensor is never materialised."""
        return self._harvest_fused(self._launch_fused(im, orig_imgs, slot=0, stream=None))

    def _launch_fused(self, im, orig_imgs, slot: int, stream, after=None):
        """Enqueue one batch on engine slot ``slot`` (on ``stream`` when given) and return what ``_harvest_fused`` needs;
        no host synchronisation.  ``after``: an event that already orders the input in front of this batch (then the batch's
        stream waits for it instead of for the current stream's tail)."""
        b, _, h, w = im.shape
        # (the one-batch-at-a-time path takes the latency profile of the program, the pipelined path the throughput profile:
        # nn/lower.py one_batch; YL_PREDICT_ONE_BATCH_PROFILE=0: one engine for both)
        eng = self.model.engine_for(b, h, w, calib=im.contiguous(),
                                    one_batch=stream is None and os.environ.get("YL_PREDICT_ONE_BATCH_PROFILE", "1") != "0")
        shapes, imgs, scale = self._shapes_and_scale(im, orig_imgs, b, im.device)
        # ONE hipGraph launch per batch (reference model.py:89-99 -> predictor.py:149-229 is ~320 module calls): the
        # input and the scale table are staged into engine-owned buffers whose addresses the graph baked in
        # (YL_PREDICT_GRAPH=0: the same kernels launched one by one)
        graph = os.environ.get("YL_PREDICT_GRAPH", "1") != "0"
        imc = im.contiguous()
        # the graph reads the caller's tensor in place: its address is written into the stem launch's node before the
        # launch (YL_PREDICT_REBIND=0: one device-to-device copy into an engine-owned input buffer instead, 39 MB per batch
        # of 32 at 640x640).  The job keeps the tensor alive until the batch is harvested.
        stage = "rebind" if graph and os.environ.get("YL_PREDICT_REBIND", "1") != "0" and not getattr(self, "_no_rebind", False) else graph

        def go():
            # (one batch per call: the counts land in pinned host memory straight from the last kernel -- the harvest
            # synchronises the stream and reads them, no device-to-host copy; YL_PREDICT_HOST_COUNTS=0: the .tolist() of a
            # device tensor instead)
            host_counts = graph and os.environ.get("YL_PREDICT_HOST_COUNTS", "1") != "0"
            kw = dict(max_det=self.args.max_det, classes=self.args.classes, agnostic=self.args.agnostic_nms, scale=scale,
                      graph=graph, slot=slot, count_host=host_counts, scale_static=True)
            try:
                out, cnt = eng.detect(imc, self.args.conf, self.args.iou, stage=stage, **kw)
            except RuntimeError as e:
                if stage != "rebind" or "rebind" not in str(e):
                    raise
                # a runtime that cannot update the captured graph's node: the staging copy instead (same kernels, same results)
                LOGGER.warning(f"input rebinding unavailable ({e}); staging the input through an engine-owned buffer")
                self._no_rebind = True
                out, cnt = eng.detect(imc, self.args.conf, self.args.iou, stage=True, **kw)
            # engine-owned output buffers are overwritten by the slot's next batch: ONE copy of the padded records,
            # ordered behind this batch on its stream
            if not graph:
                return out, cnt
            if host_counts and stream is None:
                return out, cnt  # (cloned by the harvest, behind the synchronisation: not on the path the caller waits for)
            if host_counts:
                return out.clone(), cnt  # (pipelined: the slot's pinned counts are read once the batch's event has fired)
            if stream is None:
                return out.clone(), cnt
            # the per-image counts travel to PINNED host memory on the batch's own stream: the harvest then waits for the
            # batch's event only (a .tolist() of the device tensor is a blocking copy on the null stream, which queued
            # behind every batch in flight: 1.2 ms per batch)
            pins = self.__dict__.setdefault("_cnt_pins", {})
            key = (slot, int(cnt.numel()))
            if key not in pins:
                pins[key] = torch.empty(cnt.numel(), dtype=cnt.dtype, pin_memory=True)
            pins[key].copy_(cnt, non_blocking=True)
            return out.clone(), pins[key]

        ev = None
        if stream is None:
            out, cnt = go()
        else:
            if after is not None:
                stream.wait_event(after)
            elif os.environ.get("YL_PREDICT_NOWAIT", "0") != "1":  # (development switch)
                stream.wait_stream(torch.cuda.current_stream(im.device))  # the input was produced on the caller's stream
            with torch.cuda.stream(stream):
                out, cnt = go()
                ev = torch.cuda.Event()
                ev.record(stream)
            imc.record_stream(stream)
        return dict(out=out, cnt=cnt, ev=ev, shapes=shapes, imgs=imgs, paths=self.batch[0], n=b, input=imc,
                    host_counts=stream is None and not cnt.is_cuda)

    def _harvest_fused(self, job):
        if job["ev"] is not None:
            job["ev"].synchronize()
            # the records were cloned on the batch's side stream and are consumed on the caller's stream from here on:
            # tell the caching allocator, so the block is not handed out again under a pending reader
            job["out"].record_stream(torch.cuda.current_stream(job["out"].device))
        if job.get("host_counts"):
            torch.cuda.current_stream(job["out"].device).synchronize()  # the one host sync of the batch
            counts = job["cnt"].tolist()
            job["out"] = job["out"].clone()  # engine-owned buffers are overwritten by the slot's next batch
        else:
            counts = job["cnt"].tolist()  # the one host sync of the batch
        self.last_preds = None
        names, out, shapes, imgs, paths = self.model.names, job["out"], job["shapes"], job["imgs"], job["paths"]
        recs = out.unbind(0)  # one call: the per-image (max_det, 6) records; rows are cut to the count on first use
        return [Results(imgs[i], path=paths[i], names=names, boxes=Boxes.from_record(recs[i], counts[i], shapes[i]),
                        orig_shape=shapes[i]) for i in range(job["n"])]

    def _inflight(self) -> int:
        """Batches kept in flight by ``stream_inference`` on sources of more than one batch (YL_PREDICT_INFLIGHT, default
        4; export GPU_MAX_HW_QUEUES to the same number before the first GPU call: one hardware queue per stream)."""
        n = int(os.environ.get("YL_PREDICT_INFLIGHT", "4"))
        dev = torch.device(self.device) if self.device is not None else None
        return max(1, n) if (dev is not None and dev.type == "cuda") else 1

    # ------------------------------------------------------------------ driver
    def __call__(self, source=None, model=None, stream=False, *args, **kwargs):
        self.stream = stream
        if stream:
            return self.stream_inference(source, model, *args, **kwargs)
        # one inference-mode context around the whole call and one generator step per BATCH: resuming a decorated
        # generator re-enters the context per yielded Results (33 times per batch of 32, ~35 us of host time)
        with torch.inference_mode():
            out = []
            for batch_results in self._batches(source, model, *args, **kwargs):
                out.extend(batch_results)
            return out

    def setup_source(self, source):
        self.imgsz = check_imgsz(self.args.imgsz, stride=self.model.stride, min_dim=2)
        self.dataset = load_inference_source(source=source, batch=self.args.batch, vid_stride=self.args.vid_stride,
                                             buffer=self.args.stream_buffer)
        self.source_type = self.dataset.source_type

    def setup_model(self, model, verbose=True):
        self.model = AutoBackend(weights=model or self.args.model,
                                 device=select_device(self.args.device, verbose=verbose and self.args.verbose),
                                 dnn=self.args.dnn, data=self.args.data, fp16=self.args.half, batch=self.args.batch,
                                 fuse=True, verbose=verbose)
        self.device = self.model.device
        self.args.half = self.model.fp16
        self.model.eval()

    @smart_inference_mode()
    def stream_inference(self, source=None, model=None, *args, **kwargs):
        """Generator of Results, one image at a time (reference predictor.py:149-229)."""
        for batch_results in self._batches(source, model, *args, **kwargs):
            yield from batch_results

    def _batches(self, source=None, model=None, *args, **kwargs):
        """The predict loop: yields the list of Results of one batch at a time."""
        if not self.model:
            self.setup_model(model)
        with self._lock:
            self.setup_source(source if source is not None else self.args.source)
            if self.args.save_txt:
                (self.save_dir / "labels").mkdir(parents=True, exist_ok=True)
            self.seen, self.batch = 0, None
            im = None
            # Sources of several batches are PIPELINED on the fused path: up to S batches are in flight on S engine slots /
            # HIP streams, batch k's Results are built and yielded while batches k + 1 .. k + S - 1 run (the reference
            # generator, predictor.py:149-229, finishes a batch before it touches the next).  Same kernels, same slots'
            # buffers as one batch at a time: results are bit-identical.
            S = self._inflight() if self._can_fuse() and not getattr(self.dataset, "single_batch", False) else 1
            if S > 1 and len(self.__dict__.setdefault("_streams", [])) < S:
                self._streams = [torch.cuda.Stream(device=self.device) for _ in range(S)]
            # stage timers: the reference's Profile synchronises the DEVICE around every stage (ops.py:59-63), which would
            # serialise the batches in flight -- with S > 1 they are plain wall-clock timers (a batch's "inference" time is
            # then its enqueue + the wait for its results)
            # (and on the fused path with one batch at a time: its ONE host synchronisation -- the counts' D2H copy in
            # _harvest_fused -- already ends the "inference" timer with the batch finished; six device synchronisations per call
            # around the three stages cost ~60 us of host time per batch for the same numbers)
            pdev = self.device if S == 1 and not self._can_fuse() else None
            profilers = (ops.Profile(device=pdev), ops.Profile(device=pdev), ops.Profile(device=pdev))
            pending, k = [], 0
            # everything the caller enqueued on the current stream before predict() -- e.g. the kernels that produced the
            # tensors of a tensor source -- is ordered in front of every batch by ONE event recorded here; a batch whose
            # preprocess stage enqueued device work of its own (upload, letterbox, dtype conversion) waits for the current
            # stream's tail instead (a cross-stream wait per batch on the stream that also carries the per-batch uploads
            # cost 5 % of the pipelined rate)
            entry_ev = None
            if S > 1:
                entry_ev = torch.cuda.Event()
                entry_ev.record(torch.cuda.current_stream(self.device))

            def finish(job):
                self.batch = job["batch"]
                paths, im0s, s = self.batch
                with profilers[1]:
                    self.results = self._harvest_fused(job)
                profilers[1].dt += job["t_launch"]
                profilers[2].dt = 0.0
                return self._emit(paths, im0s, s, job["im"], profilers, job["t_pre"])

            for self.batch in self.dataset:
                paths, im0s, s = self.batch
                with profilers[0]:
                    im = self.preprocess(im0s)
                if getattr(self.args, "profile", False) and not self.__dict__.get("_profiled", False):
                    # predict(profile=True): the per-layer table of the first batch (reference nn/tasks.py:155-177 prints it from
                    # inside the layer loop; here the launches are timed with HIP events by yl_engine_profile_detect)
                    self._profiled = True
                    self.layer_profile = self.model.profile_layers(im if im.dtype != torch.float64 else im.float())
                if self._can_fuse():
                    with profilers[1]:  # network + fused decode / NMS / box rescale, one host sync
                        # (a non-contiguous tensor gets a .contiguous() copy enqueued on the CURRENT stream in
                        # _launch_fused: that batch must wait for the current stream's tail, not for the entry event)
                        untouched = (isinstance(im0s, torch.Tensor) and im is im0s and im.is_contiguous()
                                     and getattr(self.dataset, "resident", False))
                        job = self._launch_fused(im, im0s, slot=k % S, stream=self._streams[k % S] if S > 1 else None,
                                                 after=entry_ev if untouched else None)
                    job.update(batch=self.batch, im=im, t_launch=profilers[1].dt, t_pre=profilers[0].dt)
                    pending.append(job)
                    k += 1
                    if len(pending) >= S:
                        yield finish(pending.pop(0))
                    continue
                with profilers[1]:
                    preds = self.inference(im, *args, **kwargs)
                with profilers[2]:
                    self.results = self.postprocess(preds, im, im0s)
                yield self._emit(paths, im0s, s, im, profilers, profilers[0].dt)
            while pending:
                yield finish(pending.pop(0))
        if self.args.verbose and self.seen and im is not None:
            t = tuple(x.t / self.seen * 1e3 for x in profilers)
            LOGGER.info("Speed: %.1fms preprocess, %.1fms inference, %.1fms postprocess per image at shape "
                        f"{(min(self.args.batch, self.seen), 3, *im.shape[2:])}" % t)

    def _emit(self, paths, im0s, s, im, profilers, t_pre):
        """Per-image bookkeeping of one finished batch (speed, sinks, log line); returns its Results."""
        n = len(im0s)
        speed = {"preprocess": t_pre * 1e3 / n, "inference": profilers[1].dt * 1e3 / n, "postprocess": profilers[2].dt * 1e3 / n}
        sinks = self.args.verbose or self.args.save_txt or self.args.save or self.args.save_crop
        self.seen += n
        for i, r in enumerate(self.results[:n]):
            r.speed = dict(speed)  # (one dict per Results, as in the reference: callers may annotate theirs)
            if sinks:
                s[i] += self.write_results(i, Path(paths[i]), im, s)
        if self.args.verbose:
            LOGGER.info("\n".join(s))
        return self.results

    def write_results(self, i, p, im, s):
        """Log line (and optional labels/*.txt) for image i (reference predictor.py:248-285)."""
        string = f"{i}: " if (self.source_type.from_img or self.source_type.tensor) else ""
        string += "{:g}x{:g} ".format(*im.shape[2:])
        result = self.results[i]
        string += f"{result.verbose()}{result.speed['inference']:.1f}ms"
        if self.args.save_txt:
            self.txt_path = self.save_dir / "labels" / p.stem
            result.save_dir = str(self.save_dir)
            result.save_txt(f"{self.txt_path}.txt", save_conf=self.args.save_conf)
        # image sinks (reference predictor.py:270-283): only when the source image is at hand (image sources;
        # tensor sources keep none unless keep_orig_img=True)
        if result.orig_img is not None and (self.args.save or self.args.save_crop):
            result.save_dir = str(self.save_dir)
            if self.args.save_crop:
                result.save_crop(save_dir=self.save_dir / "crops", file_name=p.stem)
            if self.args.save:
                self.save_dir.mkdir(parents=True, exist_ok=True)
                result.save(str(self.save_dir / p.name), conf=self.args.show_conf, labels=self.args.show_labels,
                            boxes=self.args.show_boxes, line_width=self.args.line_width)
        return string
