"""Data-parallel batched inference: one process per GPU, images sharded across ranks.

The reference has no distributed code at all (``utils/torch_utils.py:142-159`` ends up on
``cuda:0``); the path shards trivially -- images are independent through the network and NMS is per image
(reference ``utils/ops.py:219``) -- so weights are replicated and the only exchanges are (1) root -> peers input
slices and (2) peers -> root fixed-size result records.  Both go through ``torch.distributed`` (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  There is no all-reduce.

xGMI is point-to-point: the root's scatter uses its 7 links concurrently, so its cost is about one slice /
153 GB/s (uint8 32x3x640x640 = 39 MB -> ~0.26 ms) and is hidden behind the compute of the batches already in
flight by issuing it ``slots`` steps ahead (``ShardedRunner.run``).

Two transports carry those exchanges: ``torch.distributed`` (the default of ``ShardedRunner``; "nccl" = RCCL on the GPU
box, "gloo" in the CPU tests) and ``NativeComm`` -- the ``yl_comm_*`` group of the C ABI (include/yl_amd.h), RCCL send / recv
enqueued on the caller's HIP stream with no torch type in sight, which is what a non-Python binder of libyl_amd.so uses.

Result records: one step of one rank returns ONE packed tensor of ``record_floats(b, max_det)`` fp32 words --
``b * max_det * 6`` box words ``[x1, y1, x2, y2, conf, cls]`` followed by the ``b`` per-image counts stored as
int32 bit patterns -- so a step costs one gather, not two (``pack_views`` / ``unpack_record``).
"""
from __future__ import annotations

import contextlib
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

RECORD = 6  # x1, y1, x2, y2, conf, cls


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, equal image ranges; ``total`` must divide evenly (reference select_device enforces
    the same for multi-GPU strings, torch_utils.py:142-152)."""
    if total % world:
        raise ValueError(f"'batch={total}' must be a multiple of GPU count {world}.")
    per = total // world
    return rank * per, (rank + 1) * per


def record_floats(b: int, max_det: int) -> int:
    """fp32 words of one rank's packed result record of one step."""
    return b * max_det * RECORD + b


def pack_views(rec: torch.Tensor, b: int, max_det: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(out[b, max_det, 6] fp32, cnt[b] int32) views aliasing a packed record (no copy)."""
    n = b * max_det * RECORD
    return rec[:n].view(b, max_det, RECORD), rec[n: n + b].view(torch.int32)


def unpack_record(recs: Sequence[torch.Tensor], b: int, max_det: int) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """Per-rank packed records -> (outs, cnts) lists ordered by rank."""
    pairs = [pack_views(r, b, max_det) for r in recs]
    return [p[0] for p in pairs], [p[1] for p in pairs]


def scatter_images(global_batch: Optional[torch.Tensor], local: torch.Tensor, src: int = 0, async_op: bool = False,
                   group=None):
    """Root splits ``global_batch`` (world*b, 3, H, W) along dim 0; every rank receives into ``local``."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    chunks = None
    if rank == src:
        if global_batch is None or global_batch.shape[0] != world * local.shape[0]:
            raise ValueError("root must pass the full batch")
        chunks = list(global_batch.chunk(world, 0))
    return dist.scatter(local, chunks, src=src, async_op=async_op, group=group)


def gather_record(rec: torch.Tensor, into: Optional[List[torch.Tensor]], dst: int = 0, async_op: bool = False, group=None):
    """Peers send their packed record; the root receives into ``into`` (one tensor per rank, rank order)."""
    return dist.gather(rec, into if dist.get_rank(group) == dst else None, dst=dst, async_op=async_op, group=group)


def gather_results(out: torch.Tensor, cnt: torch.Tensor, dst: int = 0):
    """Blocking convenience form: (b, max_det, 6) boxes and (b,) counts of every rank, as lists on the root."""
    b, max_det = out.shape[0], out.shape[1]
    rec = torch.empty(record_floats(b, max_det), dtype=torch.float32, device=out.device)
    o, c = pack_views(rec, b, max_det)
    o.copy_(out)
    c.copy_(cnt)
    world, rank = dist.get_world_size(), dist.get_rank()
    into = [torch.empty_like(rec) for _ in range(world)] if rank == dst else None
    gather_record(rec, into, dst)
    if rank != dst:
        return None, None
    return unpack_record(into, b, max_det)


def run_resident(compute: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]], local_batches: List[torch.Tensor],
                 steps: int, dst: int = 0, finish: Optional[Callable[[], None]] = None):
    """Every rank works on image shards that already live in its own HBM: ``steps`` passes of the hot
    path with no data-path collective at all, then ONE gather of the last step's result records to ``dst``
    (the replica-throughput figure ``value_resident`` of bench.py).  Returns the gathered (outs, cnts) on ``dst``.
    ``compute`` may pipeline steps over several streams; ``finish`` (e.g. a device synchronise) is then
    called once after the last step, before the gather reads its outputs."""
    out = cnt = None
    for k in range(steps):
        out, cnt = compute(local_batches[k % len(local_batches)])
    if finish is not None:
        finish()
    return gather_results(out, cnt, dst)


class NativeComm:
    """The ``yl_comm_*`` group of libyl_amd.so (include/yl_amd.h): scatter / gather / barrier as grouped RCCL send / recv on a
    HIP stream the caller names.  No CPU transport: construction fails without RCCL or a GPU.

    ``exchange(payload)`` hands rank 0's 128-byte id to every rank: it is called with the id on rank 0 and with None
    elsewhere and returns the id everywhere -- ``id_via_torch`` (any initialised ``torch.distributed`` group, gloo will do)
    and ``id_via_file`` are the two stock ways; a C binder would use its own rendezvous."""

    def __init__(self, rank: int, world: int, exchange: Callable[[Optional[bytes]], bytes], device: Optional[torch.device] = None):
        import ctypes as C

        from . import _lib

        self._L = _lib.lib()
        self._h = None
        self.rank, self.world = int(rank), int(world)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        ident = None
        if self.rank == 0:
            buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
            _lib.check(self._L.yl_comm_get_unique_id(buf), "yl_comm_get_unique_id")
            ident = buf.raw
        ident = exchange(ident)
        if not isinstance(ident, (bytes, bytearray)) or len(ident) != _lib.COMM_ID_BYTES:
            raise ValueError("the id exchange must return rank 0's 128-byte id on every rank")
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._L.yl_comm_create(C.create_string_buffer(bytes(ident), _lib.COMM_ID_BYTES), self.rank, self.world,
                                              C.byref(h)), "yl_comm_create")
        self._h = h
        if hasattr(exchange, "done"):  # (id_via_file: every rank holds a communicator now -- ncclCommInitRank is collective)
            self.barrier()
            exchange.done(self.rank)

    @staticmethod
    def id_via_torch(group=None, src: int = 0) -> Callable[[Optional[bytes]], bytes]:
        def exchange(ident):
            box = [ident]
            dist.broadcast_object_list(box, src=src, group=group)
            return box[0]

        return exchange

    @staticmethod
    def id_via_file(path: str, token: str, timeout_s: float = 120.0) -> Callable[[Optional[bytes]], bytes]:
        """Rank 0 writes ``token`` + its id to ``path``; the others wait for a file that starts with THEIR ``token`` -- a value every
        rank of ONE job shares and no other job does (the launcher's MASTER_PORT + start time, a scheduler's job id).  A file left
        behind by an earlier job, or written by a faster rank 0 of another one, carries another token and is never taken for this
        job's id (a stale id made ncclCommInitRank hang or mismatch without a diagnostic).  ``NativeComm`` removes the file on
        rank 0 once every rank has created its communicator (``exchange.done``)."""
        import os

        tok = token.encode()
        if not tok or b"\n" in tok:
            raise ValueError("the job token must be a non-empty single line")

        def exchange(ident):
            import time

            if ident is not None:
                tmp = f"{path}.tmp{os.getpid()}"
                with open(tmp, "wb") as f:
                    f.write(tok + b"\n" + ident)
                os.replace(tmp, path)  # readers never see a partial id
                return ident
            t0 = time.monotonic()
            while True:
                try:
                    with open(path, "rb") as f:
                        blob = f.read()
                    if blob.startswith(tok + b"\n"):
                        return blob[len(tok) + 1:]
                except FileNotFoundError:
                    pass
                if time.monotonic() - t0 > timeout_s:
                    raise TimeoutError(f"no communicator id for job token {token!r} at {path} after {timeout_s} s")
                time.sleep(0.01)

        def done(rank):
            if rank == 0:
                try:
                    os.unlink(path)
                except FileNotFoundError:
                    pass

        exchange.done = done
        return exchange

    def _stream(self, stream) -> int:
        return (stream if stream is not None else torch.cuda.current_stream(self.device)).cuda_stream

    def scatter(self, global_batch: Optional[torch.Tensor], local: torch.Tensor, src: int = 0, stream=None) -> None:
        """Root: ``global_batch`` = (world * b, ...) contiguous; every rank receives its (b, ...) slice into ``local``."""
        from . import _lib

        if not local.is_contiguous():
            raise ValueError("scatter target must be contiguous")
        nbytes = local.numel() * local.element_size()
        send = 0
        if self.rank == src:
            if (global_batch is None or not global_batch.is_contiguous() or global_batch.dtype != local.dtype
                    or global_batch.numel() != self.world * local.numel()):
                raise ValueError("root must pass the full contiguous batch")
            send = global_batch.data_ptr()
        _lib.check(self._L.yl_comm_scatter(self._h, send, local.data_ptr(), nbytes, src, self._stream(stream)), "yl_comm_scatter")

    def gather(self, rec: torch.Tensor, into: Optional[torch.Tensor], dst: int = 0, stream=None) -> None:
        """Every rank sends ``rec``; the root receives into ``into`` = (world, rec.numel()) contiguous, rank order."""
        from . import _lib

        if not rec.is_contiguous():
            raise ValueError("gather source must be contiguous")
        nbytes = rec.numel() * rec.element_size()
        recv = 0
        if self.rank == dst:
            if into is None or not into.is_contiguous() or into.dtype != rec.dtype or into.numel() != self.world * rec.numel():
                raise ValueError("root must pass a (world, record) contiguous tensor")
            recv = into.data_ptr()
        _lib.check(self._L.yl_comm_gather(self._h, rec.data_ptr(), recv, nbytes, dst, self._stream(stream)), "yl_comm_gather")

    def barrier(self, stream=None) -> None:
        from . import _lib

        _lib.check(self._L.yl_comm_barrier(self._h, self._stream(stream)), "yl_comm_barrier")

    def close(self) -> None:
        if self._h:
            self._L.yl_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeShardedRunner:
    """``ShardedRunner``'s schedule on the ``yl_comm_*`` transport, exchange for exchange: ``depth`` input buffers and records
    per slot; gather k and scatter k + depth * slots are enqueued by libyl_amd.so on the SLOT's OWN stream right behind
    compute k.  Step k + depth * slots runs on the same slot, so stream order alone puts its compute behind that scatter (the
    data has landed), the scatter behind the last reader of its buffer, and the next writer of a record behind its gather: the
    schedule needs no event for its data hazards and NO stream beyond the slots' (a rank's batch streams each want a hardware
    queue of their own, see bench.py's notes on GPU_MAX_HW_QUEUES; round 6: with exchange streams of its own this runner read
    46.2 k images/s at one rank beside 53.7 k resident -- and anything from 43.4 k to 50.7 k as the queue count moved --, with one
    exchange stream 50.9 k).  Records land in ONE (world, words) tensor per lane on the root.  Same contract as ``ShardedRunner.run``.

    Order on a communicator: RCCL's operations on ONE communicator must EXECUTE in the same order on every rank (two sends to
    the same peer are matched in execution order), and operations enqueued on different streams have no order of their own.
    libyl_amd.so keeps that order itself (csrc/comm.hip, the communicator's order chain: an operation that crosses a link waits
    for the event its predecessor on that communicator recorded -- nothing at a world of one); the caller's part is that every
    rank issues the same sequence, which this schedule does.  ``comm_gather``, a SECOND communicator, gives the gathers a chain
    of their own, independent of the scatters'; without it both directions share one chain (issue order gather k,
    scatter k + depth * slots).

    ``root_in_place`` (default): the root computes on ITS slice of the global batch where it lies (``yl_comm_scatter`` with
    ``recv`` = that slice copies nothing) instead of a device-to-device copy of 39 MB per step into the lane buffer; the input
    address of a root lane then cycles through ``len(batches)`` values, each its own captured graph.

    NOT YET RUN ON MORE THAN ONE GPU (no multi-GPU box was available to this build: tests/test_gpu_dist.py holds a two-process
    test that is skipped below two devices); ``ShardedRunner`` on torch.distributed is the default transport of bench.py."""

    def __init__(self, comm: NativeComm, compute: Callable[[torch.Tensor, int, int], torch.Tensor], per_rank: int,
                 shape: Tuple[int, int, int], dtype: torch.dtype, device: torch.device, record_words: int,
                 slots: int = 2, src: int = 0, streams: Optional[Sequence] = None, depth: int = 2,
                 comm_gather: Optional[NativeComm] = None, root_in_place: bool = True):
        if streams is None or len(streams) != slots:
            raise ValueError("one HIP stream per slot")
        if comm_gather is not None and (comm_gather.rank, comm_gather.world) != (comm.rank, comm.world):
            raise ValueError("the gather communicator must span the same ranks")
        self.comm, self.compute, self.src = comm, compute, src
        self.comm_g = comm_gather if comm_gather is not None else comm
        self.per_rank, self.root_in_place = per_rank, bool(root_in_place)
        self.slots, self.lanes = slots, depth * slots
        self.streams = list(streams)
        self.rank, self.world = comm.rank, comm.world
        self.device = device
        self.bufs = [torch.empty((per_rank, *shape), dtype=dtype, device=device) for _ in range(self.lanes)]
        self.gathered = [torch.empty((self.world, record_words), dtype=torch.float32, device=device)
                         if self.rank == src else None for _ in range(self.lanes)]

    def run(self, batches: List[Optional[torch.Tensor]], steps: int) -> Optional[List[torch.Tensor]]:
        nb, S, V = len(batches), self.slots, self.lanes
        recs = [None] * V
        inputs = [None] * V   # what step k's compute reads: the lane buffer, or (root, in place) its slice of the global batch

        def scatter(k: int):
            v, st = k % V, self.streams[k % S]
            gb = batches[k % nb]
            if self.root_in_place and self.rank == self.src:
                inputs[v] = gb[self.src * self.per_rank: (self.src + 1) * self.per_rank]  # (recv == the root's own slice: no copy)
            else:
                inputs[v] = self.bufs[v]
            self.comm.scatter(gb, inputs[v], self.src, stream=st)

        for k in range(min(V, steps)):
            scatter(k)
        last = None
        for k in range(steps):
            j, v = k % S, k % V
            st = self.streams[j]
            with torch.cuda.stream(st):
                recs[v] = self.compute(inputs[v], j, v)
            self.comm_g.gather(recs[v], self.gathered[v], self.src, stream=st)
            if k + V < steps:
                scatter(k + V)
            last = v
        # (a caller that synchronises its slot streams has the gathered records too: the gathers ran on them)
        if last is None or self.rank != self.src:
            return None
        return list(self.gathered[last].unbind(0))


class ShardedRunner:
    """BASELINE.json config 4 / SURVEY 8(e): per step, root scatters the uint8 global batch, every rank runs the
    hot path on its slice, root gathers the packed ``(count, max_det x 6)`` records.

    ``compute(images, slot, lane) -> rec`` is the per-GPU hot path: ``slot`` = the workspace slot / HIP stream the
    step runs on (``slots`` batches are in flight), ``lane`` = which of the runner's ``depth * slots`` input buffers
    ``images`` is -- the SAME tensor every time that lane comes round, so a hipGraph that baked its address stays
    valid -- and the key of the engine-owned packed record the step must write (``HipEngine.detect_record(images, ...,
    slot=slot, rec_key=lane)``).  Step k runs on slot k % slots with lane k % (depth * slots).

    With ``depth`` = 2 buffers and records per slot, the collectives of a slot are TWO of its steps away from the
    compute that needs them, so a slot's stream runs compute after compute with no RCCL operation in between:

      * scatter k + depth * slots is issued right behind compute k on the slot's stream (ordered behind the last reader
        of that buffer: no WAR hazard) and has a whole extra step of the slot to land;
      * compute k waits for scatter k and for gather k - depth * slots, the last reader of the record it will overwrite
        (``Work.wait`` = a stream-side event wait on NCCL, a host wait on gloo) -- both long complete;
      * gather k is issued on the slot's stream right behind compute k.

    (``depth`` = 1 put gather k and scatter k + slots between compute k and compute k + slots of the same stream: with a
    busy GPU the RCCL kernels queue behind running workgroups, and the rehearsal with one rank lost 24 %.)
    ``streams``: one ``torch.cuda.Stream`` per slot on the GPU; None on the CPU (gloo tests).
    """

    def __init__(self, compute: Callable[[torch.Tensor, int, int], torch.Tensor], per_rank: int,
                 shape: Tuple[int, int, int], dtype: torch.dtype, device: torch.device, record_words: int,
                 slots: int = 2, src: int = 0, streams: Optional[Sequence] = None, group=None, depth: int = 2):
        if streams is not None and len(streams) != slots:
            raise ValueError("one stream per slot")
        self.compute = compute
        self.src = src
        self.slots = slots
        self.lanes = depth * slots
        self.group = group
        self.streams = list(streams) if streams is not None else None
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.bufs = [torch.empty((per_rank, *shape), dtype=dtype, device=device) for _ in range(self.lanes)]
        self.gathered: List[Optional[List[torch.Tensor]]] = [
            [torch.empty(record_words, dtype=torch.float32, device=device) for _ in range(self.world)]
            if self.rank == src else None for _ in range(self.lanes)]

    def _on(self, j: int):
        return torch.cuda.stream(self.streams[j]) if self.streams is not None else contextlib.nullcontext()

    def run(self, batches: List[Optional[torch.Tensor]], steps: int) -> Optional[List[torch.Tensor]]:
        """``batches[k % len]`` is the root's global batch for step k (None on other ranks).  Returns the root's
        gathered records (one packed tensor per rank) of the LAST step, None elsewhere.  On the GPU the returned
        tensors are complete once the slot's stream has been synchronised (the caller's timed region ends with a
        device synchronise)."""
        nb, S, V = len(batches), self.slots, self.lanes
        sc = [None] * V
        ga = [None] * V

        def scatter(k: int):
            v = k % V
            sc[v] = scatter_images(batches[k % nb], self.bufs[v], self.src, async_op=True, group=self.group)

        for k in range(min(V, steps)):
            with self._on(k % S):
                scatter(k)
        last = None
        for k in range(steps):
            j, v = k % S, k % V
            with self._on(j):
                sc[v].wait()
                if ga[v] is not None:  # the gather of step k - V still reads the record this step overwrites
                    ga[v].wait()
                rec = self.compute(self.bufs[v], j, v)
                ga[v] = gather_record(rec, self.gathered[v], self.src, async_op=True, group=self.group)
                if k + V < steps:
                    scatter(k + V)
            last = v
        for v in range(V):
            if ga[v] is not None:
                with self._on(v % S):
                    ga[v].wait()
        return self.gathered[last] if (last is not None and self.rank == self.src) else None
