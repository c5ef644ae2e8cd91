"""CPU, world_size 2, gloo: the data-parallel scatter -> compute -> gather pipeline of yololite_amd.dist.

The per-GPU compute is replaced by a pure function of the images so the sharding, the double-buffered
scatter and the ordering of the gathered records can be checked without a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


MAX_DET = 4


def _fake_compute(im):  # (b, 3, H, W) uint8 -> (out[b, 4, 6], cnt[b])
    b = im.shape[0]
    s = im.float().flatten(1).sum(1)
    out = torch.zeros(b, MAX_DET, 6)
    out[:, 0, 0] = s
    out[:, 1, 4] = im[:, 0, 0, 0].float()
    cnt = (im[:, 1, 0, 0] % 5).to(torch.int32)
    return out, cnt


def _worker(rank, world, port, steps, slots, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from yololite_amd.dist import ShardedRunner, pack_views, record_floats, shard_bounds, unpack_record

        per = 3
        g = torch.Generator().manual_seed(1234)
        batches = [torch.randint(0, 256, (world * per, 3, 8, 8), generator=g, dtype=torch.uint8) for _ in range(3)]
        # engine stand-in: one packed record per lane (two per slot), overwritten every time the lane comes round (like the
        # engine-owned static outputs of HipEngine.detect_record); the runner must hand the SAME input buffer to a lane
        recs = [torch.empty(record_floats(per, MAX_DET)) for _ in range(2 * slots)]
        seen_ptr = {}

        def compute(im, slot, lane):
            assert lane % slots == slot and 0 <= lane < 2 * slots
            assert seen_ptr.setdefault(lane, im.data_ptr()) == im.data_ptr(), "a lane's input buffer must not move"
            out, cnt = _fake_compute(im)
            o, c = pack_views(recs[lane], per, MAX_DET)
            o.copy_(out)
            c.copy_(cnt)
            return recs[lane]

        runner = ShardedRunner(compute, per, (3, 8, 8), torch.uint8, torch.device("cpu"), record_floats(per, MAX_DET),
                               slots=slots)
        last = runner.run(batches if rank == 0 else [None] * 3, steps)
        lo, hi = shard_bounds(world * per, world, rank)
        assert (lo, hi) == (rank * per, (rank + 1) * per)
        if rank == 0:
            outs, cnts = unpack_record(last, per, MAX_DET)
            ref_out, ref_cnt = _fake_compute(batches[(steps - 1) % 3])
            assert torch.equal(torch.cat(outs), ref_out) and torch.equal(torch.cat(cnts), ref_cnt)
        else:
            assert last is None
        # resident shards (bench.py's value_resident): no per-step collective, one gather at the end
        from yololite_amd.dist import run_resident

        gl = torch.Generator().manual_seed(77 + rank)
        local = [torch.randint(0, 256, (per, 3, 8, 8), generator=gl, dtype=torch.uint8) for _ in range(2)]
        got = run_resident(_fake_compute, local, steps)
        if rank == 0:
            refs = []
            for r in range(world):
                gr = torch.Generator().manual_seed(77 + r)
                lr = [torch.randint(0, 256, (per, 3, 8, 8), generator=gr, dtype=torch.uint8) for _ in range(2)]
                refs.append(_fake_compute(lr[(steps - 1) % 2]))
            assert torch.equal(torch.cat(got[0]), torch.cat([r[0] for r in refs]))
            assert torch.equal(torch.cat(got[1]), torch.cat([r[1] for r in refs]))
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("steps,slots", [(1, 2), (4, 2), (7, 5), (3, 5)])
def test_sharded_runner_world2(steps, slots):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, steps, slots, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_packed_record_views():
    from yololite_amd.dist import pack_views, record_floats, unpack_record

    rec = torch.zeros(record_floats(2, 3))
    o, c = pack_views(rec, 2, 3)
    assert o.shape == (2, 3, 6) and c.shape == (2,) and c.dtype == torch.int32
    o[1, 2, 5] = 7.0
    c[1] = 3
    assert rec[2 * 3 * 6 - 1] == 7.0 and rec[-1:].view(torch.int32).item() == 3  # aliases, counts as bit patterns
    outs, cnts = unpack_record([rec], 2, 3)
    assert outs[0][1, 2, 5] == 7.0 and cnts[0].tolist() == [0, 3]


def test_shard_bounds_errors():
    from yololite_amd.dist import shard_bounds

    assert shard_bounds(256, 8, 3) == (96, 128)
    with pytest.raises(ValueError):
        shard_bounds(30, 8, 0)


# ------------------------------------------------------------------ BASELINE.json config 4's own shapes, eight ranks (gloo)
C4_PER, C4_DET, C4_HW = 32, 300, 640


def _record_compute(im):
    """(32, 3, 640, 640) uint8 -> a REAL-sized result record's contents (out[32, 300, 6], cnt[32]) as a pure function of the
    images: per image a count in 0..300 and that many rows derived from the pixels, score-descending like the engine's."""
    b = im.shape[0]
    cnt = (im[:, 0, 0, :8].to(torch.int32).sum(1) % (C4_DET + 1)).to(torch.int32)
    out = torch.zeros(b, C4_DET, 6)
    base = im[:, :, ::80, ::80].float().flatten(1)[:, :6]                      # six pixel samples per image
    rows = torch.arange(C4_DET, dtype=torch.float32).view(1, C4_DET, 1)
    full = base.view(b, 1, 6) + rows                                           # row r = samples + r
    full[:, :, 4] = 1.0 - rows[:, :, 0] / C4_DET                               # descending "confidence"
    mask = (torch.arange(C4_DET).view(1, C4_DET) < cnt.view(b, 1)).view(b, C4_DET, 1)
    out = torch.where(mask, full, out)
    return out, cnt


def _worker_c4(rank, world, port, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from yololite_amd.dist import ShardedRunner, pack_views, record_floats, unpack_record

        slots, words = 2, record_floats(C4_PER, C4_DET)
        assert words == C4_PER * C4_DET * 6 + C4_PER == 57632  # 230 528 bytes per rank and step
        batches = [None, None]
        if rank == 0:  # the root owns the uint8 global batches: 256 x 3 x 640 x 640 = 315 MB each
            g = torch.Generator().manual_seed(4)
            batches = [torch.randint(0, 256, (world * C4_PER, 3, C4_HW, C4_HW), generator=g, dtype=torch.uint8) for _ in range(2)]
        recs = [torch.empty(words) for _ in range(2 * slots)]

        def compute(im, slot, lane):
            assert tuple(im.shape) == (C4_PER, 3, C4_HW, C4_HW) and im.dtype == torch.uint8
            out, cnt = _record_compute(im)
            o, c = pack_views(recs[lane], C4_PER, C4_DET)
            o.copy_(out)
            c.copy_(cnt)
            return recs[lane]

        runner = ShardedRunner(compute, C4_PER, (3, C4_HW, C4_HW), torch.uint8, torch.device("cpu"), words, slots=slots)
        last = runner.run(batches, steps)
        if rank == 0:
            assert len(last) == world and all(t.numel() == words for t in last)
            outs, cnts = unpack_record(last, C4_PER, C4_DET)
            ref_out, ref_cnt = _record_compute(batches[(steps - 1) % 2])  # the 1-process answer on the whole global batch
            assert torch.equal(torch.cat(cnts), ref_cnt) and torch.equal(torch.cat(outs), ref_out)
            assert int(ref_cnt.max()) > 200 and int(ref_cnt.min()) < 100  # (records of very different fill, all 300 rows travel)
        else:
            assert last is None
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_sharded_runner_world8_config4_shapes():
    """BASELINE.json config 4 at its own size with eight ranks on gloo: the root scatters a 256 x 3 x 640 x 640 uint8 global batch
    as eight 32-image slices (39 MB per peer), every rank fills a full ``record_floats(32, 300)`` record (230 KB), the root gathers
    eight of them per step; the gathered records equal the one-process answer on the whole batch, in rank order, after three
    steps on two slots (four lanes: a lane buffer is reused once).  The per-GPU compute is a stand-in (no GPU here); the
    transport, the lane schedule and the record layout are the ones bench.py --gpus 8 runs."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c4, args=(r, world, port, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res
