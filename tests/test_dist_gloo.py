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
