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


def _fake_compute(im):  # (b, 3, H, W) uint8 -> (out[b, 4, 6], cnt[b])
    b = im.shape[0]
    s = im.float().flatten(1).sum(1)
    out = torch.zeros(b, 4, 6)
    out[:, 0, 0] = s
    out[:, 1, 4] = im[:, 0, 0, 0].float()
    cnt = (im[:, 1, 0, 0] % 5).to(torch.int32)
    return out, cnt


def _worker(rank, world, port, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from yololite_amd.dist import ShardedRunner, shard_bounds

        per = 3
        g = torch.Generator().manual_seed(1234)
        batches = [torch.randint(0, 256, (world * per, 3, 8, 8), generator=g, dtype=torch.uint8) for _ in range(3)]
        runner = ShardedRunner(_fake_compute, per, (3, 8, 8), torch.uint8, torch.device("cpu"))
        last = runner.run(batches if rank == 0 else [None] * 3, steps)
        lo, hi = shard_bounds(world * per, world, rank)
        assert (lo, hi) == (rank * per, (rank + 1) * per)
        if rank == 0:
            outs, cnts = last
            ref_out, ref_cnt = _fake_compute(batches[(steps - 1) % 3])
            assert torch.equal(torch.cat(outs), ref_out) and torch.equal(torch.cat(cnts), ref_cnt)
        # resident shards (what bench.py runs for N > 1): no per-step collective, one gather at the end
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
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("steps", [1, 4])
def test_sharded_runner_world2(steps):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_shard_bounds_errors():
    from yololite_amd.dist import shard_bounds

    assert shard_bounds(256, 8, 3) == (96, 128)
    with pytest.raises(ValueError):
        shard_bounds(30, 8, 0)
