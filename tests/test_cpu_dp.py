"""world_size-2 gloo tests of the data-parallel host logic (trainner_amd/dp.py): bucketed gradient
averaging of a flat buffer, the reverse-order bucket schedule, and the 2-scalar relativistic-sum
exchange.  RCCL replaces gloo on the GPUs; the code path is the same."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    try:
        from trainner_amd import dp as dpmod
        dpmod.BUCKET_FLOATS = 1000                 # force several buckets
        group = dpmod.init_from_env()
        assert group.world_size == world and group.rank == rank
        n = 4500
        base = torch.arange(n, dtype=torch.float32)
        g = base * (rank + 1)
        group.reduce_flat(g)
        group.wait()
        expect = base * (sum(r + 1 for r in range(world)) / world)
        ok1 = torch.allclose(g, expect)

        class P:                                     # stand-in for a parameter inside a FlatParams holder
            def __init__(self, holder, off):
                self._tnr_flat = (holder, off)

        class H:
            pass
        h = H()
        h.total = n
        h.grad = base * (rank + 1)
        sched = dpmod.BucketSchedule(group, h)
        for off in (4000, 2600, 900, 0):            # backward finishes parameters from the end
            sched.mark_done(P(h, off))
        sched.flush()
        group.wait()
        ok2 = torch.allclose(h.grad, expect)
        sums = torch.tensor([1.0 + rank, 2.0, 3.0])
        group.all_reduce_sum(sums)
        ok3 = torch.allclose(sums, torch.tensor([sum(1.0 + r for r in range(world)), 2.0 * world, 3.0 * world]))
        q.put((rank, ok1, ok2, ok3))
    except Exception as e:                           # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_dp_gradient_exchange_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1:] == (True, True, True), r


def test_dp_single_process_is_noop():
    from trainner_amd import dp as dpmod
    g = dpmod.DPGroup()
    t = torch.ones(10)
    g.reduce_flat(t)
    g.wait()
    g.all_reduce_sum(t)
    assert torch.equal(t, torch.ones(10))
