"""Clip sharding + the N>1 collectives on CPU (gloo, world_size 2).  The data path has
no collective; what is exercised here is exactly what bench.py does around it: a
weight broadcast at start-up and the all-gather of (ragged) clip shards."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from after_amd import parallel


def test_shard_bounds_partition():
    for n in (0, 1, 2, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    except Exception as e:  # e.g. the probed port was taken in between: report, do not hang the parent
        q.put((rank, repr(e), None))
        raise
    try:
        torch.manual_seed(rank)  # ranks start with DIFFERENT weights

        class Leaf(torch.nn.Linear):  # a handle-owning module: must be refreshed after the broadcast
            refreshed = 0

            def refresh(self):
                self.refreshed += 1

        # small tensors (bucketed) + one >= 1 MiB weight (in-place path) + a nested handle owner
        net = torch.nn.Sequential(Leaf(5, 3), torch.nn.Sequential(Leaf(600, 500)))
        net[1][0].alias = net[0].bias  # an aliased parameter is sent once
        parallel.broadcast_module(net, src=0)
        assert net[0].refreshed == 1 and net[1][0].refreshed == 1
        assert parallel.ranks_seen() == world
        full = torch.arange(n_clips * 6, dtype=torch.float32).reshape(n_clips, 2, 3)
        local = parallel.shard(full, rank, world) * 1.0
        got = parallel.gather_clips(local, n_clips)
        buf = torch.empty(n_clips, 2, 3)
        again = parallel.gather_clips(local, n_clips, out=buf)  # preallocated result, reused per step
        assert again is buf and torch.equal(buf, got)
        w = torch.cat([net[0].weight.detach().reshape(-1), net[1][0].weight.detach().reshape(-1)])
        # numpy payloads are pickled by value: a torch tensor would travel as a shared-memory handle
        # that dies with this process if the parent has not opened it yet
        q.put((rank, w.numpy().copy(), got.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [4, 5])
def test_broadcast_and_ragged_gather_world2(n_clips):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0, [r[1] for r in res if r[2] is None]
    full = torch.arange(n_clips * 6, dtype=torch.float32).reshape(n_clips, 2, 3).numpy()
    assert (res[0][1] == res[1][1]).all()  # broadcast made the weights identical
    for _, _, got in res:
        assert got.shape == full.shape and (got == full).all()
