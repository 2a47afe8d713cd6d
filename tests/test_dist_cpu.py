"""Multi-process (gloo, world_size 2) checks of the batch-shard + all-gather path.  CPU only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from prompt_diffusion_amd.dist import all_gather_latents, shard_batch, shard_range


def test_shard_range_covers_batch_in_order():
    for n in (1, 7, 8, 64, 100):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.arange(n * 4 * 2 * 2, dtype=np.float32).reshape(n, 4, 2, 2)
    mine = shard_batch({"x": full}, rank, world)["x"]
    # stand-in for the per-rank sampling result: a rank-independent function of the shard
    got = all_gather_latents(torch.from_numpy(mine * 2.0 + 1.0))
    # with the shard sizes known up front (shard_range is deterministic) it is a single collective
    known = [b - a for a, b in (shard_range(n, r, world) for r in range(world))]
    got2 = all_gather_latents(torch.from_numpy(mine * 2.0 + 1.0), sizes=known)
    assert torch.equal(got, got2)
    q.put((rank, got.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [8, 5])
def test_all_gather_world2_gloo(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.arange(n * 4 * 2 * 2, dtype=np.float32).reshape(n, 4, 2, 2) * 2.0 + 1.0
    for r in range(2):
        np.testing.assert_array_equal(res[r], want)
