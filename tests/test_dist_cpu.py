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


class _FakeEngine:
    """Stands in for Engine's pd_comm_* methods (the RCCL communicator needs GPUs): the gather is simulated from the
    shards every rank would send, so the host-side padding / trimming / rendezvous logic runs on the CPU."""

    def __init__(self, rank, world, shards=None):
        self.rank, self.world, self.shards, self.joined = rank, world, shards, None

    def comm_new_id(self):
        return bytes(range(128))

    def comm_init(self, comm_id, world, rank):
        assert len(comm_id) == 128
        self.joined = (bytes(comm_id), world, rank)

    def comm_world(self):
        return self.world, self.rank

    def comm_all_gather(self, latents):
        mx = latents.shape[0]
        np.testing.assert_array_equal(np.asarray(latents)[:self.shards[self.rank].shape[0]], self.shards[self.rank])
        padded = [np.concatenate([s, np.zeros((mx - s.shape[0],) + s.shape[1:], s.dtype)]) for s in self.shards]
        out = np.concatenate(padded)
        return out if isinstance(latents, np.ndarray) else torch.from_numpy(out)


@pytest.mark.parametrize("n,world", [(8, 2), (5, 2), (7, 3), (3, 1)])
@pytest.mark.parametrize("as_torch", [False, True])
def test_engine_all_gather_pads_and_trims(n, world, as_torch):
    from prompt_diffusion_amd.dist import engine_all_gather_latents
    full = np.arange(n * 4 * 2 * 2, dtype=np.float32).reshape(n, 4, 2, 2) + 1.0
    spans = [shard_range(n, r, world) for r in range(world)]
    shards = [full[a:b] for a, b in spans]
    sizes = [b - a for a, b in spans]
    for rank in range(world):
        mine = torch.from_numpy(shards[rank]) if as_torch else shards[rank]
        got = engine_all_gather_latents(_FakeEngine(rank, world, shards), mine, sizes=sizes)
        np.testing.assert_array_equal(np.asarray(got), full)
    with pytest.raises(ValueError):
        engine_all_gather_latents(_FakeEngine(0, world, shards), shards[0], sizes=[sizes[0] + 1] + sizes[1:])


def test_engine_comm_file_rendezvous(tmp_path):
    import threading
    from prompt_diffusion_amd.dist import engine_comm_init
    path = str(tmp_path / "comm_id")
    engines = [_FakeEngine(r, 3) for r in range(3)]
    threads = [threading.Thread(target=engine_comm_init, args=(engines[r], r, 3, path)) for r in (2, 1, 0)]  # rank 0 last
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=30)
    assert [e.joined for e in engines] == [(bytes(range(128)), 3, r) for r in range(3)]
    with pytest.raises(TimeoutError):
        engine_comm_init(_FakeEngine(1, 2), 1, 2, str(tmp_path / "never"), timeout_s=0.05)
    solo = _FakeEngine(0, 1)
    engine_comm_init(solo, 0, 1, str(tmp_path / "unused"))
    assert solo.joined == (bytes(range(128)), 1, 0) and not os.path.exists(str(tmp_path / "unused"))
