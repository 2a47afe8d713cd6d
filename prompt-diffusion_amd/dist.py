"""Batch sharding across the GPUs of one node.

Every image is independent through all denoising steps (GroupNorm/LayerNorm are per-sample, a CFG pair
stays on one GPU), so the path shards with NO per-step exchange -- exactly how the reference shards
sampling (``batch_ids = np.arange(100)[rank::world_size]``, eval/evaluate_gen.py:55-57).  The only
collective is one all-gather of the final latents (RCCL over xGMI: backend "nccl" on ROCm; "gloo" in the
CPU tests).  One process per GPU.

Two ways to run that gather: `all_gather_latents` over a torch.distributed group, or -- for a host without torch --
`engine_comm_init` + `engine_all_gather_latents` over the RCCL communicator the engine owns (include/pdengine.h pd_comm_*).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of n items: the first n % world ranks get one extra."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(inputs: Dict[str, "np.ndarray"], rank: int, world: int) -> Dict[str, "np.ndarray"]:
    """Slice every batch-major input ([B, ...]) to this rank's contiguous shard."""
    n = next(iter(inputs.values())).shape[0]
    lo, hi = shard_range(n, rank, world)
    return {k: (v[lo:hi] if v is not None else None) for k, v in inputs.items()}


def all_gather_latents(latents, group=None, sizes=None):
    """Gather [b_r, C, h, w] shards from every rank into [sum b_r, C, h, w] (rank order = batch order).

    `sizes`: the per-rank shard sizes when the caller knows them (shard_range is deterministic) -- then this is exactly
    ONE collective, a direct RCCL all_gather_into_tensor (<= 4 MB total at bs=64, latency-bound).  Without it the sizes
    are exchanged first.  Ragged shards are padded to the largest and trimmed."""
    import torch
    import torch.distributed as dist
    t = latents if isinstance(latents, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(latents))
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    world = dist.get_world_size(group)
    if sizes is None:
        n = torch.tensor([t.shape[0]], device=t.device if dist.get_backend(group) != "gloo" else "cpu", dtype=torch.int64)
        got = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(got, n, group=group)
        sizes = [int(s.item()) for s in got]
    elif len(sizes) != world or sizes[dist.get_rank(group)] != t.shape[0]:
        raise ValueError("sizes must list every rank's shard size")
    mx = max(sizes)
    if t.shape[0] < mx:
        pad = torch.zeros((mx - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad])
    t = t.contiguous()
    dev = t.device
    if dist.get_backend(group) == "gloo":
        t = t.cpu()                       # CPU tests and single-GPU rehearsals: gloo gathers host tensors
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        out = torch.cat(parts).to(dev)
    else:
        out = torch.empty((world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        dist.all_gather_into_tensor(out, t, group=group)
    if all(s == mx for s in sizes):
        return out
    return torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)])


def engine_comm_init(engine, rank: int, world: int, id_path: str, timeout_s: float = 120.0) -> None:
    """Torch-free rendezvous for the engine-owned RCCL communicator: rank 0 writes the 128-byte id to `id_path` (a file every
    rank of the node can read; written under a temporary name and renamed, so a reader never sees half of it), the others
    wait for it, and every rank joins (`Engine.comm_init` is collective).  world == 1 needs no file."""
    import os
    import time
    if world == 1:
        engine.comm_init(engine.comm_new_id(), 1, 0)
        return
    if rank == 0:
        tmp = "%s.%d.tmp" % (id_path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(engine.comm_new_id())
        os.replace(tmp, id_path)
    else:
        t0 = time.monotonic()
        while not os.path.exists(id_path):
            if time.monotonic() - t0 > timeout_s:
                raise TimeoutError("no communicator id at %s after %.0f s" % (id_path, timeout_s))
            time.sleep(0.01)
    with open(id_path, "rb") as f:
        comm_id = f.read()
    engine.comm_init(comm_id, world, rank)


def engine_all_gather_latents(engine, latents, sizes=None):
    """`all_gather_latents` over the engine's communicator: ONE ncclAllGather on the engine's stream.  `sizes` = the per-rank
    shard sizes (shard_range is deterministic, so every rank can compute them); ragged shards are padded to the largest
    and trimmed."""
    world, rank = engine.comm_world()
    n = latents.shape[0]
    if sizes is None:
        sizes = [n] * world
    if len(sizes) != world or sizes[rank] != n:
        raise ValueError("sizes must list every rank's shard size")
    mx = max(sizes)
    is_np = isinstance(latents, np.ndarray)
    if n < mx:
        if is_np:
            latents = np.concatenate([latents, np.zeros((mx - n,) + latents.shape[1:], latents.dtype)])
        else:
            import torch
            latents = torch.cat([latents, torch.zeros((mx - n,) + tuple(latents.shape[1:]), dtype=latents.dtype, device=latents.device)])
    out = engine.comm_all_gather(latents)
    if all(s == mx for s in sizes):
        return out
    parts = [out[r * mx:r * mx + sizes[r]] for r in range(world)]
    if is_np:
        return np.concatenate(parts)
    import torch
    return torch.cat(parts)
