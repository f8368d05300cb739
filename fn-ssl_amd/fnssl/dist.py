"""Multi-GPU host logic: one process per GPU, utterance shards, no data-path collective.

Every utterance (indeed every mic pair) of the DP-IPD forward is independent
(SURVEY.md §8e), so the batch is split into contiguous utterance shards, each rank
runs the whole path on its shard, and the only communication is an optional
``all_gather`` of the small ``[nb', nt//12, 512]`` outputs when one rank needs all
results (RCCL on MI355X — backend "nccl" — gloo in the CPU tests).  The reference's
equivalents are Lightning DDP sharding the dataloader (main.py:286-288) and
``nn.DataParallel`` scatter/gather (Learner.py:27-28).
"""
from __future__ import annotations

import torch


def shard_bounds(n_items: int, world_size: int, rank: int):
    """Contiguous, balanced [lo, hi) of rank's shard (first n_items % world_size ranks get one extra)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad world_size/rank %d/%d" % (rank, world_size))
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def predict_sharded(predict_fn, batch: torch.Tensor, group=None, gather: bool = True):
    """Run ``predict_fn(batch_shard) -> [n_shard * np, ...]`` on this rank's utterance shard.

    batch: [nb, nch, ns], identical on every rank (or at least its own shard valid).
    Returns the concatenated predictions of all ranks (same on every rank) if
    ``gather`` else this rank's shard.  Ranks with an empty shard contribute nothing.
    """
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(batch.shape[0], world, rank)
    out = predict_fn(batch[lo:hi]) if hi > lo else None
    if not gather or world == 1:
        return out
    # shards may differ by one utterance: gather sizes first, then padded payloads
    n_local = torch.tensor([0 if out is None else out.shape[0]], dtype=torch.int64,
                           device=batch.device if out is None else out.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    tail = None
    for r in range(world):                      # learn the trailing shape from a non-empty rank
        if sizes[r] > 0:
            shape_src = r
            break
    else:
        return None
    tshape = torch.zeros(8, dtype=torch.int64, device=n_local.device)
    if rank == shape_src:
        tshape[0] = out.ndim - 1
        tshape[1:out.ndim] = torch.tensor(out.shape[1:], dtype=torch.int64)
    dist.broadcast(tshape, src=shape_src, group=group)
    tail = tuple(int(v) for v in tshape[1:1 + int(tshape[0])])
    nmax = max(sizes)
    dev = n_local.device
    pad = torch.zeros((nmax,) + tail, dtype=torch.float32, device=dev)
    if out is not None:
        pad[:out.shape[0]] = out
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)
