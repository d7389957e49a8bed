"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL
on ROCm).  Radar frames are independent, so the path shards with NO data-path collective: every rank classifies
its contiguous slab of frames; the only exchange is an all-gather of the per-frame labels (4 B/frame) at the end.
"""
import torch
import torch.distributed as dist


def shard_range(n_frames, rank, world):
    """Contiguous slab [lo, hi) of rank ``rank``: sizes differ by at most one, ranks in order."""
    base, rem = divmod(int(n_frames), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_labels(local, n_frames=None, group=None):
    """All-gather per-frame results (first dim = frames of this rank's slab) from every rank, in rank order.

    Equal slabs use one ``all_gather_into_tensor`` (a single RCCL ring/direct all-gather over xGMI); ragged
    slabs (n_frames not divisible by the world size) are padded to the largest slab and trimmed after."""
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    if world == 1:
        return local
    rank = dist.get_rank(group)
    if n_frames is None:
        sizes = [local.shape[0]] * world
    else:
        sizes = [shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0] for r in range(world)]
    assert local.shape[0] == sizes[rank], "local slab does not match shard_range"
    mx = max(sizes)
    if local.shape[0] < mx:
        pad = torch.zeros((mx - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    # the collective is chosen from the backend up front, the same on every rank (a fallback taken by some ranks only would
    # desynchronise them): RCCL ("nccl") has the flat single-buffer form; gloo gathers into a list
    if dist.get_backend(group) == "nccl":
        out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
    else:
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local, group=group)
        out = torch.cat(parts, dim=0)
    if min(sizes) == mx:
        return out
    return torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)], dim=0)


def classify_sharded(classify_fn, n_frames, make_volumes, group=None):
    """Run ``classify_fn(volumes) -> labels`` on this rank's slab (``make_volumes(lo, hi)`` produces / loads it
    locally -- no input scatter) and return the labels of all ``n_frames`` frames on every rank."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_range(n_frames, rank, world)
    labels = classify_fn(make_volumes(lo, hi))
    return gather_labels(labels, n_frames, group)
