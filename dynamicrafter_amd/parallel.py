"""Clip-level data parallelism: one process per GPU, independent DDIM trajectories, no data-path collective.

The reference shards by rank-slicing the prompt list — `indices = range(split*rank, split*(rank+1))` with
`split = num_samples // gpu_num` (scripts/evaluation/inference.py:350-356; remainder samples are dropped) — and
initialises NCCL without ever issuing a collective (scripts/evaluation/ddp_wrapper.py:8-12). Here rank 0 owns the
conditioning (it ran the image/text encoders) and scatters each rank's share once over RCCL (xGMI, point-to-point)
before the loop; every rank then runs its own hipGraph-captured loop. Works with backend "gloo" on CPU tensors
(tests) and "nccl" (= RCCL) on the GPUs.
"""
import torch
import torch.distributed as dist


def shard_indices(num_samples, world_size, rank):
    """The reference's rank slice (inference.py:350-356): floor split, tail dropped."""
    split = num_samples // world_size
    return list(range(split * rank, split * (rank + 1)))


def scatter_conditioning(tensors, src=0, group=None):
    """tensors: on rank `src` a dict name -> tensor whose dim 0 enumerates clips (num_clips % world == 0);
    on the other ranks a dict name -> (shape_per_rank, dtype) template or None to learn shapes from `src`.
    Returns this rank's shard dict. One broadcast of metadata + one scatter per tensor."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    meta = [None]
    if rank == src:
        meta[0] = {k: (tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in tensors.items()}
    dist.broadcast_object_list(meta, src=src, group=group)
    out = {}
    ref_dev = None
    if rank == src:
        ref_dev = next(iter(tensors.values())).device
    else:
        ref_dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    for name, (shape, dtype) in meta[0].items():
        n = shape[0]
        if n % world != 0:
            raise ValueError(f"{name}: {n} clips do not divide over {world} ranks")
        per = n // world
        recv = torch.empty((per,) + tuple(shape[1:]), dtype=getattr(torch, dtype), device=ref_dev)
        chunks = None
        if rank == src:
            chunks = [c.contiguous() for c in tensors[name].chunk(world, dim=0)]
        dist.scatter(recv, chunks, src=src, group=group)
        out[name] = recv
    return out


def gather_clips(local, dst=0, group=None):
    """Optional return path: concatenate every rank's decoded clips on `dst` (dim 0)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local.contiguous(), bufs, dst=dst, group=group)
    return torch.cat(bufs, dim=0) if rank == dst else None
