"""Clip-batch sharding across the GPUs of one node (one process per GPU,
torch.distributed with backend "nccl" = RCCL over xGMI).

The sampling path has no exchange step: every clip is independent
through encode / sample / decode, so the only traffic is (optionally) one weight
broadcast at start-up and one all-gather of the finished clips -- 2 MiB of audio or
64 KiB of latents per clip.  No all-reduce, no tensor / sequence parallelism."""
from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_clips: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of this rank's clips (first n % world ranks get one more)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n_clips, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _host_staged() -> bool:
    """gloo has no device path for these collectives: stage GPU tensors through the host (CPU tests,
    and the AFTER_BENCH_SHARE_GPU test hook of bench.py).  RCCL ("nccl") works on device memory."""
    return dist.get_backend() == "gloo"


def broadcast_module(module: torch.nn.Module, src: int = 0) -> None:
    """One bucketed broadcast of every parameter / float buffer from `src` (start-up only)."""
    if not is_distributed():
        return
    tensors = [t for t in list(module.parameters()) + list(module.buffers())
               if t.is_floating_point()]
    if not tensors:
        return
    flat = torch.cat([t.detach().reshape(-1) for t in tensors])
    if flat.is_cuda and _host_staged():
        host = flat.cpu()
        dist.broadcast(host, src=src)
        flat = host.to(flat.device)
    else:
        dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    if hasattr(module, "refresh"):
        module.refresh()


def gather_clips(local: torch.Tensor, n_clips: int) -> torch.Tensor:
    """All-gather the ranks' clip shards (possibly ragged) back into [n_clips, ...]."""
    if not is_distributed():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes: List[int] = []
    for r in range(world):
        lo, hi = shard_bounds(n_clips, r, world)
        sizes.append(hi - lo)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} clips, expected {sizes[rank]}")
    m = max(sizes)
    pad = local
    if local.shape[0] < m:
        pad = torch.cat([local, local.new_zeros((m - local.shape[0], ) + tuple(local.shape[1:]))])
    dev = pad.device
    if pad.is_cuda and _host_staged():
        pad = pad.cpu()
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous())
    return torch.cat([o[:s] for o, s in zip(out, sizes)]).to(dev)
