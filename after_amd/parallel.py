"""Clip-batch sharding across the GPUs of one node (one process per GPU,
torch.distributed with backend "nccl" = RCCL over xGMI).

The sampling path has no exchange step: every clip is independent
through encode / sample / decode, so the only traffic is (optionally) one weight
broadcast at start-up and one all-gather of the finished clips -- 2 MiB of audio or
64 KiB of latents per clip.  No all-reduce, no tensor / sequence parallelism.

xGMI is point-to-point (7 links per GPU): the gather is ONE `all_gather_into_tensor`
straight into the preallocated [n_clips, ...] result (no list of per-rank tensors, no
concatenation) whenever the shards are even; ragged shards (n_clips % world != 0) and the
gloo test backend take the padded list form."""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

BUCKET_BYTES = 32 << 20  # small tensors are coalesced up to this; larger ones go in place


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


def _bcast(t: torch.Tensor, src: int) -> None:
    """In-place broadcast of one contiguous tensor."""
    if t.is_cuda and _host_staged():
        host = t.cpu()
        dist.broadcast(host, src=src)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src)


def broadcast_module(module: torch.nn.Module, src: int = 0) -> None:
    """Broadcast every parameter / float buffer of `module` from `src` (start-up only).

    Tensors of a megabyte or more travel IN PLACE (no staging copy); the many small ones
    (biases, norm scales, BatchNorm statistics) are coalesced into buckets of at most
    BUCKET_BYTES, so a 112-MB codec costs a handful of collectives and no 112-MB temporary.
    Afterwards every handle-owning sub-module is refreshed: the HIP handles keep their own
    re-laid-out weight copies and must not serve the pre-broadcast values."""
    if not is_distributed():
        return
    seen, tensors = set(), []
    for t in list(module.parameters()) + list(module.buffers()):
        if t.is_floating_point() and id(t) not in seen:  # aliased parameters (rotary freqs) once
            seen.add(id(t))
            tensors.append(t)
    small: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal small, size
        if not small:
            return
        flat = torch.cat([t.detach().reshape(-1) for t in small])
        _bcast(flat, src)
        off = 0
        with torch.no_grad():
            for t in small:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
        small, size = [], 0

    with torch.no_grad():
        for t in tensors:
            nbytes = t.numel() * t.element_size()
            if nbytes >= (1 << 20) and t.is_contiguous():
                _bcast(t.detach(), src)
                continue
            if size + nbytes > BUCKET_BYTES:
                flush()
            small.append(t)
            size += nbytes
        flush()
    for m in module.modules():
        if hasattr(m, "refresh"):
            m.refresh()


def gather_clips(local: torch.Tensor, n_clips: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """All-gather the ranks' clip shards back into [n_clips, ...] (on every rank).

    `out`: optional preallocated result (reused across steps: nothing is allocated on the hot
    path).  Even shards on RCCL: one all_gather_into_tensor into `out`."""
    if not is_distributed():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes: List[int] = []
    for r in range(world):
        lo, hi = shard_bounds(n_clips, r, world)
        sizes.append(hi - lo)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} clips, expected {sizes[rank]}")
    shape = (n_clips, ) + tuple(local.shape[1:])
    if out is not None and (tuple(out.shape) != shape or out.dtype != local.dtype or out.device != local.device):
        raise ValueError(f"out must be {shape} {local.dtype} on {local.device}")
    even = min(sizes) == max(sizes)
    if even and not (local.is_cuda and _host_staged()):
        if out is None:
            out = local.new_empty(shape)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    # ragged shards / host-staged backend: padded list form
    m = max(sizes)
    pad = local
    if local.shape[0] < m:
        pad = torch.cat([local, local.new_zeros((m - local.shape[0], ) + tuple(local.shape[1:]))])
    dev = pad.device
    if pad.is_cuda and _host_staged():
        pad = pad.cpu()
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad.contiguous())
    res = torch.cat([o[:s] for o, s in zip(parts, sizes)]).to(dev)
    if out is not None:
        out.copy_(res)
        return out
    return res


def ranks_seen() -> int:
    """How many ranks the collective backend actually connected (an all-reduce of ones): lets the
    bench line prove RCCL saw `--gpus` ranks."""
    if not is_distributed():
        return 1
    one = torch.ones(1, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(one)
    return int(one.item())
