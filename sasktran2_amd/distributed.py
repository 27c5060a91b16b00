"""Wavelength sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL).

Wavelengths are independent (the reference threads over them, cpp/lib/engine/engine.cpp:610-622), so the only exchange
is the final reassembly of the radiance / weighting-function block: one all_gather per output array.  Shards are
contiguous wavelength ranges, so the gathered result is already wavelength ordered.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(nwavel: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, near-equal wavelength range of `rank` (first `nwavel % world` ranks get one more)."""
    base, extra = divmod(nwavel, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_wavelength_axis(local: torch.Tensor, nwavel: int, axis: int = 0) -> torch.Tensor:
    """All-gather `local` (this rank's wavelength slice along `axis`) into the full-spectrum tensor on every rank."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    moved = local.movedim(axis, 0).contiguous()
    sizes = [shard_bounds(nwavel, r, world) for r in range(world)]
    maxn = max(e - s for s, e in sizes)
    pad = moved
    if moved.shape[0] < maxn:
        pad = torch.zeros((maxn,) + tuple(moved.shape[1:]), dtype=moved.dtype, device=moved.device)
        pad[: moved.shape[0]] = moved
    out = torch.empty((world * maxn,) + tuple(moved.shape[1:]), dtype=moved.dtype, device=moved.device)
    dist.all_gather_into_tensor(out, pad) if hasattr(dist, "all_gather_into_tensor") and moved.is_cuda else \
        dist.all_gather(list(out.view((world, maxn) + tuple(moved.shape[1:])).unbind(0)), pad)
    pieces = [out[r * maxn: r * maxn + (e - s)] for r, (s, e) in enumerate(sizes)]
    return torch.cat(pieces, 0).movedim(0, axis)
