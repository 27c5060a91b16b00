"""Wavelength sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL).

Wavelengths are independent (the reference threads over them, cpp/lib/engine/engine.cpp:610-622), so the only exchange is the final
reassembly of the radiance / weighting-function block: every rank solves its contiguous wavelength range into ONE
wavelength-major slab -- row w = [radiance | los_od | wf_extinction(q) | wf_ssa(q) | wf_albedo] x nlos doubles, written by the engine's
output kernel -- ONE all-gather moves the slabs (208 MB at BASELINE configs[2]; a ring over xGMI, ~1-2 ms) and one kernel unpacks the
gathered block into the reference layouts.  Shards are contiguous, so the gathered rows are already wavelength ordered.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(nwavel: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, near-equal wavelength range of `rank` (first `nwavel % world` ranks get one more)."""
    base, extra = divmod(nwavel, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def rows_per_rank(nwavel: int, world: int) -> int:
    """slab rows every rank contributes to the all-gather: the largest shard (smaller shards pad)"""
    return -(-nwavel // world)


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def all_gather_slab(local: torch.Tensor) -> torch.Tensor:
    """THE collective of a sharded run: [rows_per_rank, row] of every rank -> [world * rows_per_rank, row] on every rank."""
    world = _world()
    if world == 1:
        return local
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.is_cuda:
        dist.all_gather_into_tensor(out, local)
    else:   # gloo (CPU tests)
        dist.all_gather(list(out.view((world,) + tuple(local.shape)).unbind(0)), local)
    return out


def gather_wavelength_axis(local: torch.Tensor, nwavel: int, axis: int = 0) -> torch.Tensor:
    """One array sharded along `axis` -> the full spectrum on every rank (one collective; the radiance-only engines use it: their
    whole result is that one array)."""
    world = _world()
    if world == 1:
        return local
    moved = local.movedim(axis, 0).contiguous()
    rows = rows_per_rank(nwavel, world)
    pad = moved
    if moved.shape[0] < rows:
        pad = torch.zeros((rows,) + tuple(moved.shape[1:]), dtype=moved.dtype, device=moved.device)
        pad[: moved.shape[0]] = moved
    out = all_gather_slab(pad)
    pieces = [out[r * rows: r * rows + (shard_bounds(nwavel, r, world)[1] - shard_bounds(nwavel, r, world)[0])] for r in range(world)]
    return torch.cat(pieces, 0).movedim(0, axis)


class ShardedSolver:
    """Rank-local part of a wavelength-sharded spectrum on the device: `step()` = solve into the slab, all-gather, unpack.
    `out` holds the FULL spectrum on every rank in the reference layouts (DeviceSolver.allocate_outputs)."""

    def __init__(self, solver, nwavel, derivatives=True, device="cuda"):
        self.solver, self.nwavel, self.derivatives = solver, nwavel, derivatives
        self.world = _world()
        self.rank = dist.get_rank() if self.world > 1 else 0
        if self.world > nwavel:   # (the same on every rank: all of them refuse; a rank without wavelengths has nothing to solve)
            raise ValueError(f"more ranks than wavelengths ({self.world} > {nwavel})")
        self.lo, self.hi = shard_bounds(nwavel, self.rank, self.world)
        self.rows = rows_per_rank(nwavel, self.world)
        self.local = torch.zeros((self.rows, solver.slab_row_doubles(derivatives)), dtype=torch.float64, device=device)
        self.status = torch.zeros(self.rows, dtype=torch.int32, device=device)
        self.out = solver.allocate_outputs(nwavel, derivatives=derivatives, device=device)

    def step(self, extinction, ssa, leg, albedo, irradiance=None):
        """inputs: this rank's wavelength range [lo, hi) in the reference storage layout (contiguous float64 CUDA tensors)"""
        self.solver.solve_slab(extinction, ssa, leg, albedo, self.local, self.status, self.derivatives, irradiance)
        gathered = all_gather_slab(self.local)
        return self.solver.unpack_slab(gathered, self.nwavel, self.world, self.rows, self.out)


# ---- the same slab on the host (numpy): the layout's specification, used by the world_size-2 gloo test and by host-side callers
def pack_slab_host(radiance, los_od, wf_extinction=None, wf_ssa=None, wf_albedo=None, rows=None):
    """radiance, los_od, wf_albedo [nw, nlos]; wf_extinction, wf_ssa [ngrid, nw, nlos] -> [rows, fields * nlos] (zero padded)"""
    nw, nlos = radiance.shape
    fields = [radiance[:, None, :], los_od[:, None, :]]
    if wf_extinction is not None:
        fields += [np.transpose(wf_extinction, (1, 0, 2)), np.transpose(wf_ssa, (1, 0, 2)), wf_albedo[:, None, :]]
    slab = np.concatenate(fields, axis=1).reshape(nw, -1)
    if rows is not None and rows > nw:
        slab = np.concatenate([slab, np.zeros((rows - nw, slab.shape[1]))], axis=0)
    return np.ascontiguousarray(slab)


def unpack_slab_host(slab_all, nwavel, world, nlos, ngrid=None):
    """gathered slabs [world * rows, fields * nlos] -> dict of full-spectrum arrays in the reference layouts"""
    rows = slab_all.shape[0] // world
    pieces = [slab_all[r * rows: r * rows + (shard_bounds(nwavel, r, world)[1] - shard_bounds(nwavel, r, world)[0])] for r in range(world)]
    full = np.concatenate(pieces, axis=0).reshape(nwavel, -1, nlos)
    out = {"radiance": full[:, 0, :].copy(), "los_optical_depth": full[:, 1, :].copy()}
    if full.shape[1] > 2:
        ngrid = (full.shape[1] - 3) // 2 if ngrid is None else ngrid
        out["wf_extinction"] = np.transpose(full[:, 2:2 + ngrid, :], (1, 0, 2)).copy()
        out["wf_ssa"] = np.transpose(full[:, 2 + ngrid:2 + 2 * ngrid, :], (1, 0, 2)).copy()
        out["wf_albedo"] = full[:, 2 + 2 * ngrid, :].copy()
    return out
