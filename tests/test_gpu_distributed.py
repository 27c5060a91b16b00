"""Wavelength-sharded runs on the device: the engine's slab output and the unpack kernel (sk_amd_engine_solve_device_slab, sk_amd_unpack_slab)
against the ordinary outputs and the numpy mirror of the slab layout; a torchrun-launched two-rank run over RCCL where two GPUs exist."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests import scenarios as sc

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _device_inputs(s, lo, hi, dev):
    import torch
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    return (to(s["extinction"][:, lo:hi].T), to(s["ssa"][:, lo:hi].T), to(np.transpose(s["leg"][:, :, lo:hi], (2, 1, 0))), to(s["albedo"][lo:hi]))


@pytest.mark.parametrize("world", [1, 3])
def test_slabs_of_a_partition_unpack_to_the_full_solve(world):
    """every 'rank' of a partition solved in turn on this GPU into its slab, the slabs laid side by side as the all-gather would, unpacked:
    bit-identical to the ordinary full-spectrum outputs (uneven shards: 7 wavelengths over 3 ranks pad the last two slabs)"""
    import torch
    from sasktran2_amd.device import DeviceSolver
    from sasktran2_amd.distributed import pack_slab_host, rows_per_rank, shard_bounds
    from tests.gpu_harness import build_engine
    nw = 7
    s = sc.limb_c2(num_wavelengths=nw)
    engine, _, _ = build_engine(s)
    solver = DeviceSolver(engine)
    dev = torch.device("cuda", 0)
    full = solver.allocate_outputs(nw, derivatives=True, device=dev)
    solver.solve(*_device_inputs(s, 0, nw, dev), full)
    rows, row = rows_per_rank(nw, world), solver.slab_row_doubles(True)
    gathered = torch.zeros((world * rows, row), dtype=torch.float64, device=dev)
    status = torch.zeros(rows, dtype=torch.int32, device=dev)
    for r in range(world):
        lo, hi = shard_bounds(nw, r, world)
        local = gathered[r * rows:(r + 1) * rows]
        solver.solve_slab(*_device_inputs(s, lo, hi, dev), local, status, True)
        # the slab is what the numpy mirror packs from the ordinary outputs of the same wavelengths
        want = pack_slab_host(full["radiance"][lo:hi].cpu().numpy(), full["los_optical_depth"][lo:hi].cpu().numpy(),
                              full["wf_extinction"][:, lo:hi].cpu().numpy(), full["wf_ssa"][:, lo:hi].cpu().numpy(),
                              full["wf_albedo"][lo:hi].cpu().numpy(), rows=rows)
        np.testing.assert_array_equal(local.cpu().numpy()[: hi - lo], want[: hi - lo])
    out = solver.allocate_outputs(nw, derivatives=True, device=dev)
    solver.unpack_slab(gathered, nw, world, rows, out)
    torch.cuda.synchronize()
    for k in ("radiance", "los_optical_depth", "wf_extinction", "wf_ssa", "wf_albedo"):
        np.testing.assert_array_equal(out[k].cpu().numpy(), full[k].cpu().numpy(), err_msg=k)


def test_two_ranks_over_rccl_equal_one_gpu():
    """torchrun, one process per GPU, ShardedSolver (slab solve + ONE all_gather_into_tensor + unpack): every rank ends up with the
    single-GPU result bit for bit.  Needs two GPUs (skipped on the one-GPU test box)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=str(ROOT))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", str(ROOT / "tests" / "dist_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "sharded == single" in res.stdout


def test_sharded_host_entry_point_with_a_one_rank_communicator():
    """sk_amd_engine_calculate_radiance_sharded through the Python surface with a real RCCL communicator of one rank: slicing, packing, the
    all-gather and the unpacking run as they do with more ranks; the result equals the plain call bit for bit"""
    import sasktran2_amd as sk
    from tests.gpu_harness import build_engine
    nw = 6
    s = sc.limb_c2(num_wavelengths=nw)
    engine, config, geometry = build_engine(s)
    atmosphere = sk.Atmosphere(geometry, config, numwavel=nw, calculate_derivatives=True)
    atmosphere.storage.total_extinction[:] = s["extinction"]
    atmosphere.storage.ssa[:] = s["ssa"]
    atmosphere.leg_coeff.a1[:] = s["leg"]
    atmosphere.surface.albedo[:] = s["albedo"]
    plain = engine.calculate_radiance(atmosphere, as_dataset=False)
    comm = sk.Communicator(sk.Communicator.unique_id(), 0, 1)
    sharded = engine.calculate_radiance(atmosphere, as_dataset=False, shard=(comm, 0, 1))
    for k in ("radiance", "los_optical_depth", "wf_extinction", "wf_ssa", "wf_albedo"):
        np.testing.assert_array_equal(sharded[k], plain[k], err_msg=k)


def test_sharded_host_entry_point_partition_arithmetic_on_one_gpu():
    """The sharded host entry point's slicing (offsets > 0 of extinction, ssa, Legendre coefficients, albedo, delta-M fractions, every
    derivative mapping), packing and unpacking with the three ranks of an UNEVEN partition of seven wavelengths played in turn by this
    process (sk_amd_engine_calculate_radiance_sharded_emulated: the entry point's own functions, the all-gather replaced by memcpy): bit
    for bit the plain call -- also with delta-M scaling, where the truncation fractions are sliced too."""
    import sasktran2_amd as sk
    from tests.gpu_harness import build_engine
    nw = 7
    for delta_m in (False, True):
        s = sc.limb_c2(num_wavelengths=nw)
        if delta_m:
            s = dict(s, leg=np.concatenate([s["leg"], 0.3 * s["leg"][-4:]], axis=0))   # 20 stored moments for 16 streams
        engine, config, geometry = build_engine(s)
        config.delta_m_scaling = delta_m
        atmosphere = sk.Atmosphere(geometry, config, numwavel=nw, calculate_derivatives=True)
        atmosphere.storage.total_extinction[:] = s["extinction"]
        atmosphere.storage.ssa[:] = s["ssa"]
        atmosphere.leg_coeff.a1[:] = s["leg"]
        atmosphere.surface.albedo[:] = s["albedo"]
        plain = engine.calculate_radiance(atmosphere, as_dataset=False)
        for world in (3, 7):
            sharded = engine.calculate_radiance(atmosphere, as_dataset=False, shard=("emulate", world, -1))
            for k in ("radiance", "los_optical_depth", "wf_extinction", "wf_ssa", "wf_albedo"):
                np.testing.assert_array_equal(sharded[k], plain[k], err_msg=f"{k} world={world} delta_m={delta_m}")


def test_sharded_host_entry_point_reports_a_failing_rank_on_every_rank():
    """A rank whose local solve fails still contributes its (zeroed) slab with a status word; after the exchange EVERY rank returns that
    rank's code and names it -- nobody is left waiting in the collective.  Emulated ranks: (1) a simulated failure of rank 1, (2) a real
    one: a non-physical single-scatter albedo in the last rank's shard only.  More ranks than wavelengths and flux derivative outputs are
    refused before anything is solved."""
    import sasktran2_amd as sk
    from tests.gpu_harness import build_engine
    nw = 5
    s = sc.limb_c2(num_wavelengths=nw)
    engine, config, geometry = build_engine(s)
    atmosphere = sk.Atmosphere(geometry, config, numwavel=nw, calculate_derivatives=True)
    atmosphere.storage.total_extinction[:] = s["extinction"]
    atmosphere.storage.ssa[:] = s["ssa"]
    atmosphere.leg_coeff.a1[:] = s["leg"]
    atmosphere.surface.albedo[:] = s["albedo"]
    with pytest.raises(RuntimeError, match="rank 1 failed"):
        engine.calculate_radiance(atmosphere, as_dataset=False, shard=("emulate", 3, 1))
    atmosphere.storage.ssa[3, nw - 1] = 1.5
    with pytest.raises(RuntimeError, match="rank 2 failed"):
        engine.calculate_radiance(atmosphere, as_dataset=False, shard=("emulate", 3, -1))
    atmosphere.storage.ssa[3, nw - 1] = s["ssa"][3, nw - 1]
    with pytest.raises(RuntimeError, match="more ranks than wavelengths"):
        engine.calculate_radiance(atmosphere, as_dataset=False, shard=("emulate", nw + 1, -1))
    ok = engine.calculate_radiance(atmosphere, as_dataset=False, shard=("emulate", 2, -1))
    assert np.all(np.isfinite(ok["radiance"]))
