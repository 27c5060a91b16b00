"""world_size-2 gloo test of the wavelength sharding + all-gather reassembly (CPU; the oracle stands in for the solver)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sasktran2_amd.distributed import gather_wavelength_axis, shard_bounds
from tests import scenarios as sc


def test_shard_bounds_cover_everything():
    for nw in (1, 7, 10_000, 10_001):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(nw, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == nw
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1


def _worker(rank, world, port, nw, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.test_oracle_golden import make_oracle
    s = sc.regression_1d(nw)
    lo, hi = shard_bounds(nw, rank, world)
    out = make_oracle(s).run(s["extinction"][:, lo:hi], s["ssa"][:, lo:hi], s["leg"][:, :, lo:hi], s["albedo"][lo:hi],
                             derivatives=True)
    rad = gather_wavelength_axis(torch.from_numpy(out["radiance"]), nw, 0)
    wf = gather_wavelength_axis(torch.from_numpy(np.transpose(out["d_extinction"], (2, 0, 1)).copy()), nw, 1)
    if rank == 0:
        np.save(os.path.join(tmp, "rad.npy"), rad.numpy())
        np.save(os.path.join(tmp, "wf.npy"), wf.numpy())
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process(tmp_path):
    nw, world = 5, 2
    with socket.socket() as sk_:
        sk_.bind(("127.0.0.1", 0))
        port = sk_.getsockname()[1]
    mp.spawn(_worker, args=(world, port, nw, str(tmp_path)), nprocs=world, join=True)
    from tests.test_oracle_golden import run_oracle
    full = run_oracle(sc.regression_1d(nw), derivatives=True)
    # wavelengths are independent, so the gathered result is bit-identical to the single-process run
    np.testing.assert_array_equal(np.load(tmp_path / "rad.npy"), full["radiance"])
    np.testing.assert_array_equal(np.load(tmp_path / "wf.npy"), np.transpose(full["d_extinction"], (2, 0, 1)))
