"""world_size-2 gloo test of the wavelength sharding: one packed slab per rank, ONE all-gather, unpack (CPU; the oracle stands in for the
solver and the numpy mirror of the slab layout for the device kernels -- tests/test_gpu_distributed.py checks those against it)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sasktran2_amd.distributed import all_gather_slab, pack_slab_host, rows_per_rank, shard_bounds, unpack_slab_host
from tests import scenarios as sc


def test_shard_bounds_cover_everything():
    for nw in (1, 7, 10_000, 10_001):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(nw, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == nw
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1
            assert rows_per_rank(nw, world) == max(e - s for s, e in b)


def _worker(rank, world, port, nw, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.test_oracle_golden import make_oracle
    s = sc.regression_1d(nw)
    lo, hi = shard_bounds(nw, rank, world)
    out = make_oracle(s).run(s["extinction"][:, lo:hi], s["ssa"][:, lo:hi], s["leg"][:, :, lo:hi], s["albedo"][lo:hi],
                             derivatives=True)
    local = pack_slab_host(out["radiance"], out["los_optical_depth"], np.transpose(out["d_extinction"], (2, 0, 1)),
                           np.transpose(out["d_ssa"], (2, 0, 1)), out["d_albedo"], rows=rows_per_rank(nw, world))
    calls = []
    orig = dist.all_gather
    dist.all_gather = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    gathered = all_gather_slab(torch.from_numpy(local))
    dist.all_gather = orig
    assert len(calls) == 1   # one collective for radiance and every weighting function
    full = unpack_slab_host(gathered.numpy(), nw, world, out["radiance"].shape[1])
    if rank == 0:
        np.savez(os.path.join(tmp, "full.npz"), **full)
    dist.destroy_process_group()


def test_two_rank_slab_gather_equals_single_process(tmp_path):
    nw, world = 5, 2   # shards of 3 and 2 wavelengths: the second slab is padded
    with socket.socket() as sk_:
        sk_.bind(("127.0.0.1", 0))
        port = sk_.getsockname()[1]
    mp.spawn(_worker, args=(world, port, nw, str(tmp_path)), nprocs=world, join=True)
    from tests.test_oracle_golden import run_oracle
    full = run_oracle(sc.regression_1d(nw), derivatives=True)
    got = np.load(tmp_path / "full.npz")
    # wavelengths are independent, so the gathered result is bit-identical to the single-process run
    np.testing.assert_array_equal(got["radiance"], full["radiance"])
    np.testing.assert_array_equal(got["los_optical_depth"], full["los_optical_depth"])
    np.testing.assert_array_equal(got["wf_extinction"], np.transpose(full["d_extinction"], (2, 0, 1)))
    np.testing.assert_array_equal(got["wf_ssa"], np.transpose(full["d_ssa"], (2, 0, 1)))
    np.testing.assert_array_equal(got["wf_albedo"], full["d_albedo"])
