"""Worker of tests/test_gpu_distributed.py::test_two_ranks_over_rccl_equal_one_gpu (launched by torch.distributed.run, one rank per GPU)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    rank, local = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from sasktran2_amd.device import DeviceSolver
    from sasktran2_amd.distributed import ShardedSolver
    from tests import scenarios as sc
    from tests.gpu_harness import build_engine
    nw = 37
    s = sc.limb_c2(num_wavelengths=nw)
    engine, _, _ = build_engine(s)
    solver = DeviceSolver(engine)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    sh = ShardedSolver(solver, nw, derivatives=True, device=dev)
    lo, hi = sh.lo, sh.hi
    got = sh.step(to(s["extinction"][:, lo:hi].T), to(s["ssa"][:, lo:hi].T), to(np.transpose(s["leg"][:, :, lo:hi], (2, 1, 0))), to(s["albedo"][lo:hi]))
    full = solver.allocate_outputs(nw, derivatives=True, device=dev)
    solver.solve(to(s["extinction"].T), to(s["ssa"].T), to(np.transpose(s["leg"], (2, 1, 0))), to(s["albedo"]), full)
    torch.cuda.synchronize()
    for k in ("radiance", "los_optical_depth", "wf_extinction", "wf_ssa", "wf_albedo"):
        assert torch.equal(got[k], full[k]), (rank, k)
    # the host entry point of drop-in callers: full atmosphere in, full result out on every rank, one ncclAllGather inside the library
    import sasktran2_amd as sk
    _, config, geometry = build_engine(s)
    idt = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        idt = torch.frombuffer(bytearray(sk.Communicator.unique_id()), dtype=torch.uint8).to(dev)
    dist.broadcast(idt, 0)
    comm = sk.Communicator(bytes(idt.cpu().numpy().tobytes()), rank, dist.get_world_size())
    atmosphere = sk.Atmosphere(geometry, config, numwavel=nw, calculate_derivatives=True)
    atmosphere.storage.total_extinction[:] = s["extinction"]
    atmosphere.storage.ssa[:] = s["ssa"]
    atmosphere.leg_coeff.a1[:] = s["leg"]
    atmosphere.surface.albedo[:] = s["albedo"]
    host = engine.calculate_radiance(atmosphere, as_dataset=False, shard=(comm, rank, dist.get_world_size()))
    assert np.array_equal(host["radiance"][:, :, 0], full["radiance"].cpu().numpy()), rank
    assert np.array_equal(host["wf_extinction"][:, :, :, 0], full["wf_extinction"].cpu().numpy()), rank
    dist.barrier()
    if rank == 0:
        print("sharded == single", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
