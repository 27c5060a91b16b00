"""`bench.py --workload twostream`: the reference's own two-stream benchmark (cpp/lib/tests/twostream/test_twostream.cpp:6-66 and
asv_bench/benchmarks/twostream.py): plane parallel, 20 layers, 220 000 wavelengths, one line of sight, 16 stored Legendre moments,
multiple_scatter_source = twostream with weighting functions (extinction, single-scatter albedo, first phase moment, albedo).  One step =
one pass over the spectrum with the optical inputs resident in HBM; for N > 1 the SAME spectrum is sharded over the ranks and the radiance
block reassembled with one RCCL all-gather ("scaling": "strong").

Roofline bookkeeping: the path is a stream -- per wavelength it must read the level profiles it uses (extinction, ssa, first
moment: 3 x 21 x 8 B, albedo, irradiance) and write the radiance and the weighting functions (3 x 21 + 2 values): 1 040 B.  `achieved`
prices the dominant kernel with those bytes; `traffic` is its measured HBM traffic from the committed rocprofv3 PMC summary."""
from __future__ import annotations

import gc
import json
import os
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0
ROOT = Path(__file__).resolve().parent
PMC_KERNEL_OF = {"k_ts_solve": "ts_solve", "k_ts_view": "ts_view", "k_ts_adjoint": "ts_adjoint", "k_ts_up": "ts_up"}


def pmc_bytes_per_wavelength():
    """Timer name -> measured HBM bytes per wavelength and launch (read + write), from profiles/*/pmc_and_trace_summary_ts.json."""
    import re
    files = sorted((ROOT / "profiles").glob("*/pmc_and_trace_summary_ts.json"))
    out = {}
    if not files:
        return out
    d = json.loads(files[-1].read_text())
    per = float(d.get("wavelengths_per_launch", 0) or 0)
    for r in d.get("pmc", []):
        m = re.match(r"(k_ts_[a-z]+)", r["kernel"])
        name = PMC_KERNEL_OF.get(m.group(1)) if m else None
        if name and per:
            b = (r.get("hbm_read_bytes_per_launch_x2_corrected") or 0) + (r.get("hbm_write_bytes_per_launch") or 0)
            out[name] = out.get(name, 0.0) + b / per
    return out


def scenario(num_wavelengths, nlyr=20, nleg=16):
    from tests import scenarios as sc
    alt = np.arange(nlyr + 1.0)
    nw = num_wavelengths
    leg = np.zeros((nleg, nlyr + 1, nw))
    leg[0] = 1.0
    if nleg > 2:
        leg[2] = 0.5
    # the reference benchmark fills every wavelength with the same numbers; a spectral ramp on the extinction (total optical depth 0.1 ... 4)
    # and a slightly absorbing, slightly anisotropic medium keep every derivative alive
    leg[1] = 0.3
    return dict(geometry=dict(cos_sza=0.6, saa=0.0, earth_radius=6_371_000.0, altitudes=alt, interp_method=sc.INTERP_LOWER,
                              geotype=sc.GEO_PLANEPARALLEL),
                config=dict(nstr=2, num_sza=1), rays=[(sc.GROUND_VIEWING_SOLAR, 0.6, np.pi / 3, 200_000.0, 0.6)],
                extinction=0.01 * np.linspace(0.5, 20.0, nw)[None, :] * np.ones((nlyr + 1, 1)),
                ssa=np.full((nlyr + 1, nw), 0.98) - 0.1 * np.linspace(0, 1, nlyr + 1)[:, None], leg=leg, albedo=np.full(nw, 0.5))


def run_twostream(args, world, rank, dev):
    from sasktran2_amd.device import TsDeviceSolver
    from sasktran2_amd.distributed import gather_wavelength_axis, shard_bounds
    from tests.gpu_harness import build_ts_engine

    NW = args.wavelengths or 220_000
    s = scenario(NW, nleg=int(os.environ.get("TS_BENCH_NLEG", "16")))   # (TS_BENCH_NLEG: probe of the cost of the strided moment read)
    ngrid, nlos = s["extinction"].shape[0], len(s["rays"])
    engine, config, geometry = build_ts_engine(s)
    solver = TsDeviceSolver(engine)
    lo, hi = shard_bounds(NW, rank, world)
    nloc = hi - lo
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    ext, ssa = to(s["extinction"][:, lo:hi].T), to(s["ssa"][:, lo:hi].T)
    leg, alb = to(np.transpose(s["leg"][:, :, lo:hi], (2, 1, 0))), to(s["albedo"][lo:hi])
    out = solver.allocate_outputs(nloc, derivatives=True, device=dev)

    def step():
        solver.solve(ext, ssa, leg, alb, out)
        return gather_wavelength_axis(out["radiance"], NW, 0) if world > 1 else out["radiance"]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    gc.collect()   # (no generation-2 pass of Python's collector inside the timed region: 26-42 ms with torch loaded, bench_frontend.py)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = bool(torch.isfinite(res).all().item()) and bool(torch.isfinite(out["wf_extinction"]).all().item())

    solver.set_profiling(True)
    solver.solve(ext, ssa, leg, alb, out)
    torch.cuda.synchronize()
    ktimes = solver.kernel_times()
    solver.set_profiling(False)
    chunk, _, scratch = solver.launch_shape()
    chunk = min(chunk, nloc)

    algorithmic = 8 * (3 * ngrid + 2) + 8 * (nlos * (3 * ngrid + 2))   # inputs the path reads + outputs it writes, per wavelength
    dominant = max(ktimes, key=lambda k: ktimes[k][0])
    dom_ms, dom_n = ktimes[dominant]
    avg_launch_s = dom_ms / dom_n * 1e-3
    achieved = algorithmic * chunk / avg_launch_s / 1e9
    traffic = pmc_bytes_per_wavelength().get(dominant)
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "note": "algorithmic bytes of the whole path (level profiles in, radiance + weighting functions out) per launch of this kernel",
                "traffic": traffic * chunk if traffic else None, "algorithmic_bytes_per_wavelength": algorithmic,
                "wavelengths_per_launch": chunk, "avg_launch_ms": dom_ms / dom_n, "launches": dom_n}
    if rank != 0:
        return
    value = NW * args.steps / elapsed
    line = {
        "metric": "wavelengths/sec (radiance + weighting functions) two-stream 20 layer / 1 line of sight",
        "value": value, "unit": "wavelengths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "two-stream benchmark of the reference (test_twostream.cpp:6-66): plane parallel, 20 layers, one line of sight, "
                               f"{NW} wavelengths, multiple_scatter_source = twostream with weighting functions",
                   "wavelengths_per_gpu": nloc, "parallelism": f"wavelength-shard x{world} + RCCL all-gather" if world > 1 else "single GPU",
                   "wavelengths_per_launch": chunk},
        "valid": finite, "roofline": roofline,
        "kernels_ms": {k: {"total_ms": round(v[0], 4), "launches": v[1]} for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1][0])},
        "scratch_gib": scratch / 2 ** 30,
    }
    if world == 1 and not args.no_cpu_baseline:
        from tests.test_ts_oracle import ts_oracle
        cores = len(os.sched_getaffinity(0))
        eng = ts_oracle(s)
        n = args.cpu_sample or NW   # the whole spectrum: a few seconds on the GPU host's cores
        idx = np.linspace(0, NW - 1, n).astype(int)
        t = time.perf_counter()
        ref = eng.run(s["extinction"][:, idx], s["ssa"][:, idx], s["leg"][:, :, idx], s["albedo"][idx], derivatives=True, nthreads=cores)
        dt = time.perf_counter() - t
        got = out["radiance"].cpu().numpy()[idx]
        wf = np.transpose(out["wf_extinction"].cpu().numpy()[:, idx], (1, 2, 0))
        cpu_model = ""
        try:
            with open("/proc/cpuinfo") as f:
                cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
        except OSError:
            pass
        line["cpu_baseline"] = {"value": n / dt, "unit": "wavelengths/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
                                "sample": f"{n} of the {NW} wavelengths (evenly spaced), same workload incl. weighting functions "
                                          f"(operation-tape differentiation), {dt:.1f} s",
                                "parity_radiance_max_rel": float(np.max(np.abs(got / ref["radiance"] - 1))),
                                "parity_wf_extinction_max_rel_to_profile_max": float(np.max(
                                    np.abs(wf - ref["d_extinction"]) / np.max(np.abs(ref["d_extinction"]), axis=-1, keepdims=True)))}
        line["speedup_vs_cpu_baseline"] = value / (n / dt)
    print(json.dumps(line), flush=True)
