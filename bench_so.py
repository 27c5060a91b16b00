"""`bench.py --workload c5`: BASELINE.json configs[4] -- exact single scatter + 10 successive orders of scattering (the reference's
SuccessiveOrdersLegacy source, 110 incoming / 110 outgoing directions per diffuse point), 50 layers, 25 limb lines of sight,
20 000 wavelengths.  One step = one pass over the spectrum with the optical inputs resident in HBM; for N > 1 the SAME spectrum is
sharded over the ranks and the radiance block reassembled with one RCCL all-gather ("scaling": "strong").

Roofline bookkeeping: every kernel of this path is a stream over [item][wavelength] arrays, its algorithmic bytes per
wavelength and launch are counted from the engine's own sizes (below); the scattering kernel is priced against the FP64 peak with its
exact operation count 2 x rows x cols x moments per diffuse point."""
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
FP64_PEAK_TFLOPS = 78.6


ROOT = Path(__file__).resolve().parent
PMC_KERNEL_OF = {"k_so_accumulate": "so_accumulate", "k_so_layers": "so_layers", "k_so_scan": "so_scan", "k_so_sh": "so_scatter",
                 "k_so_scatter": "so_scatter", "k_so_march": "so_march_los", "k_so_tables": "so_tables", "k_so_transpose": "so_transpose"}


def pmc_bytes_per_wavelength():
    """Timer name -> measured HBM bytes per wavelength and launch (read + write), from profiles/*/pmc_and_trace_summary_so.json."""
    import re
    files = sorted((ROOT / "profiles").glob("*/pmc_and_trace_summary_so.json"))
    out = {}
    if not files:
        return out
    for r in json.loads(files[-1].read_text()).get("pmc_4096_wavelengths_per_launch", []):
        m = re.match(r"(k_so_[a-z]+)", r["kernel"])
        name = PMC_KERNEL_OF.get(m.group(1)) if m else None
        if name:
            b = (r.get("hbm_read_bytes_per_launch_x2_corrected") or 0) + (r.get("hbm_write_bytes_per_launch") or 0)
            out[name] = out.get(name, 0.0) + b / 4096
    return out


def run_c5(args, world, rank, dev):
    from sasktran2_amd.device import SoDeviceSolver
    from sasktran2_amd.distributed import gather_wavelength_axis, shard_bounds
    from tests import scenarios as sc
    from tests.gpu_harness import build_so_engine

    NW = args.wavelengths or 20_000
    s = sc.limb_c5(num_wavelengths=NW)
    c = s["config"]
    engine, config, geometry = build_so_engine(s)
    solver = SoDeviceSolver(engine)
    lo, hi = shard_bounds(NW, rank, world)
    nloc = hi - lo
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    ext, ssa = to(s["extinction"][:, lo:hi].T), to(s["ssa"][:, lo:hi].T)
    leg, alb = to(np.transpose(s["leg"][:, :, lo:hi], (2, 1, 0))), to(s["albedo"][lo:hi])
    wf = bool(getattr(args, "weighting_functions", False))
    out = solver.allocate_outputs(nloc, derivatives=wf, device=dev)

    def step():
        solver.solve(ext, ssa, leg, alb, out)
        if world == 1:
            return out["radiance"]
        if not wf:
            return gather_wavelength_axis(out["radiance"], NW, 0)
        # one wavelength-major slab per rank -- [radiance | wf_extinction | wf_ssa | wf_albedo] per wavelength -- and ONE all-gather
        slab = torch.cat([out["radiance"].reshape(nloc, -1), out["wf_extinction"].movedim(0, 1).reshape(nloc, -1),
                          out["wf_ssa"].movedim(0, 1).reshape(nloc, -1), out["wf_albedo"].reshape(nloc, -1)], dim=1)
        return gather_wavelength_axis(slab, NW, 0)

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
    finite = bool(torch.isfinite(res).all().item())
    sizes = solver.sizes()

    solver.set_profiling(True)
    solver.solve(ext, ssa, leg, alb, out)
    torch.cuda.synchronize()
    ktimes = solver.kernel_times()
    solver.set_profiling(False)
    sizes = solver.sizes()
    chunk = min(int(sizes["wavelengths_per_block"]), nloc)

    # algorithmic bytes / FP64 operations per wavelength and launch, from the discretisation the engine reports
    nr, nl, ne, no = (sizes[k] for k in ("incoming_rays", "incoming_layers", "accumulation_entries", "outgoing_directions"))
    ndir, nmom, ngrid, nlos = c["num_outgoing"], c["num_streams"], len(s["geometry"]["altitudes"]), len(s["rays"])
    npts = (ngrid - 1) * c["num_sza"]   # interior diffuse points (one per layer mid point and solar zenith profile)
    model = {
        # reads: layer factors, first order, outgoing (once: it is re-read from L2); writes: incoming
        "so_accumulate": {"bytes": 8 * (nl + 3 * nr + no), "flops": 2 * ne},
        # reads incoming + the points' Legendre coefficients, writes outgoing; 2 x moments x rows x cols per interior point
        # interior points through spherical-harmonic space (k_so_sh): 2 x L^2 x (nin + nout) multiply-adds per point instead of the direct
        # L x nin x nout; reads incoming + the points' Legendre coefficients, writes / reads the L^2 harmonic coefficients, writes outgoing
        "so_scatter": {"bytes": 8 * (nr + no + npts * nmom + 2 * npts * nmom * nmom),
                       "flops": 2 * nmom * nmom * (ndir + c["num_incoming"]) * npts, "direct_flops": 2 * nmom * ndir * c["num_incoming"] * npts},
        # per layer: writes optical depth, unattenuated source and omega (1 - e^-od); the table / optics it reads are L2 resident
        "so_layers": {"bytes": 8 * 3 * nl, "flops": None},
        # reads those three, writes the layer factors in place + first order / incoming
        "so_scan": {"bytes": 8 * (4 * nl + 3 * nr), "flops": None},
        "so_march_los": {"bytes": 8 * (nlos * 2), "flops": None},
        "so_tables": {"bytes": 8 * (100 * ngrid + npts * nmom), "flops": None},
        "so_transpose": {"bytes": 16 * ngrid * (2 + nmom), "flops": None},
    }
    dominant = max(ktimes, key=lambda k: ktimes[k][0])
    dom_ms, dom_n = ktimes[dominant]
    nblocks = -(-nloc // chunk)
    avg_launch_s = dom_ms / dom_n * 1e-3
    m = model.get(dominant, {"bytes": 0, "flops": None})
    if dominant == "so_scatter":
        achieved = m["flops"] * chunk / avg_launch_s / 1e12
        roofline = {"bound": "mfma", "kernel": dominant, "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP64_PEAK_TFLOPS, "flops_per_launch": m["flops"] * chunk,
                    "note": "exact FP64 operation count 2 x L^2 x (incoming + outgoing directions) per interior diffuse point and wavelength "
                            "(scattering through spherical-harmonic space); vector and matrix FP64 peaks are both 78.6 TFLOP/s on MI355X"}
    else:
        achieved = m["bytes"] * chunk / avg_launch_s / 1e9
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "note": "algorithmic bytes of this kernel per wavelength x wavelengths per launch / average launch duration"}
    # HBM bytes per launch from the newest committed rocprofv3 PMC summary of this workload (tools/profile_so.sh: separate --pmc passes at
    # 4096 wavelengths per launch, FETCH_SIZE x2-corrected as the MI355X guide prescribes), scaled to this launch's wavelengths
    traffic = pmc_bytes_per_wavelength().get(dominant)
    roofline.update({"traffic": traffic * chunk if traffic else None, "algorithmic_bytes_per_wavelength": m["bytes"], "wavelengths_per_launch": chunk,
                     "avg_launch_ms": dom_ms / dom_n, "launches": dom_n})
    if rank != 0:
        return
    value = NW * args.steps / elapsed
    per_solve = {}
    for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1][0]):
        e = {"total_ms": round(v[0], 3), "launches": v[1]}
        if k in model and v[0] > 0:
            launches_per_block = v[1] / nblocks
            e["algorithmic_GBs"] = model[k]["bytes"] * nloc * launches_per_block / (v[0] * 1e-3) / 1e9
            if model[k]["flops"]:
                e["fp64_TFLOPs"] = model[k]["flops"] * nloc * launches_per_block / (v[0] * 1e-3) / 1e12
        per_solve[k] = e
    line = {
        "metric": ("wavelengths/sec (radiance + wf_extinction / wf_ssa / wf_albedo, exact single scatter + 10 successive orders) 50 layer / 110 directions" if wf
                   else "wavelengths/sec (radiance, exact single scatter + 10 successive orders) 50 layer / 110 directions"),
        "value": value, "unit": "wavelengths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "C5 limb-scatter spherical: exact single scatter + SuccessiveOrdersLegacy, 10 orders, 110 incoming / 110 outgoing "
                               f"directions, 50 layers, 25 tangent heights 10-58 km, {NW} wavelengths, Stokes I"
                               + (", weighting functions of all lines of sight (discrete adjoint of the 10 passes)" if wf else ""),
                   "wavelengths_per_gpu": nloc, "parallelism": f"wavelength-shard x{world} + RCCL all-gather" if world > 1 else "single GPU",
                   "wavelengths_per_launch": chunk, **{k: int(v) for k, v in sizes.items()}},
        "valid": finite, "roofline": roofline, "kernels_ms": per_solve,
        "scratch_gib": sizes["scratch_bytes_per_wavelength"] * chunk / 2 ** 30,
    }
    if world == 1 and not args.no_cpu_baseline:
        from tests.test_so_oracle import make_so_oracle
        from bench import host_cpus
        cores = host_cpus()[0]   # the share the container grants (cgroup cpu.max), not the affinity mask
        eng = make_so_oracle(s)
        probe = max(cores, 8)
        idx = np.linspace(0, NW - 1, probe).astype(int)
        t = time.perf_counter()
        eng.run(s["extinction"][:, idx], s["ssa"][:, idx], s["leg"][:, :, idx], s["albedo"][idx], nthreads=cores, derivatives=wf)
        rate = probe / (time.perf_counter() - t)
        n = args.cpu_sample or int(min(max(rate * 15.0, probe), 8192))
        idx = np.linspace(0, NW - 1, n).astype(int)
        t = time.perf_counter()
        ref = eng.run(s["extinction"][:, idx], s["ssa"][:, idx], s["leg"][:, :, idx], s["albedo"][idx], nthreads=cores, derivatives=wf)
        dt = time.perf_counter() - t
        got = out["radiance"].cpu().numpy()[idx]
        cpu_model = ""
        try:
            with open("/proc/cpuinfo") as f:
                cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
        except OSError:
            pass
        line["cpu_baseline"] = {"value": n / dt, "unit": "wavelengths/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
                                "sample": f"{n} of the {NW} wavelengths (evenly spaced), same workload, {dt:.1f} s",
                                "parity_radiance_max_rel": float(np.max(np.abs(got / ref["radiance"] - 1)))}
        if wf:   # [ngrid, nwavel, nlos, S] -> the oracle's [nwavel, nlos, S, ngrid], relative to the largest value of each profile
            for key, name in (("wf_extinction", "d_extinction"), ("wf_ssa", "d_ssa")):
                g_ = out[key].cpu().numpy()[:, idx].transpose(1, 2, 3, 0)
                line["cpu_baseline"][f"parity_{key}_max_rel_to_profile_max"] = float(np.max(np.abs(g_ - ref[name]) / np.abs(ref[name]).max(axis=3, keepdims=True)))
        line["speedup_vs_cpu_baseline"] = value / (n / dt)
    print(json.dumps(line), flush=True)
