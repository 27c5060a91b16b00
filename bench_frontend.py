"""`bench.py --workload frontend`: the optical-property front end at the size of BASELINE configs[1] -- 51 altitudes,
10 000 wavelengths, 16 stored Legendre moments.  One step = `Atmosphere.internal_object()`: Rayleigh + a temperature-dependent absorber
(5 table temperatures) + a Henyey-Greenstein aerosol given as number density + one given as extinction, assembled in HBM together with
their 9 derivative mappings (wf_o3_vmr, wf_pressure_pa and wf_temperature_k of Rayleigh and of the absorber, the table's own
temperature derivative, wf_aerosol_number_density, wf_cloud_extinction, the Planck term's wf_temperature_k).  The inputs of this path are profiles and spectra on the host
(a few hundred KB): the step includes preparing and uploading them; what it produces stays in HBM for the engines.

Roofline bookkeeping: every kernel is a stream that writes its output once.  Algorithmic bytes per wavelength = arrays
written (storage: extinction, ssa, emission, nleg Legendre rows; per mapping d_extinction, d_ssa and, for scatterers, scat_factor and
nleg d_legendre rows) + the totals the mapping stage reads once (extinction, ssa, leg_coeff), x ngrid x 8 B.  `achieved` prices the
dominant kernel with ITS bytes per launch over its HIP-event time."""
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


def pmc_bytes_per_launch():
    """kernel -> measured HBM bytes per launch (read, corrected as MI355X_MICROARCH.md prescribes, + write) at the profiled size, from
    profiles/*/pmc_and_trace_summary_frontend.json, and the wavelengths of that run"""
    files = sorted((ROOT / "profiles").glob("*/pmc_and_trace_summary_frontend.json"))
    if not files:
        return {}, 0
    d = json.loads(files[-1].read_text())
    return ({r["kernel"]: (r.get("hbm_read_bytes_per_launch_x2_corrected") or 0) + (r.get("hbm_write_bytes_per_launch") or 0) for r in d.get("pmc", [])},
            int(d.get("wavelengths_per_launch", 0) or 0))


def case(nwavel, ngrid=51):
    from tests import scenarios as sc
    c = sc.frontend_case(nwavel=nwavel, ngrid=ngrid)
    c["altitudes"] = np.linspace(0.0, 100000.0, ngrid)
    c["temperature"] = np.where(c["altitudes"] < 11000.0, 288.15 - 6.5e-3 * c["altitudes"], 216.65 + 1.0e-3 * np.maximum(c["altitudes"] - 20000.0, 0.0))
    c["pressure"] = 101325.0 * np.exp(-c["altitudes"] / 7400.0)
    return c


def kernel_bytes(ngrid, nleg, nw, nplain, nscat):
    """algorithmic bytes per launch of every front-end kernel at this size: arrays written once, totals read once"""
    cell = ngrid * 8 * nw
    return {"k_fe_assemble": 2 * cell, "k_fe_legendre": nleg * cell, "k_fe_planck": cell,
            "k_fe_map_all": (2 + 2 * nplain + 3 * nscat) * cell,          # extinction, ssa in; d_extinction, d_ssa (, scat_factor) out per mapping
            "k_fe_map_legendre_all": (1 + nscat) * nleg * cell}           # leg_coeff in once; d_legendre out per scatterer mapping


def run_frontend(args, world, rank, dev):
    import sasktran2_amd as sk
    from sasktran2_amd.distributed import shard_bounds
    from tests import scenarios as sc

    NW = args.wavelengths or 10_000
    lo, hi = shard_bounds(NW, rank, world)
    c = case(NW)
    c["wavelengths"] = c["wavelengths"][lo:hi]
    nloc = hi - lo
    nmoments = 16
    cfg = sk.Config()
    cfg.num_streams = 16
    cfg.num_singlescatter_moments = nmoments
    geo = sk.Geometry1D(0.6, 0.0, 6372000.0, c["altitudes"], sk.InterpolationMethod.LinearInterpolation, sk.GeometryType.Spherical)
    atmo = sc.frontend_product_atmosphere(sk, c, geo, cfg, thermal=True)
    ngrid = len(c["altitudes"])

    def step():
        return atmo.internal_object()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # a step is under a millisecond; a generation-2 pass of Python's cyclic collector over torch's module objects is 26-42 ms and would
    # otherwise fall on one of the first timed steps (tools/probes/frontend_steps.py): collect now, as timeit does before a measurement
    gc.collect()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fe = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    mappings = dict(fe.mappings)
    nscat = sum(m.scatterer for m in mappings.values())
    nplain = sum((not m.scatterer) and (not m.emission) for m in mappings.values())
    nem = sum(m.emission for m in mappings.values())
    cell = ngrid * 8
    written = cell * (3 + nmoments) + nplain * 2 * cell + nscat * (3 + nmoments) * cell + nem * 3 * cell
    read = 2 * cell + (nmoments * cell if nscat else 0)   # the totals, once (the mappings are formed by two launches)
    algorithmic = written + read

    fe.set_profiling(True)
    step()
    torch.cuda.synchronize()
    ktimes = fe.kernel_times()
    fe.set_profiling(False)
    dominant = max(ktimes, key=lambda k: ktimes[k][0])
    dom_ms, dom_n = ktimes[dominant]
    per_launch = kernel_bytes(ngrid, nmoments, nloc, nplain, nscat)[dominant]
    achieved = per_launch / (dom_ms / dom_n * 1e-3) / 1e9
    kernels_total_ms = sum(v[0] for v in ktimes.values())
    pmc, pmc_nw = pmc_bytes_per_launch()
    traffic = pmc.get(dominant) * nloc / pmc_nw if pmc.get(dominant) and pmc_nw else None
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "algorithmic_bytes_per_launch": per_launch, "avg_launch_ms": dom_ms / dom_n, "launches": dom_n,
                "all_kernels": {"algorithmic_bytes_per_wavelength": algorithmic, "total_ms": kernels_total_ms,
                                "achieved_GBs": algorithmic * nloc / (kernels_total_ms * 1e-3) / 1e9},
                "note": "bytes the kernel must write (and the totals a mapping must read) per launch over its HIP-event time"}
    if rank != 0:
        return
    value = NW * args.steps / elapsed
    line = {
        "metric": "wavelengths/sec, atmosphere assembled from constituents (storage + derivative mappings in HBM)",
        "value": value, "unit": "wavelengths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"optical-property front end at the C2 size: {ngrid} altitudes, {NW} wavelengths, {nmoments} moments; Rayleigh + "
                               "temperature-dependent absorber (5 table temperatures) + 2 Henyey-Greenstein scatterers + thermal emission, "
                               f"{len(mappings)} derivative mappings ({nscat} with d_legendre)",
                   "wavelengths_per_gpu": nloc, "parallelism": f"wavelength-shard x{world}" if world > 1 else "single GPU"},
        "roofline": roofline,
        "kernels_ms": {k: {"total_ms": round(v[0], 4), "launches": v[1]} for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1][0])},
        "host_and_launch_ms": elapsed / args.steps * 1e3 - kernels_total_ms,
    }
    if world == 1 and not args.no_cpu_baseline:
        n = min(args.cpu_sample or 2000, NW)
        idx = np.linspace(0, NW - 1, n).astype(int)
        cs = dict(c, wavelengths=c["wavelengths"][idx])
        t = time.perf_counter()
        want = sc.frontend_oracle_atmosphere(cs, nmoments, thermal=True).assemble()
        dt = time.perf_counter() - t
        st = atmo.storage
        got_ext, got_leg = st.total_extinction[:, idx], st.leg_coeff[:, :, idx]
        cpu_model = ""
        try:
            with open("/proc/cpuinfo") as f:
                cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
        except OSError:
            pass
        m = st.get_derivative_mapping("wf_aerosol_number_density")
        line["cpu_baseline"] = {"value": n / dt, "unit": "wavelengths/s", "cores": 1, "kind": "port", "cpu_model": cpu_model,
                                "sample": f"{n} of the {NW} wavelengths (evenly spaced), the same constituents and mappings through the numpy oracle "
                                          f"(oracle/frontend_oracle.py), {dt:.1f} s",
                                "parity_extinction_max_rel": float(np.max(np.abs(got_ext / want.ext - 1))),
                                "parity_leg_coeff_max_abs": float(np.max(np.abs(got_leg - want.leg))),
                                "parity_d_legendre_max_abs": float(np.max(np.abs(m.d_leg_coeff[:, :, idx] - want.mappings["wf_aerosol_number_density"].d_legendre)))}
        line["speedup_vs_cpu_baseline"] = value / (n / dt)
        # not like for like: the timed device step is the steady state (spectra and altitude matrices of the constituents stay in HBM between
        # assemblies), the numpy leg builds its constituents and evaluates their cross sections from scratch
        line["speedup_note"] = "device: steady-state re-assembly with cached spectra; CPU: cold construction + assembly -- an upper bound, not a like-for-like ratio"
    print(json.dumps(line), flush=True)
