"""`bench.py --workload c4`: BASELINE.json configs[3] -- polarised (NSTOKES = 3) discrete ordinates, 32 streams (96 x 96 layer matrices),
60 layers, 25 limb lines of sight, 5 000 wavelengths, Stokes I / Q / U + wf_extinction / wf_ssa / wf_albedo.

One step = one Engine.calculate_radiance() of the whole spectrum.  This path is driven through the reference-compatible HOST entry point
(sk_engine_calculate_radiance: the run-time-n / polarised kernels have no device-resident entry point), so the step includes the upload of
the optical properties (0.19 GB) and the download of the weighting functions (0.37 GB): about 0.05 s of PCIe time in a step of several
seconds, stated in `config.transfers`.

Roofline: FP64 operations executed by the dominant kernel (rocprofv3 PMC pass of tools/profile_c4.sh: vector FP64 instructions x 64 lanes +
matrix-core MOPS x 512, per wavelength and launch at the profiled block size) over its HIP-event duration, against the 78.6 TFLOP/s FP64
peak; `traffic` = its measured HBM bytes per launch."""
from __future__ import annotations

import gc
import json
import os
import re
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
HBM_PEAK_GBS = 8000.0
FP64_PEAK_TFLOPS = 78.6
ALGORITHMIC_BYTES_PER_WAVELENGTH = 137_856   # inputs 63 456 B + outputs 74 400 B
PMC_KERNEL_OF = {"kern_v1": "kv1_layer_solution", "kern_v1p": "kv1_layer_solution", "kern_v2": "kv2_bvp", "kern_v2_w48": "kv2_bvp", "kern_v2_g96": "kv2_bvp",
                 "kern_v2c": "kv2c_adjoint_ops",
                 "kern_v2b": "kv2b_adjoint", "kern_v2bt": "kv2b_adjoint", "kern_v3": "kv3_table_values", "kern_v4": "kv4_los",
                 "kern_vk4a": "k4a_chain", "kern_v0": "k0_layer_optics"}


def load_pmc():
    """timer name -> (FP64 operations, HBM bytes) per wavelength and launch from the newest profiles/*/pmc_and_trace_summary_c4.json"""
    files = sorted((ROOT / "profiles").glob("*/pmc_and_trace_summary_c4.json"))
    if not files:
        return {}, None
    doc = json.loads(files[-1].read_text())
    per = float(doc.get("wavelengths_per_launch", 4096))
    out = {}
    for r in doc.get("pmc", doc.get("pmc_4096_wavelengths_per_launch", [])):
        m = re.match(r"(kern_v\w+?)(<|\(|$)", r["kernel"])
        name = PMC_KERNEL_OF.get(m.group(1)) if m else None
        if not name:
            continue
        f = (r.get("fp64_valu_flops_per_launch_upper_bound") or 0) + (r.get("fp64_mfma_flops_per_launch") or 0)
        b = (r.get("hbm_read_bytes_per_launch_x2_corrected") or 0) + (r.get("hbm_write_bytes_per_launch") or 0)
        e = out.setdefault(name, [0.0, 0.0])
        e[0] += f / per
        e[1] += b / per
    return out, files[-1].parent.name


def run_c4(args, world, rank, dev):
    if world != 1:
        raise SystemExit("--workload c4 is a single-GPU workload (the host entry point of one process drives one GPU)")
    from bench import host_cpus
    from sasktran2_amd.device import DeviceSolver
    from tests import scenarios as sc
    from tests.gpu_harness import build_engine, run_product

    NW = args.wavelengths or 5_000
    s = sc.to_vector(sc.limb_c2(num_wavelengths=NW, nlyr=60, nstr=32), rayleigh_fraction=0.6)
    eng = build_engine(s)
    solver = DeviceSolver(eng[0])
    for _ in range(max(args.warmup, 1)):
        out = run_product(s, derivatives=True, engine=eng)
    gc.collect()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run_product(s, derivatives=True, engine=eng)
    elapsed = time.perf_counter() - t0
    value = NW * args.steps / elapsed
    finite = bool(np.all(np.isfinite(out["radiance"])) and np.all(out["radiance"][..., 0] > 0))
    # per-kernel durations (HIP events around every launch; this driver runs its launches in order on one stream)
    solver.set_profiling(True)
    run_product(s, derivatives=True, engine=eng)
    ktimes = solver.kernel_times()
    solver.set_profiling(False)
    chunk, nblocks_slots, scratch = solver.launch_shape()
    pmc, pmc_source = load_pmc()
    dominant = max(ktimes, key=lambda k: ktimes[k][0])
    dom_ms, dom_n = ktimes[dominant]
    avg_launch_s = dom_ms / dom_n * 1e-3
    flops_w, bytes_w = pmc.get(dominant, (None, None))
    if flops_w:
        achieved = flops_w * chunk / avg_launch_s / 1e12
        roofline = {"bound": "mfma", "kernel": dominant, "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP64_PEAK_TFLOPS, "flops_per_launch": flops_w * chunk,
                    "note": "FP64 operations executed (PMC: vector FP64 instructions x 64 lanes + matrix-core MOPS x 512) per launch / HIP-event "
                            f"duration of the launch; counters from profiles/{pmc_source}"}
    else:
        achieved = ALGORITHMIC_BYTES_PER_WAVELENGTH * chunk / avg_launch_s / 1e9
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "note": "no PMC summary of this workload is committed: algorithmic bytes of the whole path per launch of the dominant kernel"}
    roofline.update({"traffic": bytes_w * chunk if bytes_w else None, "algorithmic_bytes_per_wavelength": ALGORITHMIC_BYTES_PER_WAVELENGTH,
                     "wavelengths_per_launch": chunk, "avg_launch_ms": dom_ms / dom_n, "launches": dom_n})
    ngrid, nlos = s["extinction"].shape[0], len(s["rays"])
    line = {
        "metric": "wavelengths/sec (Stokes I/Q/U + full WF) 60 layer / 32 stream polarised",
        "value": value, "unit": "wavelengths/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 1),
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"C4 limb-scatter spherical, NSTOKES = 3: 32-stream DISCO (96 x 96 layer matrices), 60 layers, {nlos} tangent heights, "
                               f"{NW} wavelengths, I/Q/U + wf_extinction/wf_ssa/wf_albedo",
                   "wavelengths_per_launch": chunk, "parallelism": "single GPU",
                   "transfers": "host entry point: optical properties uploaded and radiances / weighting functions downloaded inside the step "
                                f"({(s['extinction'].nbytes + s['ssa'].nbytes + s['leg'].nbytes) / 1e9:.2f} GB in, "
                                f"{(2 * ngrid + 2) * NW * nlos * 3 * 8 / 1e9:.2f} GB out)"},
        "valid": finite, "roofline": roofline,
        "kernels_ms": {k: {"total_ms": round(v[0], 3), "launches": v[1]} for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1][0])},
        "scratch_gib": scratch / 2 ** 30,
    }
    if not args.no_cpu_baseline:
        from tests.test_oracle_golden import make_oracle
        cores, logical_cpus, physical_cores = host_cpus()
        orc = make_oracle(s)
        n = args.cpu_sample or 2 * cores   # ~11 s per wavelength and core with the dense forward-mode linearisation
        idx = np.linspace(0, NW - 1, n).astype(int)
        t = time.perf_counter()
        ref = orc.run(s["extinction"][:, idx], s["ssa"][:, idx], s["leg"][:, :, idx], s["albedo"][idx], derivatives=True, nthreads=cores)
        dt = time.perf_counter() - t
        cpu_model = ""
        try:
            with open("/proc/cpuinfo") as f:
                cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
        except OSError:
            pass
        got = out["radiance"][idx]
        scale = np.abs(ref["d_extinction"]).max(axis=3, keepdims=True)
        line["cpu_baseline"] = {"value": n / dt, "unit": "wavelengths/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
                                "host_logical_cpus": logical_cpus, "host_physical_cores": physical_cores,
                                "sample": f"{n} of the {NW} wavelengths (evenly spaced), same workload incl. weighting functions, {dt:.1f} s",
                                "parity_I_max_rel": float(np.max(np.abs(got[..., 0] / ref["radiance"][..., 0] - 1))),
                                "parity_QU_max_abs_over_I_max": float(np.max(np.abs(got - ref["radiance"])) / np.abs(ref["radiance"]).max()),
                                "parity_wf_extinction_max_rel_to_profile_max":
                                    float(np.max(np.abs(out["d_extinction"][idx] - ref["d_extinction"]) / np.where(scale > 0, scale, 1.0)))}
        line["speedup_vs_cpu_baseline"] = value / (n / dt)
    print(json.dumps(line), flush=True)
