"""Device-resident front end: inputs and outputs are torch CUDA tensors (float64) that stay in HBM.

PyTorch is plumbing here (device memory, streams, torch.distributed); all arithmetic is in the HIP library via
`sk_amd_engine_solve_device` (include/sasktran2_amd.h)."""
from __future__ import annotations

import ctypes

import torch

from ._lib_loader import last_error, lib


class DeviceSolver:
    def __init__(self, engine):
        """engine: sasktran2_amd.Engine (owns the geometry tables already uploaded to the GPU)."""
        self.engine = engine
        n = ctypes.c_int()
        lib().sk_amd_engine_num_los(engine._h, ctypes.byref(n))
        self.nlos = n.value
        self.ngrid = len(engine._geometry.altitudes())

    def allocate_outputs(self, nwavel, derivatives=True, device="cuda"):
        kw = dict(dtype=torch.float64, device=device)
        out = {"radiance": torch.empty((nwavel, self.nlos), **kw), "los_optical_depth": torch.empty((nwavel, self.nlos), **kw),
               "status": torch.empty(nwavel, dtype=torch.int32, device=device)}
        if derivatives:
            out["wf_extinction"] = torch.empty((self.ngrid, nwavel, self.nlos), **kw)
            out["wf_ssa"] = torch.empty((self.ngrid, nwavel, self.nlos), **kw)
            out["wf_albedo"] = torch.empty((nwavel, self.nlos), **kw)
        return out

    def solve(self, extinction, ssa, leg, albedo, out, irradiance=None, max_chunk=0):
        """extinction, ssa: [nwavel, ngrid]; leg: [nwavel, ngrid, nleg]; albedo: [nwavel] -- contiguous float64 CUDA tensors
        (this is the reference storage layout, location fastest).  `out` from allocate_outputs().  Asynchronous on torch's
        current stream except for one 4-byte read-back per azimuth order (stop rule)."""
        for t in (extinction, ssa, leg, albedo):
            assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        nw = extinction.shape[0]
        derivs = "wf_extinction" in out
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        rc = lib().sk_amd_engine_solve_device(
            self.engine._h, nw, leg.shape[2], p(extinction), p(ssa), p(leg), p(albedo), p(irradiance), int(derivs),
            p(out["radiance"]), p(out.get("wf_extinction")), p(out.get("wf_ssa")), p(out.get("wf_albedo")),
            p(out["los_optical_depth"]), p(out["status"]), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), int(max_chunk))
        if rc != 0:
            raise RuntimeError(f"sk_amd_engine_solve_device returned {rc}: {last_error()}")
        return out

    # ---- wavelength-sharded runs: one slab per rank, one all-gather, one unpack (sasktran2_amd.distributed.ShardedSolver)
    def slab_row_doubles(self, derivatives=True):
        n = ctypes.c_ulonglong()
        lib().sk_amd_slab_row_doubles(self.engine._h, int(derivatives), ctypes.byref(n))
        return n.value

    def solve_slab(self, extinction, ssa, leg, albedo, slab, status, derivatives=True, irradiance=None, max_chunk=0):
        """as solve(), results into the first `nwavel` rows of the wavelength-major `slab` [rows, slab_row_doubles] (float64 CUDA tensor)"""
        for t in (extinction, ssa, leg, albedo, slab):
            assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        nw = extinction.shape[0]
        assert slab.shape[0] >= nw and slab.shape[1] == self.slab_row_doubles(derivatives) and status.shape[0] >= nw
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        rc = lib().sk_amd_engine_solve_device_slab(self.engine._h, nw, leg.shape[2], p(extinction), p(ssa), p(leg), p(albedo), p(irradiance),
                                                   int(derivatives), p(slab), p(status), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
                                                   int(max_chunk))
        if rc != 0:
            raise RuntimeError(f"sk_amd_engine_solve_device_slab returned {rc}: {last_error()}")

    def unpack_slab(self, slab_all, nwavel, world, rows_per_rank, out):
        """gathered slabs [world * rows_per_rank, row] -> `out` (allocate_outputs(nwavel, ...)) in the reference layouts"""
        derivs = "wf_extinction" in out
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        rc = lib().sk_amd_unpack_slab(self.engine._h, p(slab_all), int(nwavel), int(world), int(rows_per_rank), int(derivs), p(out["radiance"]),
                                      p(out.get("los_optical_depth")), p(out.get("wf_extinction")), p(out.get("wf_ssa")), p(out.get("wf_albedo")),
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"sk_amd_unpack_slab returned {rc}: {last_error()}")
        return out

    # ---- per-kernel HIP-event timing (serialises launches; use outside the timed region)
    def set_profiling(self, on):
        """True / 1: per-kernel timing with the launches serialised; 2: event pairs on each launch's own stream, nothing serialised
        (kernel durations inside the pipelined step); False / 0: off"""
        lib().sk_amd_engine_set_profiling(self.engine._h, int(on))
        lib().sk_amd_engine_reset_profile(self.engine._h)

    def reset_profile(self):
        lib().sk_amd_engine_reset_profile(self.engine._h)

    def kernel_times(self):
        buf = ctypes.create_string_buffer(4096)
        lib().sk_amd_engine_kernel_names(self.engine._h, buf, 4096)
        res = {}
        for name in filter(None, buf.value.decode().split(",")):
            ms, n = ctypes.c_double(), ctypes.c_long()
            lib().sk_amd_engine_kernel_time(self.engine._h, name.encode(), ctypes.byref(ms), ctypes.byref(n))
            res[name] = (ms.value, n.value)
        return res

    def launch_shape(self):
        """(wavelengths per launch, blocks in flight, scratch bytes allocated) of the last solve."""
        c, n, b = ctypes.c_int(), ctypes.c_int(), ctypes.c_ulonglong()
        lib().sk_amd_engine_launch_shape(self.engine._h, ctypes.byref(c), ctypes.byref(n), ctypes.byref(b))
        return c.value, n.value, b.value

    def scratch_bytes(self, nwavel):
        b = ctypes.c_ulonglong()
        lib().sk_amd_engine_scratch_bytes(self.engine._h, nwavel, ctypes.byref(b))
        return b.value


class SoDeviceSolver(DeviceSolver):
    """Device-resident front end of an engine with single_scatter_source = exact and / or multiple_scatter_source =
    successive_orders_legacy (`sk_amd_so_engine_solve_device`): radiances [nwavel, nlos, nstokes] and LOS optical depths."""

    def __init__(self, engine):
        super().__init__(engine)
        self.nstokes = int(engine._config.num_stokes)

    def allocate_outputs(self, nwavel, derivatives=False, device="cuda"):
        kw = dict(dtype=torch.float64, device=device)
        out = {"radiance": torch.empty((nwavel, self.nlos, self.nstokes), **kw), "los_optical_depth": torch.empty((nwavel, self.nlos), **kw)}
        if derivatives:   # `sk_amd_so_engine_solve_device_wf`
            out["wf_extinction"] = torch.empty((self.ngrid, nwavel, self.nlos, self.nstokes), **kw)
            out["wf_ssa"] = torch.empty((self.ngrid, nwavel, self.nlos, self.nstokes), **kw)
            out["wf_albedo"] = torch.empty((nwavel, self.nlos, self.nstokes), **kw)
        return out

    def solve(self, extinction, ssa, leg, albedo, out, irradiance=None, max_chunk=0):
        """extinction, ssa: [nwavel, ngrid]; leg: [nwavel, ngrid, nleg (x 4 for NSTOKES = 3)]; albedo: [nwavel] -- contiguous float64
        CUDA tensors in the reference storage layout.  Asynchronous on torch's current stream."""
        for t in (extinction, ssa, leg, albedo):
            assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        per = 4 if self.nstokes == 3 else 1
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if "wf_extinction" in out:
            rc = lib().sk_amd_so_engine_solve_device_wf(
                self.engine._h, extinction.shape[0], leg.shape[2] // per, p(extinction), p(ssa), p(leg), p(albedo), p(irradiance),
                p(out["radiance"]), p(out["los_optical_depth"]), p(out["wf_extinction"]), p(out["wf_ssa"]), p(out["wf_albedo"]), stream)
        else:
            rc = lib().sk_amd_so_engine_solve_device(
                self.engine._h, extinction.shape[0], leg.shape[2] // per, p(extinction), p(ssa), p(leg), p(albedo), p(irradiance),
                p(out["radiance"]), p(out["los_optical_depth"]), stream)
        if rc != 0:
            raise RuntimeError(f"sk_amd_so_engine_solve_device returned {rc}: {last_error()}")
        return out

    def sizes(self):
        v = (ctypes.c_ulonglong * 6)()
        lib().sk_amd_so_engine_sizes(self.engine._h, v)
        return dict(incoming_rays=v[0], incoming_layers=v[1], accumulation_entries=v[2], outgoing_directions=v[3],
                    scratch_bytes_per_wavelength=v[4], wavelengths_per_block=v[5])


class TsDeviceSolver(DeviceSolver):
    """Device-resident front end of an engine whose source is the two-stream one (multiple_scatter_source = twostream and / or
    emission_source = twostream; `sk_amd_ts_engine_solve_device`): radiances [nwavel, nlos], LOS optical depths and, with derivatives,
    wf_extinction / wf_ssa / wf_b1 / wf_emission [ngrid, nwavel, nlos] and wf_albedo / wf_surface_emission [nwavel, nlos]."""

    def __init__(self, engine):
        super().__init__(engine)
        self.thermal = int(engine._config.emission_source) == 4

    def allocate_outputs(self, nwavel, derivatives=True, device="cuda"):
        kw = dict(dtype=torch.float64, device=device)
        out = {"radiance": torch.empty((nwavel, self.nlos), **kw), "los_optical_depth": torch.empty((nwavel, self.nlos), **kw)}
        if derivatives:
            for name in ("wf_extinction", "wf_ssa", "wf_b1") + (("wf_emission",) if self.thermal else ()):
                out[name] = torch.empty((self.ngrid, nwavel, self.nlos), **kw)
            for name in ("wf_albedo",) + (("wf_surface_emission",) if self.thermal else ()):
                out[name] = torch.empty((nwavel, self.nlos), **kw)
        return out

    def solve(self, extinction, ssa, leg, albedo, out, irradiance=None, max_chunk=0, delta_f=None, emission=None, surface_emission=None):
        """extinction, ssa, delta_f, emission: [nwavel, ngrid]; leg: [nwavel, ngrid, nleg]; albedo, irradiance, surface_emission: [nwavel]
        -- contiguous float64 CUDA tensors in the reference storage layout.  Asynchronous on torch's current stream."""
        for t in (extinction, ssa, leg, albedo, irradiance, delta_f, emission, surface_emission):
            assert t is None or (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous())
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        rc = lib().sk_amd_ts_engine_solve_device(
            self.engine._h, extinction.shape[0], leg.shape[2], p(extinction), p(ssa), p(leg), p(delta_f), p(albedo), p(irradiance),
            p(emission), p(surface_emission), p(out["radiance"]), p(out.get("los_optical_depth")), p(out.get("wf_extinction")),
            p(out.get("wf_ssa")), p(out.get("wf_b1")), p(out.get("wf_albedo")), p(out.get("wf_emission")),
            p(out.get("wf_surface_emission")), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), int(max_chunk))
        if rc != 0:
            raise RuntimeError(f"sk_amd_ts_engine_solve_device returned {rc}: {last_error()}")
        return out
