"""Page-locked host arrays for the mirror's large buffers (the atmosphere storage a caller fills, the radiance / weighting-function
arrays a call returns).  The reference hands the engine plain ndarrays (src/sasktran2/atmosphere.py:719-858, engine.py:488-554); so does
this mirror, only that arrays of a megabyte and more come from `sk_amd_host_alloc` (hipHostMalloc): the DMA engines move them without a
staging copy, which is the difference between ~8 and ~50 GB/s for the 74 MB in / 208 MB out of one 10 000-wavelength call.

Blocks are pooled by size (pinning costs tens of milliseconds per 100 MB): an array's block returns to the pool when the array and all of
its views are gone.  EVERY array a call's DMA writes into comes from here, the 2 MB radiance / wf_albedo arrays too: a pageable array that was
the target of a device-to-host copy and is then dropped by the caller is munmap'ed, the driver invalidates its GPU mapping, and the next
device synchronisation stalls 20-29 ms (measured round 4, `tools/probes/host_path_patterns.py`: 138.5 -> 110.3 ms per 10 000-wavelength call
for a caller that drops the previous result before the next call).  Without a GPU, or when pinning fails, ordinary NumPy memory is used -- this is memory placement, not a compute path."""
from __future__ import annotations

import ctypes
import os
import threading
import weakref

import numpy as np

_SMALL = 128 << 10        # below this an ordinary array (glibc serves it from the heap: no munmap when it is dropped, see below)
_GRANULE = 2 << 20        # pooled block sizes are multiples of this
_KEEP = 2 << 30           # free bytes the pool keeps before it returns blocks to the driver


class _PinnedPool:
    def __init__(self):
        self._free = {}
        self._free_bytes = 0
        self._lock = threading.Lock()
        self._failed = os.environ.get("SASKTRAN2_AMD_HOST_ALLOC", "pinned") == "pageable"   # (ordinary NumPy memory throughout)

    def _release(self, nbytes, ptr):
        from ._lib_loader import lib
        with self._lock:
            if self._free_bytes + nbytes <= _KEEP:
                self._free.setdefault(nbytes, []).append(ptr)
                self._free_bytes += nbytes
                return
        if callable(lib):   # (interpreter shutdown)
            lib().sk_amd_host_free(ptr)

    def empty(self, shape, order="C"):
        """Uninitialised float64 array of `shape`; pinned when it is large and pinning works."""
        shape = tuple(int(v) for v in np.atleast_1d(shape))
        n = int(np.prod(shape)) if shape else 1
        if n * 8 < _SMALL or self._failed:
            return np.empty(shape, order=order)
        nbytes = -(-n * 8 // _GRANULE) * _GRANULE
        ptr = None
        with self._lock:
            blocks = self._free.get(nbytes)
            if blocks:
                ptr = blocks.pop()
                self._free_bytes -= nbytes
        if ptr is None:
            from ._lib_loader import lib
            ptr = lib().sk_amd_host_alloc(nbytes)
            if not ptr:
                self._failed = True   # no device / no pinnable memory: do not ask again
                return np.empty(shape, order=order)
        buf = (ctypes.c_double * n).from_address(ptr)
        weakref.finalize(buf, self._release, nbytes, ptr)
        return np.frombuffer(buf, dtype=np.float64).reshape(shape, order=order)

    def zeros(self, shape, order="C"):
        a = self.empty(shape, order=order)
        a[...] = 0.0
        return a


pool = _PinnedPool()
