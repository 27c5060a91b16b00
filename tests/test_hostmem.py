"""The mirror's page-locked pool (sasktran2_amd/hostmem.py) without a GPU: it hands out ordinary arrays of the asked shape / order, blocks are
reused, and small arrays stay ordinary NumPy memory.  (With a GPU the same interface returns page-locked blocks; the reason every DMA target of
a call comes from it is measured in tools/probes/host_path_patterns.py.)"""
import numpy as np

from sasktran2_amd import hostmem


def test_pool_arrays_have_the_asked_shape_order_and_dtype():
    for shape, order in (((3, 5), "C"), ((51, 4000, 3), "F"), ((2_000_000,), "C")):
        a = hostmem.pool.empty(shape, order=order)
        assert a.shape == shape and a.dtype == np.float64
        assert a.flags.f_contiguous if order == "F" else a.flags.c_contiguous
        a[...] = 1.5
        assert float(a.sum()) == 1.5 * a.size
    z = hostmem.pool.zeros((600, 700))
    assert not z.any()


def test_small_arrays_are_ordinary_memory_and_the_threshold_is_the_mmap_size():
    assert hostmem._SMALL == 128 << 10   # glibc serves smaller requests from the heap: dropping them unmaps nothing
    a = hostmem.pool.empty((100,))
    assert isinstance(a, np.ndarray) and a.base is None
