"""The C-ABI library loads without a GPU and exports every symbol include/sasktran2_amd.h declares; compute entry
points fail loudly (no CPU fallback) when no HIP device is present."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from sasktran2_amd import _lib_loader
    return _lib_loader


def test_header_symbols_are_exported(built):
    header = (ROOT / "include" / "sasktran2_amd.h").read_text()
    declared = set(re.findall(r"\b(sk_[a-z0-9_]+)\s*\(", header))
    assert len(declared) > 60
    L = built.lib()
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert declared == {n for n, _, _ in built.SIGNATURES}


def test_handles_without_gpu(built):
    import torch
    import sasktran2_amd as sk
    c = sk.Config()
    c.num_streams = 8
    assert c.num_streams == 8 and c.num_stokes == 1
    if not torch.cuda.is_available():
        from tests import scenarios as sc
        from tests.gpu_harness import build_engine
        with pytest.raises(RuntimeError, match="no HIP device"):
            build_engine(sc.regression_1d())
