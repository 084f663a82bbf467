"""CPU-side checks of the C-ABI boundary: the library loads without a GPU and exports every
symbol include/pmstep.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pysco_amd import build

    build.build()
    from pysco_amd import _lib

    return _lib.load_library()


def _declared():
    src = open(os.path.join(ROOT, "include", "pmstep.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/pmstep.h but not exported"


def test_binding_covers_header(lib):
    from pysco_amd import _lib

    assert set(_declared()) == set(_lib.SIGNATURES), set(_declared()) ^ set(_lib.SIGNATURES)
    assert lib.pm_version() >= 100


def test_halo_buffer_sizes(lib):
    """Pure size functions (no GPU): a slab's halo buffer holds its rows of tiles only, so the slabs of a G-way split
    add up to the single-GPU buffer; windows that are not whole rows of tiles have none."""
    for n in (64, 1024):
        whole = lib.pm_khalo_floats(n)
        cells = n**3
        # K: 2 x 10 x 18 floats per tile of 8 x 16 x 16 cells; Hi: 2 x 18 lines, Hj: 2 x 8 lines of n per tile column
        assert whole == (cells // 2048) * 360 + (n // 8) * (n // 16) * (36 + 16) * n
        assert lib.pm_khalo_floats_slab(n, n) == whole
        for G in (2, 4, 8):
            assert G * lib.pm_khalo_floats_slab(n, n // G) == whole
    assert lib.pm_khalo_floats_slab(64, 12) == 0 and lib.pm_khalo_floats_slab(8, 8) == 0 and lib.pm_khalo_floats(8) == 0


def test_no_oracle_in_product():
    """The product package must never import the oracle."""
    pkg = os.path.join(ROOT, "pysco_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "pysco_oracle" not in text, f


def test_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np

    from pysco_amd import _lib, mesh

    with pytest.raises(_lib.PMStepError):
        mesh.TSC(np.zeros((4, 3), np.float32), 16)
