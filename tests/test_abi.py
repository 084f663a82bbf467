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
