"""pysco_amd/hdf5lite.py (the HDF5 reader / writer used when h5py is not installed): round trips, a snapshot written
by the genuine h5py the way the reference writes one (tests/golden/snapshot_h5py.h5, oracle/gen_golden_h5.py), and --
where this image's conda h5py is present -- the genuine library reading what hdf5lite wrote.  CPU only."""
import json
import os
import subprocess

import numpy as np
import pytest

from pysco_amd import hdf5lite

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONDA_PY = "/opt/conda/bin/python3.9"


def test_reads_a_snapshot_written_by_h5py():
    with hdf5lite.File(os.path.join(ROOT, "tests", "golden", "snapshot_h5py.h5"), "r") as f:
        assert sorted(f.keys()) == ["position", "velocity"]
        pos, vel = f["position"][:], f["velocity"][:]
        attrs = {k: f.attrs[k] for k in f.attrs.keys()}
    rng = np.random.default_rng(7)
    assert pos.dtype == np.float32 and np.array_equal(pos, rng.random((257, 3), dtype=np.float32))
    assert np.array_equal(vel, rng.normal(size=(257, 3)).astype(np.float32))
    assert len(attrs) == 59
    assert attrs["theory"] == "newton" and attrs["z_out"] == "[10, 5, 2, 1, 0.5, 0]"
    assert attrs["npart"] == 2097152 and isinstance(attrs["npart"], np.integer)
    assert attrs["t"] == -12.25 and isinstance(attrs["t"], np.floating)
    assert attrs["write_snapshot"] == True and attrs["fixed_ICS"] == False  # noqa: E712
    assert isinstance(attrs["write_snapshot"], (bool, np.bool_))
    assert attrs["tolerance"] == np.float32(1e-3) and attrs["tolerance"].dtype == np.float32
    assert attrs["base"] == "examples/boxlen100_n128_lcdmw7v2_00000/"


def _write(path):
    rng = np.random.default_rng(3)
    pos, vel = rng.random((1000, 3), dtype=np.float32), rng.normal(size=(1000, 3)).astype(np.float32)
    param = {"theory": "newton", "aexp": 0.5, "nsteps": 12, "npart": 1000, "write_snapshot": True, "save_pk": False,
             "t": np.float64(-1.5), "tolerance": np.float32(2.5e-3), "z_out": "[1, 0]", "extra": "newton_fft_ncoarse5",
             "unicode": "héllo", "boxlen": 100.0, "i_snap": np.int64(2)}
    with hdf5lite.File(path, "w") as f:
        f.create_dataset("position", data=pos)
        f.create_dataset("velocity", data=vel)
        for k, v in param.items():
            f.attrs[k] = v
    return pos, vel, param


def test_round_trip(tmp_path):
    path = str(tmp_path / "s.h5")
    pos, vel, param = _write(path)
    with hdf5lite.File(path, "r") as f:
        assert np.array_equal(f["position"][:], pos) and np.array_equal(f["velocity"][:], vel)
        for k, v in param.items():
            assert f.attrs[k] == v, k
        assert f.attrs["tolerance"].dtype == np.float32 and isinstance(f.attrs["nsteps"], np.integer)


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no conda h5py in this environment")
def test_genuine_h5py_reads_what_hdf5lite_wrote(tmp_path):
    path = str(tmp_path / "s.h5")
    pos, vel, param = _write(path)
    code = (
        "import h5py, json, numpy as np, sys\n"
        "f = h5py.File(sys.argv[1], 'r')\n"
        "a = {k: (v.item() if hasattr(v, 'item') else v) for k, v in f.attrs.items()}\n"
        "print(json.dumps({'keys': sorted(f.keys()), 'attrs': a, 'pos_sum': float(f['position'][:].astype('f8').sum()),\n"
        "                  'vel_dtype': str(f['velocity'].dtype), 'vel_shape': list(f['velocity'].shape),\n"
        "                  'vel_last': f['velocity'][-1].tolist(), 'bool_dtype': str(f.attrs['write_snapshot'].dtype)}))\n")
    out = subprocess.run([CONDA_PY, "-W", "ignore", "-c", code, path], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["keys"] == ["position", "velocity"]
    assert got["vel_dtype"] == "float32" and got["vel_shape"] == [1000, 3]
    assert got["vel_last"] == vel[-1].tolist()
    assert got["pos_sum"] == pytest.approx(float(pos.astype("f8").sum()), rel=1e-12)
    assert got["bool_dtype"] == "bool"
    for k, v in param.items():
        assert got["attrs"][k] == (v.item() if hasattr(v, "item") else v), k
