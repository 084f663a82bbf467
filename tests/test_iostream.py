"""Parameter files and snapshot side files (pysco_amd/iostream.py) against what the reference's iostream.py made of
the same inputs (tests/golden/param_io.json, oracle/gen_golden_io.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from pysco_amd import initial_conditions, iostream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "param_io.json")))


def _typed(series):
    return {str(k): [type(v).__name__, v] for k, v in series.items()}


def test_read_param_file_syntax_cases(gold, tmp_path):
    path = tmp_path / "p.ini"
    path.write_text(gold["synthetic_text"])
    got = _typed(iostream.read_param_file(str(path)))
    assert list(got) == list(gold["synthetic_parsed"])  # same keys, same order
    for k, (tname, value) in gold["synthetic_parsed"].items():
        assert got[k][0] == tname, k
        assert got[k][1] == value, k


def test_example_param_file_semantics(gold):
    """examples/param.ini of this repository carries the reference example's settings (config 1 of BASELINE.json):
    parsed, it must be what the reference's reader made of the reference's file."""
    got = _typed(iostream.read_param_file(os.path.join(ROOT, "examples", "param.ini")))
    want = gold["example_parsed"]
    assert set(got) == set(want)
    for k, (tname, value) in want.items():
        if k in ("base", "power_spectrum_file"):  # paths are this repository's
            continue
        assert got[k] == [tname, value], k


def test_parquet_snapshot_and_side_file(gold, tmp_path):
    order = [line.split("=", 1)[0] for line in gold["side_file_text"].splitlines()]  # key order of the writer's input
    def value(k):
        tname, v = gold["side_file_param"][k]
        return eval(v, {"np": np}) if tname.startswith("float") and isinstance(v, str) else v  # np.float32(...) reprs

    param = {k: (str(tmp_path) if k == "base" else value(k)) for k in order}
    pos = np.arange(24, dtype=np.float32).reshape(8, 3) / 24
    iostream.write_snapshot_particles(pos, -pos, param)
    side = tmp_path / "output_00002" / f"param_{param['extra']}_00002.txt"
    assert side.read_text().replace(str(tmp_path), "BASE") == gold["side_file_text"]
    back = _typed(iostream.read_param_file(str(side)))
    for k, (tname, value) in gold["side_file_parsed"].items():
        assert back[k] == [tname, value], k
    import pyarrow.parquet as pq

    t = pq.read_table(tmp_path / "output_00002" / f"particles_{param['extra']}.parquet")
    assert t.column_names == gold["parquet_columns"] and [str(x) for x in t.schema.types] == gold["parquet_types"]
    p2, v2 = iostream.read_snapshot_particles_parquet(str(tmp_path / "output_00002" / f"particles_{param['extra']}.parquet"))
    assert p2.dtype == np.float32 and np.array_equal(p2, pos) and np.array_equal(v2, -pos)


@pytest.mark.parametrize("fmt", ["parquet", "hdf5"])
def test_restart_reads_snapshot_and_parameters(fmt, tmp_path):
    rng = np.random.default_rng(0)
    pos, vel = rng.random((100, 3), dtype=np.float32), rng.normal(size=(100, 3)).astype(np.float32)
    written = {"base": str(tmp_path), "extra": "newton_fft_ncoarse4", "i_snap": 3, "aexp": 0.25, "nsteps": 17,
               "t": -3.5, "output_snapshot_format": fmt, "theory": "newton", "boxlen": 100.0, "npart": 100,
               "write_snapshot": True, "z_out": "[1, 0]", "unit_t": 1.5e17}
    iostream.write_snapshot_particles(pos, vel, written)
    param = {"base": str(tmp_path), "extra": "newton_fft_ncoarse4", "output_snapshot_format": fmt.upper(),
             "initial_conditions": 3, "aexp": 0.02, "nsteps": 0, "nthreads": 1}
    p2, v2 = initial_conditions.generate(param, tables=None)
    assert np.array_equal(p2, pos) and np.array_equal(v2, vel)
    assert param["nsteps"] == 17 and param["i_snap"] == 3 and float(param["aexp"]) == 0.25 and float(param["t"]) == -3.5
    assert str(param["theory"]) == "newton" and bool(param["write_snapshot"]) is True
