"""Background cosmology tables (pysco_amd/cosmotable.py, astropy-free) against the reference's cosmotable.generate
run with the genuine astropy 4.3.1 (tests/golden/cosmotable_*.npz, oracle/gen_golden_cosmo.py).  CPU only."""
import ast
import os

import numpy as np
import pytest

from pysco_amd import cosmotable

NAMES = ["lna_of_t", "t_of_lna", "H", "d1", "f1", "d2", "f2", "d3a", "f3a", "d3b", "f3b", "d3c", "f3c"]


@pytest.mark.parametrize("tag", ["lcdm", "w0wa", "parametrized", "norad"])
def test_tables_vs_reference(golden, tag, tmp_path):
    g = golden(f"cosmotable_{tag}")
    param = ast.literal_eval(str(g["param_json"]))
    param.update(base=str(tmp_path), extra="t")
    tables = cosmotable.generate(param)
    assert len(tables) == 13
    assert param["Om_r"] == pytest.approx(float(g["Om_r"]), rel=1e-12, abs=1e-18)
    assert param["Om_lambda"] == pytest.approx(float(g["Om_lambda"]), rel=1e-12)
    lna = g["lna"]
    # time <-> scale factor: the same trapezoid on the same grid
    np.testing.assert_allclose(tables[1](lna), g["t_of_lna"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(tables[0](g["t"]), g["lna_of_t"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(tables[2](lna), g["H"], rtol=1e-12)
    # growth factors and rates: the same RK45 at rtol 1e-13 (different SciPy releases: 1e-9 is ample)
    for k in range(3, 13):
        np.testing.assert_allclose(tables[k](lna), g[NAMES[k]], rtol=2e-9, err_msg=NAMES[k])
    rows = np.loadtxt(os.path.join(tmp_path, "evolution_table_pysco.txt"))
    assert rows.shape == (100_000, 13)
    np.testing.assert_allclose(rows[::250], g["table_rows"], rtol=2e-9, atol=1e-13)


def test_cosmology_functions():
    c = cosmotable.FlatW0WaCDM(72, 0.25733, 2.726, 3.044, -1.0, 0.0)
    # astropy 4.3.1 values for this cosmology (oracle/gen_golden_cosmo.py session)
    assert c.Ogamma0 == pytest.approx(4.773901471708128e-05, rel=1e-12)
    assert c.Onu0 == pytest.approx(3.3002681441908394e-05, rel=1e-12)
    assert c.Ode0 == pytest.approx(0.7425892583038409, rel=1e-14)
    assert float(c.efunc(49.0)) == pytest.approx(180.75294794346232, rel=1e-13)
    assert float(c.Om(1.0)) == pytest.approx(0.7345671657247682, rel=1e-13)
    assert float(c.Ogamma(1.0)) == pytest.approx(0.00027254896619297703, rel=1e-12)
    assert float(c.Onu(1.0)) == pytest.approx(0.0001884170999735752, rel=1e-12)
    assert float(c.Ode(1.0)) == pytest.approx(0.2649718682090653, rel=1e-13)
