"""Parameter files, particle snapshots and the ascii P(k) file: drop-in for the reference's iostream.py.

Host-side I/O around the hot path.  Device tensors are accepted wherever the
reference takes NumPy arrays (they are copied to the host for writing).  Parquet goes through pyarrow like the
reference; HDF5 through h5py when it is importable and through the minimal built-in reader / writer
(pysco_amd/hdf5lite.py: the two datasets + root attributes layout of iostream.py:230-265) when it is not.
"""
from __future__ import annotations

import logging
import os
from typing import Tuple

import numpy as np


def _host(x) -> np.ndarray:
    if type(x).__module__.startswith("torch"):
        return x.detach().cpu().numpy()
    return np.asarray(x)


# ---- parameter file (iostream.py:13-69) ------------------------------------------------------------------


def _parse_value(text):
    """One value of a parameter file with the reference's rules (iostream.py:48-67): missing -> False; true/false in
    any case -> bool; anything Python evaluates to a non-list (numbers, 128**3, booleans) -> that value; everything
    else (names, paths, lists like z_out) stays a string."""
    if text is None or text == "":
        return False
    if text.casefold() == "true":
        return True
    if text.casefold() == "false":
        return False
    try:
        value = eval(text, {"__builtins__": {}}, {})  # noqa: S307  (arithmetic of a trusted config file, no names)
    except Exception:
        return text
    return text if isinstance(value, list) else value


def read_param_file(name: str):
    """`key = value  # comment` lines -> parameter container (a pandas Series like the reference's, iostream.py:13)."""
    import pandas as pd

    out = {}
    with open(name) as fh:
        for line in fh:
            line = line.split("#", 1)[0]
            if not line.strip():
                continue
            key, sep, value = line.partition("=")
            out[key.strip()] = _parse_value(value.strip() if sep else None)
    return pd.Series(out, dtype=object)


def write_param_file(filename: str, param) -> None:
    """The `key=value` side file of a Parquet snapshot: what param.to_csv(sep="=", header=False) writes
    (iostream.py:160-164, initial_conditions.py:266-270); read back by read_param_file on restart."""
    import pandas as pd

    pd.Series(dict(param.items()), dtype=object).to_csv(filename, sep="=", header=False)


# ---- particle snapshots (iostream.py:72-265) --------------------------------------------------------------


def read_snapshot_particles_parquet(filename: str) -> Tuple[np.ndarray, np.ndarray]:
    """(position [Np,3], velocity [Np,3]) float32 from the six columns x, y, z, vx, vy, vz (iostream.py:102-133)."""
    import pyarrow.parquet as pq

    t = pq.read_table(filename, columns=["x", "y", "z", "vx", "vy", "vz"])
    cols = [np.asarray(t.column(c).to_numpy(), dtype=np.float32) for c in ("x", "y", "z", "vx", "vy", "vz")]
    return np.ascontiguousarray(np.stack(cols[:3], axis=1)), np.ascontiguousarray(np.stack(cols[3:], axis=1))


def write_snapshot_particles_parquet(filename: str, position, velocity) -> None:
    """Six float32 columns x, y, z, vx, vy, vz (iostream.py:187-226)."""
    import pyarrow as pa
    import pyarrow.parquet as pq

    p, v = _host(position), _host(velocity)
    pq.write_table(pa.table({"x": p[:, 0], "y": p[:, 1], "z": p[:, 2], "vx": v[:, 0], "vy": v[:, 1], "vz": v[:, 2]}),
                   filename)


def _h5():
    try:
        import h5py

        return h5py
    except ImportError:
        from . import hdf5lite

        return hdf5lite


def read_snapshot_particles_hdf5(filename: str) -> Tuple[np.ndarray, np.ndarray]:
    """Datasets "position" and "velocity" (iostream.py:72-99)."""
    with _h5().File(filename, "r") as h5r:
        return np.asarray(h5r["position"][:]), np.asarray(h5r["velocity"][:])


def read_snapshot_attributes_hdf5(filename: str) -> dict:
    """The root attributes of a snapshot: every `param` key the writer had (initial_conditions.py:96-101)."""
    with _h5().File(filename, "r") as h5r:
        return {k: h5r.attrs[k] for k in h5r.attrs.keys()}


def write_snapshot_particles_hdf5(filename: str, position, velocity, param) -> None:
    """Datasets "position", "velocity" + every `param` item as a root attribute (iostream.py:229-265)."""
    with _h5().File(filename, "w") as h5f:
        h5f.create_dataset("position", data=_host(position))
        h5f.create_dataset("velocity", data=_host(velocity))
        for key, item in param.items():
            if key.startswith("_") or item is None or type(item).__module__.startswith("torch"):
                continue  # private bookkeeping of this library (device tensors, unset policies), not parameters
            h5f.attrs[key] = item


def write_snapshot_particles(position, velocity, param) -> None:
    """{base}/output_{i_snap:05d}/particles_{extra}.{parquet|h5} (+ the parameter side file for Parquet),
    iostream.py:135-184."""
    fmt = param["output_snapshot_format"].casefold()
    directory = f"{param['base']}/output_{param['i_snap']:05d}"
    os.makedirs(directory, exist_ok=True)
    if fmt == "parquet":
        filename = f"{directory}/particles_{param['extra']}.parquet"
        write_snapshot_particles_parquet(filename, position, velocity)
        param_filename = f"{directory}/param_{param['extra']}_{param['i_snap']:05d}.txt"
        write_param_file(param_filename, _public(param))
        logging.warning(f"Parameter file written at ...{param_filename=}")
    elif fmt == "hdf5":
        filename = f"{directory}/particles_{param['extra']}.h5"
        write_snapshot_particles_hdf5(filename, position, velocity, param)
    else:
        raise NotImplementedError(f"{param['output_snapshot_format']=}, should be 'parquet' or 'hdf5'")
    logging.warning(f"Snapshot written at ...{filename=} {param['aexp']=}")


def _public(param) -> dict:
    return {k: v for k, v in param.items()
            if not k.startswith("_") and v is not None and not type(v).__module__.startswith("torch")}


# ---- P(k) (iostream.py:268-304) ---------------------------------------------------------------------------


def write_power_spectrum_to_ascii_file(k, Pk, Nmodes, param) -> None:
    out_dir = f"{param['base']}/power"
    os.makedirs(out_dir, exist_ok=True)
    output_pk = f"{out_dir}/pk_{param['extra']}_{param['nsteps']:05d}.dat"
    logging.warning(f"Write P(k) in {output_pk}")
    np.savetxt(output_pk, np.c_[k, Pk, Nmodes],
               header=f"aexp = {param['aexp']}\nboxlen = {param['boxlen']} Mpc/h \nnpart = {param['npart']} \n"
                      "k [h/Mpc] P(k) [Mpc/h]^3 Nmodes")
