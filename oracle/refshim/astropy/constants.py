"""CODATA-2018 values of the three constants the reference imports from astropy.constants
(utils.py:11, solver.py:14): only `.value` is used."""


class _C:
    def __init__(self, value):
        self.value = value


G = _C(6.6743e-11)  # m3 / (kg s2)
c = _C(299792458.0)  # m / s
pc = _C(3.0856775814913674e16)  # m
