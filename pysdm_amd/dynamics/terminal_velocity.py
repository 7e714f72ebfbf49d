"""Gunn & Kinzer (1949) terminal velocities as a 601-point table + linear interpolation.

Host-side table construction follows PySDM/dynamics/terminal_velocity/gunn_and_kinzer.py:15-137:
radial-basis-function fit (scipy Rbf) through Table 2 of Gunn & Kinzer 1949 on a 10-micron grid
up to 6 mm radius, the sub-40-micron entries replaced by the small-droplet regime of the
temperature/pressure-dependent approximation; the device only sees the table (`a` values,
`b` slopes) through `backend.interpolation`.
"""
import numpy as np
from scipy.interpolate import Rbf

from ..physics.constants import si

# Gunn & Kinzer 1949, Table 2: drop diameter [mm] -> terminal velocity [cm/s]
_GK_DIAMETER_MM = (
    0.078, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.2, 1.4, 1.6, 1.8, 2.0, 2.2, 2.4,
    2.6, 2.8, 3.0, 3.2, 3.4, 3.6, 3.8, 4.0, 4.2, 4.4, 4.6, 4.8, 5.0, 5.2, 5.4, 5.6, 5.8,
)
_GK_VELOCITY_CM_S = (
    18, 27, 72, 117, 162, 206, 247, 287, 327, 367, 403, 464, 517, 565, 609, 649, 690, 727, 757,
    782, 806, 826, 844, 860, 872, 883, 892, 898, 903, 907, 909, 912, 914, 916, 917,
)


def small_droplet_velocity(radius):
    """small-droplet branch (r < 40 um) of the T,p-dependent fit at T=293.15 K, p=1000 hPa;
    operation order as in gunn_and_kinzer.py:146-205 (the mixed cm / m units are the reference's)"""
    si_cm = si.cm
    T = 293.15
    p = 1000 * si.hPa
    p0 = 1013.25 * si.hPa
    rho0 = 1.204
    n = 1.832e-5
    rho = 0.348 * p / T
    l0 = 6.62e-6 * si.cm
    n0 = 1.818e-5
    l = l0 * (n / n0) * (p0 * rho0 / p * rho) ** (1 / 2)
    c4 = (10.5035, 1.08750, -0.133245, -0.00659969)
    r = np.asarray(radius) / si_cm
    f4 = (n0 / n) * (1 + 1.255 * l / r) / (1 + 1.255 * l0 / r)
    log2r = np.log(2 * r)
    sum_r = 0
    for j, coeff in enumerate(c4):
        sum_r = sum_r + coeff * (log2r**j)
    return f4 * np.exp(sum_r) * si_cm


class GunnKinzer1949:  # pylint: disable=too-few-public-methods
    def __init__(self, particulator, small_r_limit=None):
        self.particulator = particulator
        ir = np.array(_GK_DIAMETER_MM) * 1e-3 / 2
        iu = np.array(_GK_VELOCITY_CM_S) / 100
        rbf = Rbf(ir, iu)
        self.factor = 100000
        num = 6 * self.factor // 1000 + 1
        self.minimum_radius = 0
        self.maximum_radius = 0.6 * si.cm
        space, step = np.linspace(self.minimum_radius, self.maximum_radius, num, retstep=True)
        u = np.empty(num)
        u[:] = rbf(space)
        u[0] = 0
        small_r_limit = small_r_limit or 40 * si.um
        small = np.flatnonzero(space[1:] < small_r_limit) + 1
        u[small] = small_droplet_velocity(space[small])
        self.table_a = u
        self.table_b = np.append(np.diff(u), [u[-1] - u[-2]]) / step
        self.a = particulator.backend.Storage.from_ndarray(self.table_a)
        self.b = particulator.backend.Storage.from_ndarray(self.table_b)

    def __call__(self, output, radius):
        r_max = radius.amax()
        if r_max > self.maximum_radius:
            raise ValueError(
                f"Radii can be interpolated up to {self.maximum_radius} m"
                + f" (max value of {r_max} m within input data)"
            )
        self.particulator.backend.interpolation(
            output=output, radius=radius, factor=self.factor, b=self.a, c=self.b
        )


class RogersYau:  # pylint: disable=too-few-public-methods
    """Rogers & Yau eqs 8.5, 8.6, 8.8 (PySDM/dynamics/terminal_velocity/rogers_and_yau.py)"""

    def __init__(self, particulator):
        self.particulator = particulator

    def __call__(self, output, radius):
        self.particulator.backend.terminal_velocity(values=output.data, radius=radius.data)


class PowerSeries:  # pylint: disable=too-few-public-methods
    """sum of power laws in the particle volume with user-given coefficients
    (PySDM/dynamics/terminal_velocity/power_series.py:18-35)"""

    def __init__(self, particulator, *, prefactors=None, powers=None):
        from ..physics import constants as const  # pylint: disable=import-outside-toplevel

        self.particulator = particulator
        self.prefactors = np.array(prefactors or [2.0e-1 * si.m / si.s / np.sqrt(si.m)])
        self.powers = np.array(powers or [1 / 6])
        assert len(self.prefactors) == len(self.powers)
        for i, power in enumerate(self.powers):
            self.prefactors[i] *= const.PI_4_3**power / si.um ** (3 * power)

    def __call__(self, output, radius):
        self.particulator.backend.power_series(
            values=output.data, radius=radius.data, num_terms=len(self.powers),
            prefactors=self.prefactors, powers=self.powers)
