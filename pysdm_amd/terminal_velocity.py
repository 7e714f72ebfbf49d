"""Terminal-velocity laws: what turns the radius column into the "relative fall velocity" column.

* `GunnKinzerTable` (the default of the reference, `Formulae(terminal_velocity="GunnKinzer1949")`):
  a 601-point table on a 10-micrometre grid up to 6 mm radius with linear interpolation on the
  device (`sdm_interpolation`).  The table is built on the host exactly as
  PySDM/dynamics/terminal_velocity/gunn_and_kinzer.py:99-137 builds it - a radial-basis-function
  fit through Table 2 of Gunn & Kinzer (1949), with the entries below 40 micrometres replaced by the
  small-droplet regime of their temperature / pressure dependent fit (:146-205) - because the
  table values are inputs the collision probabilities depend on bit for bit.
* `RogersYau`, `PowerSeries`: closed forms evaluated on the device (`sdm_terminal_velocity`,
  `sdm_power_series`; terminal_velocity_methods.py:32-66).
"""
import ctypes
import functools

import numpy as np

from .physics import constants as const
from .physics.constants import si

# Gunn & Kinzer 1949, Table 2: (drop diameter [mm], terminal velocity [cm/s])
_TABLE_2 = (
    (0.078, 18), (0.1, 27), (0.2, 72), (0.3, 117), (0.4, 162), (0.5, 206), (0.6, 247),
    (0.7, 287), (0.8, 327), (0.9, 367), (1.0, 403), (1.2, 464), (1.4, 517), (1.6, 565),
    (1.8, 609), (2.0, 649), (2.2, 690), (2.4, 727), (2.6, 757), (2.8, 782), (3.0, 806),
    (3.2, 826), (3.4, 844), (3.6, 860), (3.8, 872), (4.0, 883), (4.2, 892), (4.4, 898),
    (4.6, 903), (4.8, 907), (5.0, 909), (5.2, 912), (5.4, 914), (5.6, 916), (5.8, 917),
)
TABLE_POINTS_PER_METRE = 100000      # `factor` of the interpolation
TABLE_TOP = 0.6 * si.cm
SMALL_DROPLET_LIMIT = 40 * si.um


def _small_droplet_regime(radius):
    """r < 40 um branch at 293.15 K, 1000 hPa; term by term as gunn_and_kinzer.py:146-205 (the
    mixed cm / m units are the reference's own and are kept: the numbers must agree)"""
    temperature, pressure, pressure_0 = 293.15, 1000 * si.hPa, 1013.25 * si.hPa
    density_0, viscosity, viscosity_0 = 1.204, 1.832e-5, 1.818e-5
    density = 0.348 * pressure / temperature
    path_0 = 6.62e-6 * si.cm
    path = path_0 * (viscosity / viscosity_0) * (pressure_0 * density_0 / pressure * density) ** (
        1 / 2)
    r_cm = np.asarray(radius) / si.cm
    slip = (viscosity_0 / viscosity) * (1 + 1.255 * path / r_cm) / (1 + 1.255 * path_0 / r_cm)
    log_2r = np.log(2 * r_cm)
    series = 0
    for power, coefficient in enumerate((10.5035, 1.08750, -0.133245, -0.00659969)):
        series = series + coefficient * (log_2r**power)
    return slip * np.exp(series) * si.cm


@functools.lru_cache(maxsize=None)
def gunn_kinzer_table():
    """(values a, slopes b): u(r) = a[i] + (r - r_i) b[i] with i = int(1e5 r)"""
    from scipy.interpolate import Rbf  # pylint: disable=import-outside-toplevel

    diameters_mm, velocities_cm_s = (np.array(column) for column in zip(*_TABLE_2))
    fit = Rbf(diameters_mm * 1e-3 / 2, velocities_cm_s / 100)
    n_points = 6 * TABLE_POINTS_PER_METRE // 1000 + 1
    nodes, spacing = np.linspace(0, TABLE_TOP, n_points, retstep=True)
    values = np.empty(n_points)
    values[:] = fit(nodes)
    values[0] = 0
    small = np.flatnonzero(nodes[1:] < SMALL_DROPLET_LIMIT) + 1
    values[small] = _small_droplet_regime(nodes[small])
    slopes = np.append(np.diff(values), [values[-1] - values[-2]]) / spacing
    values.setflags(write=False)
    slopes.setflags(write=False)
    return values, slopes


class GunnKinzerTable:
    factor = TABLE_POINTS_PER_METRE
    maximum_radius = TABLE_TOP

    def __init__(self, engine):
        values, slopes = gunn_kinzer_table()
        self.a = engine.upload(values)
        self.b = engine.upload(slopes)
        self.length = len(values)

    def evaluate(self, engine, out, radius, n, check_range=True):
        if check_range:
            largest = engine.scalar_out("sdm_reduce_f64", ctypes.c_double, 1,
                                        radius, n)
            if largest > self.maximum_radius:
                raise ValueError(f"Radii can be interpolated up to {self.maximum_radius} m"
                                 f" (max value of {largest} m within input data)")
        engine.call("sdm_interpolation", out, radius, n, float(self.factor), self.a, self.b,
                    self.length)


class RogersYau:
    """Rogers & Yau eqs 8.5, 8.6, 8.8 (three radius regimes)"""

    def __init__(self, engine=None):  # pylint: disable=unused-argument
        self.consts = (const.ROGERS_YAU_TERM_VEL_SMALL_K, const.ROGERS_YAU_TERM_VEL_MEDIUM_K,
                       const.ROGERS_YAU_TERM_VEL_LARGE_K,
                       const.ROGERS_YAU_TERM_VEL_SMALL_R_LIMIT,
                       const.ROGERS_YAU_TERM_VEL_MEDIUM_R_LIMIT)

    def evaluate(self, engine, out, radius, n):
        engine.call("sdm_terminal_velocity", out, radius, n, self.consts)


class PowerSeries:
    """sum_j prefactor_j * volume^power_j, volume in cubic micrometres
    (PySDM/dynamics/terminal_velocity/power_series.py:18-35)"""

    def __init__(self, engine=None, *, prefactors=None, powers=None):  # pylint: disable=unused-argument
        self.powers = np.array(powers or [1 / 6], dtype=float)
        self.prefactors = np.array(prefactors or [2.0e-1 * si.m / si.s / np.sqrt(si.m)],
                                   dtype=float)
        if len(self.prefactors) != len(self.powers):
            raise ValueError("one prefactor per power")
        for j, power in enumerate(self.powers):
            self.prefactors[j] *= const.PI_4_3**power / si.um ** (3 * power)

    def evaluate(self, engine, out, radius, n):
        engine.call("sdm_power_series", out, radius, n, len(self.powers),
                    [float(v) for v in self.prefactors], [float(v) for v in self.powers])


LAWS = {"GunnKinzer1949": GunnKinzerTable, "RogersYau": RogersYau, "PowerSeries": PowerSeries}
