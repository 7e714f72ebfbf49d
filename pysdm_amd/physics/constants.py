"""SI unit multipliers and the physical constants the collision path uses.

Values as in the reference's catalogue (PySDM/physics/constants.py:32-70,
PySDM/physics/constants_defaults.py:190,359-366,667), expressed in SI base units.
"""
import math
import os
import time
from types import SimpleNamespace


def _make_si():
    prefixes = {"n": 1e-9, "u": 1e-6, "m": 1e-3, "c": 1e-2, "d": 1e-1, "": 1.0, "h": 1e2,
                "k": 1e3}
    long_prefixes = {"nano": 1e-9, "micro": 1e-6, "milli": 1e-3, "centi": 1e-2, "deci": 1e-1,
                     "": 1.0, "hecto": 1e2, "kilo": 1e3}
    short_units = {"m": 1.0, "g": 1e-3, "s": 1.0, "J": 1.0, "K": 1.0, "Pa": 1.0, "l": 1e-3,
                   "N": 1.0, "W": 1.0, "Hz": 1.0, "mol": 1.0}
    long_units = {"metre": 1.0, "meter": 1.0, "gram": 1e-3, "second": 1.0, "joule": 1.0,
                  "kelvin": 1.0, "pascal": 1.0, "litre": 1e-3, "liter": 1e-3, "newton": 1.0,
                  "watt": 1.0, "hertz": 1.0, "mole": 1.0}
    table = {"dimensionless": 1.0, "min": 60.0, "minute": 60.0, "minutes": 60.0,
             "h": 3600.0, "hour": 3600.0, "hours": 3600.0, "day": 86400.0}
    for p_name, p_val in prefixes.items():
        for u_name, u_val in short_units.items():
            table.setdefault(p_name + u_name, p_val * u_val)
    for p_name, p_val in long_prefixes.items():
        for u_name, u_val in long_units.items():
            table[p_name + u_name] = p_val * u_val
            table[p_name + u_name + "s"] = p_val * u_val
    return SimpleNamespace(**table)


si = _make_si()

PI = math.pi
PI_4_3 = PI * 4 / 3
ONE_THIRD = 1 / 3
TWO_THIRDS = 2 / 3
ONE_AND_A_HALF = 3 / 2
CM = 1 * si.cm

rho_w = 1 * si.kilograms / si.litres
sgm_w = 0.072 * si.joule / si.metre**2

STRAUB_E_D1 = 0.04 * si.cm
STRAUB_MU2 = 0.095 * si.cm
VEDDER_1987_b = 89 / 880
VEDDER_1987_A = 993 / 880 / 3 / VEDDER_1987_b

# PySDM/physics/constants.py:50-54
default_random_seed = 44 if "CI" in os.environ else time.time_ns()

# Rogers & Yau terminal velocity (PySDM/physics/constants_defaults.py:625-635)
ROGERS_YAU_TERM_VEL_SMALL_K = 1.19e6 / si.cm / si.s
ROGERS_YAU_TERM_VEL_MEDIUM_K = 8e3 / si.s
ROGERS_YAU_TERM_VEL_LARGE_K = 2.01e3 * si.cm**0.5 / si.s
ROGERS_YAU_TERM_VEL_SMALL_R_LIMIT = 35 * si.um
ROGERS_YAU_TERM_VEL_MEDIUM_R_LIMIT = 600 * si.um


def namespace(overrides=None):
    """the numeric constants of this module as one namespace, optionally with overrides"""
    values = {name: value for name, value in globals().items()
              if not name.startswith("_") and isinstance(value, (int, float))
              and not isinstance(value, bool)}
    values.update(overrides or {})
    return SimpleNamespace(**values)
