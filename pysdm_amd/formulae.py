"""`Formulae`: the options + constants object a PySDM-shaped backend is constructed with (a
stand-in for PySDM's own where PySDM is absent; under PySDM the real one is passed in).

Collision-path subset of PySDM/formulae.py:27-67 (same keyword names: `seed`, `constants`,
`terminal_velocity`, `fragmentation_function`, `handle_all_breakups`,
`particle_shape_and_density`, `particle_advection`); everything unrelated to the path is absent.
"""
from types import SimpleNamespace

import numpy as np

from .physics import constants as _const


class _Trivia:  # PySDM/physics/trivia.py:19-28
    @staticmethod
    def volume(radius):
        return _const.PI_4_3 * np.power(radius, 3)

    @staticmethod
    def radius(volume):
        return np.power(volume / _const.PI_4_3, _const.ONE_THIRD)


class _LiquidSpheres:  # PySDM/physics/particle_shape_and_density/liquid_spheres.py:9-23
    __name__ = "LiquidSpheres"

    @staticmethod
    def supports_mixed_phase(_=None):
        return False

    @staticmethod
    def mass_to_volume(mass):
        return mass / _const.rho_w

    @staticmethod
    def volume_to_mass(volume):
        return _const.rho_w * volume


class _ImplicitInSpace:  # PySDM/physics/particle_advection/implicit_in_space.py:11-13
    __name__ = "ImplicitInSpace"
    scheme_id = 0  # SDM scheme code of sdm_calculate_displacement

    @staticmethod
    def displacement(position_in_cell, c_l, c_r):
        return (c_l * (1 - position_in_cell) + c_r * position_in_cell) / (1 - c_r + c_l)


class _ExplicitInSpace:  # PySDM/physics/particle_advection/explicit_in_space.py:11-13
    __name__ = "ExplicitInSpace"
    scheme_id = 1

    @staticmethod
    def displacement(position_in_cell, c_l, c_r):
        return c_l * (1 - position_in_cell) + c_r * position_in_cell


class Formulae:  # pylint: disable=too-few-public-methods,too-many-arguments
    def __init__(
        self,
        *,
        constants=None,
        seed=None,
        fastmath=True,
        fragmentation_function="AlwaysN",
        particle_shape_and_density="LiquidSpheres",
        terminal_velocity="GunnKinzer1949",
        handle_all_breakups=False,
        particle_advection="ImplicitInSpace",
    ):
        if particle_shape_and_density != "LiquidSpheres":
            raise NotImplementedError(particle_shape_and_density)
        if terminal_velocity not in ("GunnKinzer1949", "RogersYau", "PowerSeries"):
            raise NotImplementedError(terminal_velocity)
        values = {
            k: getattr(_const, k)
            for k in dir(_const)
            if not k.startswith("_") and isinstance(getattr(_const, k), (int, float))
        }
        values.update(constants or {})
        self.constants = SimpleNamespace(**values)
        self.seed = seed if seed is not None else _const.default_random_seed
        self.fastmath = fastmath
        self.fragmentation_function = fragmentation_function
        self.handle_all_breakups = handle_all_breakups
        self.trivia = _Trivia()
        self.particle_shape_and_density = _LiquidSpheres()
        self.terminal_velocity = terminal_velocity
        schemes = {"ImplicitInSpace": _ImplicitInSpace, "ExplicitInSpace": _ExplicitInSpace}
        if particle_advection not in schemes:
            raise NotImplementedError(particle_advection)
        self.particle_advection = schemes[particle_advection]()

    def __str__(self):
        return f"Formulae(seed={self.seed}, fragmentation_function={self.fragmentation_function})"
