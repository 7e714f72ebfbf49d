"""`Displacement`: advection by a prescribed Courant-number field plus sedimentation -- the step
that precedes collisions in 1-D/2-D/3-D set-ups (it is what moves super-droplets between cells
and unsorts the state the collision step then re-sorts).

Host-side mirror of PySDM/dynamics/displacement.py:19-153: same constructor keywords, attributes
(`courant`, `displacement`, `precipitation_mass_in_last_step`) and method names; the per-droplet
arithmetic runs in the backend: method by method (`calculate_displacement`,
`flag_precipitated`, `flag_out_of_column`, Storage ops) or, where the backend offers
`displacement_step`, all sub-steps in one library call.
"""
from collections import namedtuple

import numpy as np

from . import builder_owned

DEFAULTS = namedtuple("_", ("rtol", "adaptive"))(rtol=1e-2, adaptive=True)


@builder_owned
class Displacement:  # pylint: disable=too-many-instance-attributes
    def __init__(self, enable_sedimentation=False, precipitation_counting_level_index: int = 0,
                 adaptive=DEFAULTS.adaptive, rtol=DEFAULTS.rtol, fused=None):
        self.particulator = None
        self.enable_sedimentation = enable_sedimentation
        self.dimension = None
        self.grid = None
        self.courant = None
        self.displacement = None
        self.temp = None
        self.precipitation_mass_in_last_step = 0
        self.precipitation_counting_level_index = precipitation_counting_level_index
        self.adaptive = adaptive
        self.rtol = rtol
        self._n_substeps = 1
        # None: take the backend's one-call route if it has one; False: method by method
        self.fused = fused

    def register(self, builder):
        builder.request_attribute("relative fall velocity")
        self.particulator = builder.particulator
        grid = tuple(int(g) for g in builder.particulator.environment.mesh.grid)
        self.dimension = len(grid)
        if self.dimension not in (1, 2, 3):
            raise NotImplementedError()
        Storage = self.particulator.Storage
        self.grid = Storage.from_ndarray(np.asarray(grid, dtype=np.int64))
        # Arakawa-C: component d lives on cell faces normal to d -> one more point along d
        self.courant = tuple(
            Storage.from_ndarray(np.full(
                tuple(g + (1 if axis == d else 0) for axis, g in enumerate(grid)), np.nan))
            for d in range(self.dimension)
        )
        n_sd = self.particulator.n_sd
        self.displacement = Storage.from_ndarray(np.zeros((self.dimension, n_sd)))
        self.temp = Storage.from_ndarray(np.zeros((self.dimension, n_sd), dtype=np.int64))

    def upload_courant_field(self, courant_field):
        for component, values in zip(self.courant, courant_field):
            component.upload(values)
        if not self.adaptive:
            return
        # sub-steps doubled until implicit and explicit Euler agree to rtol (Arabas et al. 2015,
        # eqs 13-16): |(I - E) / E| = 1 / (1 / max|dC| - 1)
        n_substeps = 1
        while True:
            error_estimate = 0
            for axis, values in enumerate(courant_field):
                delta = np.amax(np.abs(np.diff(values, axis=axis))) / n_substeps
                error_estimate = max(error_estimate, 0 if delta == 0 else 1 / (1 / delta - 1))
            if error_estimate < self.rtol:
                break
            n_substeps *= 2
        self._n_substeps = n_substeps

    def __call__(self):
        attributes = self.particulator.attributes
        cell_origin = attributes["cell origin"]
        position_in_cell = attributes["position in cell"]
        self.precipitation_mass_in_last_step = 0.0
        if self.fused is not False and hasattr(self.particulator.backend, "displacement_step"):
            self.precipitation_mass_in_last_step = self.particulator.backend.displacement_step(self)
            for key in ("position in cell", "cell origin", "cell id"):
                attributes.mark_updated(key)
            return
        for _ in range(self._n_substeps):
            self.calculate_displacement(self.displacement, self.courant, cell_origin,
                                        position_in_cell)
            self.update_position(position_in_cell, self.displacement)
            if self.enable_sedimentation:
                self.precipitation_mass_in_last_step += self.particulator.remove_precipitated(
                    displacement=self.displacement,
                    precipitation_counting_level_index=self.precipitation_counting_level_index,
                )
            self.particulator.flag_out_of_column()
            self.update_cell_origin(cell_origin, position_in_cell)
            self.boundary_condition(cell_origin)
            self.particulator.recalculate_cell_id()
        for key in ("position in cell", "cell origin", "cell id"):
            attributes.mark_updated(key)

    def calculate_displacement(self, displacement, courant, cell_origin, position_in_cell):
        self.particulator.calculate_displacement(
            displacement=displacement, courant=courant, cell_origin=cell_origin,
            position_in_cell=position_in_cell, n_substeps=self._n_substeps,
        )
        if self.enable_sedimentation:
            vertical = displacement[self.dimension - 1, :]
            dt_over_dz = self.particulator.dt / self._n_substeps / self.particulator.mesh.dz
            vertical *= 1 / dt_over_dz
            vertical -= self.particulator.attributes["relative fall velocity"]
            vertical *= dt_over_dz

    @staticmethod
    def update_position(position_in_cell, displacement):
        position_in_cell += displacement

    def update_cell_origin(self, cell_origin, position_in_cell):
        whole_cells = self.temp
        whole_cells.floor(position_in_cell)
        cell_origin += whole_cells
        position_in_cell -= whole_cells

    def boundary_condition(self, cell_origin):
        cell_origin %= self.grid
