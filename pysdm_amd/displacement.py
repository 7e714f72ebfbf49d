"""`DisplacementRunner`: advection of super-droplets by a prescribed Courant-number field plus
sedimentation - the step that precedes collisions in 1-D / 2-D / 3-D set-ups.  It is what moves
super-droplets between cells (and, across ranks, what makes them migrate), removes the ones that
precipitate or leave the column, and leaves the permutation unsorted for the collision step.

One call of `run()` is one `Displacement.__call__` of the reference
(PySDM/dynamics/displacement.py:100-153): `n_substeps` times { displacement of every dimension
from the Arakawa-C field (displacement_methods.py:14-129), sedimentation, position update,
precipitation flagging + removal, out-of-column flagging + removal, whole-cell carry into the cell
origin, periodic boundary, cell id }.  Routes: "fused" = `sdm_displacement_step`, all sub-steps in
one library call; "chain" = one ABI symbol per stage.
"""
import ctypes

import numpy as np

from . import abi
from .engine import FLOAT, INT
from .population import grid_strides
from .terminal_velocity import LAWS

SCHEMES = {"ImplicitInSpace": 0, "ExplicitInSpace": 1}
_EW_ADD, _EW_SUB, _EW_MUL, _EW_MOD = 0, 1, 2, 11


def substeps_for(courant_field, rtol):
    """number of sub-steps (a power of two) at which implicit and explicit Euler advection agree
    to `rtol` (Arabas et al. 2015, eqs 13-16; displacement.py:76-98): the relative difference is
    1 / (1 / max|dC| - 1) with dC the largest change of a Courant component along its own axis"""
    count = 1
    while True:
        worst = 0.0
        for axis, component in enumerate(courant_field):
            step = np.amax(np.abs(np.diff(component, axis=axis))) / count
            worst = max(worst, 0.0 if step == 0 else 1 / (1 / step - 1))
        if worst < rtol:
            return count
        count *= 2


class DisplacementRunner:  # pylint: disable=too-many-instance-attributes
    def __init__(self, population, *, dt, size, enable_sedimentation=False,
                 precipitation_counting_level_index=0, adaptive=True, rtol=1e-2,
                 scheme="ImplicitInSpace", route="fused", terminal_velocity="GunnKinzer1949"):
        if population.grid is None or population.cell_origin is None:
            raise ValueError("displacement needs a population with a grid, cell origins and "
                             "positions in cell")
        if len(population.grid) not in (1, 2, 3):
            raise NotImplementedError("1, 2 or 3 dimensions")
        self.population = population
        self.engine = eng = population.engine
        self.dt = float(dt)
        self.grid = population.grid
        self.n_dims = len(self.grid)
        self.dz = float(size[-1]) / self.grid[-1]
        self.enable_sedimentation = bool(enable_sedimentation)
        self.level = float(precipitation_counting_level_index)
        self.adaptive, self.rtol = adaptive, rtol
        self.scheme = SCHEMES[scheme]
        self.route = route
        self.n_substeps = 1
        self.precipitation_mass_in_last_step = 0.0
        n_sd = population.n_sd
        self.courant = [eng.full(tuple(g + (1 if axis == d else 0)
                                       for axis, g in enumerate(self.grid)), FLOAT, np.nan)
                        for d in range(self.n_dims)]
        self.displacement = eng.zeros((self.n_dims, n_sd), FLOAT)
        self.whole_cells = eng.zeros((self.n_dims, n_sd), INT)
        self.strides = eng.upload(grid_strides(self.grid))
        self.ctl = eng.zeros(8, INT)
        self.law = LAWS[terminal_velocity](eng) if enable_sedimentation else None
        # a sharded run (pysdm_amd.sharding.attach_displacement): this process moves the
        # super-droplets of its own cells, `sdm_displacement_step_sharded`
        self.shard = None
        self.shard_stats = {"moved": 0, "left": 0, "arrived": 0, "words": 0, "removed": 0,
                            "calls": 0}

    def set_courant(self, courant_field):
        """component d on the cell faces normal to d: grid shape with one more point along d"""
        for target, values in zip(self.courant, courant_field):
            values = np.asarray(values, dtype=float)
            if tuple(values.shape) != tuple(target.shape):
                raise ValueError(f"Courant component of shape {values.shape}, expected "
                                 f"{tuple(target.shape)}")
            self.engine.assign(target, self.engine.upload(values))
        self.n_substeps = substeps_for(courant_field, self.rtol) if self.adaptive else 1

    # ---- one call ------------------------------------------------------------------------------------
    def run(self):
        pop = self.population
        pop.refresh_bookkeeping()  # (after fused collision steps without read-back)
        pop.compact()
        if self.route == "fused":
            self._run_fused()
        else:
            self._run_chain()
        pop.touch_cells()
        return self.precipitation_mass_in_last_step

    __call__ = run

    def _run_fused(self):
        pop, eng = self.population, self.engine

        def address(array):
            return abi.c_ptr(array.data_ptr() if hasattr(array, "data_ptr")
                             else array.ctypes.data)

        cfg = abi.DispCfg()
        cfg.n_sd, cfg.n_dims, cfg.scheme = pop.n_sd, self.n_dims, self.scheme
        cfg.enable_sedimentation, cfg.n_substeps = int(self.enable_sedimentation), self.n_substeps
        cfg.grid = (abi.c_i64 * 3)(*self.grid, *([1] * (3 - self.n_dims)))
        cfg.strides = (abi.c_i64 * 3)(*[int(s) for s in grid_strides(self.grid)],
                                      *([0] * (3 - self.n_dims)))
        if self.enable_sedimentation:
            cfg.dt_over_dz = self.dt / self.n_substeps / self.dz
        cfg.level = self.level
        state = abi.DispState()
        for d in range(self.n_dims):
            state.courant[d] = address(self.courant[d]).value
        state.displacement = address(self.displacement)
        state.position_in_cell = address(pop.position_in_cell)
        state.cell_origin = address(pop.cell_origin)
        state.cell_id = address(pop.cell_id)
        if self.enable_sedimentation:
            state.fall_velocity = address(pop.fall_velocity(self.law))
        state.water_mass = address(pop.mass)
        state.multiplicity = address(pop.multiplicity)
        state.idx = address(pop.perm)
        eng.assign(self.ctl, eng.upload(np.asarray([pop.live, pop.live, 0, 1, 0, 0, 0, 0],
                                                   dtype=np.int64)))
        state.ctl = address(self.ctl)
        rainfall, survivors = ctypes.c_double(), ctypes.c_int64()
        if self.shard is None:
            eng.call("sdm_displacement_step", cfg, state, rainfall, survivors)
        else:
            shard = self.shard
            n_attr = int(pop.extensive.shape[0])
            counts, words = shard.displacement_buffers(
                pop.n_sd * (6 + 2 * self.n_dims + n_attr))
            if pop.cell_id_by_id is None:  # until now every id's cell_id entry was its own cell
                pop.cell_id_by_id = (pop.cell_id.clone() if hasattr(pop.cell_id, "clone")
                                     else pop.cell_id.copy())
            part = abi.DispShard()
            part.cell_owned, part.n_cell = address(shard.owned), pop.n_cell
            part.exchange = ctypes.cast(shard.callback, ctypes.c_void_p)
            part.exchange_user = None
            part.shard_rank, part.shard_world = shard.rank, shard.world
            part.xchg_counts, part.xchg_words = address(counts), address(words)
            part.word_capacity = int(words.shape[0])
            part.cell_id_by_id = address(pop.cell_id_by_id)
            part.role, part.role_ready = address(shard.role), int(shard.role_ready)
            part.multiplicity, part.attributes = address(pop.multiplicity), address(pop.extensive)
            part.n_attr = n_attr
            shard.error = None
            try:
                eng.call("sdm_displacement_step_sharded", cfg, state, part, rainfall, survivors)
            except RuntimeError as error:
                raise (shard.error or error) from error
            for key in ("moved", "left", "arrived", "words", "removed"):
                self.shard_stats[key] += int(getattr(part, "n_" + key))
            self.shard_stats["calls"] += 1
            shard.role_ready = True
            pop.touch_state()  # rows of arriving super-droplets: the collision step's mirror
        pop.live = pop.working = survivors.value
        self.precipitation_mass_in_last_step = rainfall.value

    def _run_chain(self):
        pop, eng = self.population, self.engine
        n_sd, dims = pop.n_sd, self.n_dims
        total = dims * n_sd
        rain = 0.0
        for _ in range(self.n_substeps):
            for d in range(dims):
                shape = (abi.c_i64 * 3)(*self.courant[d].shape, *([1] * (3 - dims)))
                eng.call("sdm_calculate_displacement", d, dims, self.scheme, self.displacement,
                         self.courant[d], shape, pop.cell_origin, pop.position_in_cell, n_sd,
                         float(self.n_substeps))
            if self.enable_sedimentation:
                vertical = self.displacement[dims - 1]
                dt_over_dz = self.dt / self.n_substeps / self.dz
                ew = lambda op, other, scalar: eng.call(  # noqa: E731
                    "sdm_elementwise_f64", op, vertical, vertical, other, scalar, n_sd)
                ew(_EW_MUL, None, 1 / dt_over_dz)
                ew(_EW_SUB, pop.fall_velocity(self.law), 0.0)
                ew(_EW_MUL, None, dt_over_dz)
            eng.call("sdm_elementwise_f64", _EW_ADD, pop.position_in_cell, pop.position_in_cell,
                     self.displacement, 0.0, total)
            if self.enable_sedimentation:
                rain += eng.scalar_out(
                    "sdm_flag_precipitated", ctypes.c_double, pop.cell_origin,
                    pop.position_in_cell, pop.mass, pop.multiplicity, pop.perm, pop.live, n_sd,
                    dims, pop.healthy, self.level, self.displacement)
                pop.compact()
            eng.call("sdm_flag_out_of_column", pop.cell_origin, pop.position_in_cell, pop.perm,
                     pop.live, n_sd, dims, pop.healthy, float(self.grid[-1]))
            pop.compact()
            eng.call("sdm_floor_to_i64", self.whole_cells, pop.position_in_cell, total)
            eng.call("sdm_elementwise_i64", _EW_ADD, pop.cell_origin, pop.cell_origin,
                     self.whole_cells, 0, total)
            eng.call("sdm_subtract_i64", pop.position_in_cell, self.whole_cells, total)
            for d in range(dims):  # periodic boundary
                row = pop.cell_origin[d]
                eng.call("sdm_elementwise_i64", _EW_MOD, row, row, None, int(self.grid[d]), n_sd)
            eng.call("sdm_cell_id", pop.cell_id, pop.cell_origin, self.strides, dims, n_sd)
        self.precipitation_mass_in_last_step = rain
