"""`Population`: the super-droplet state of one rank, laid out as the library consumes it.

Structure-of-arrays, every column an array of the engine (device memory for the HIP engine):

    perm, perm_spare   int64[n_sd]      permutation of live super-droplets + its double buffer
    multiplicity       int64[n_sd]
    extensive          float64[A, n_sd] extensive attributes, one row each ("signed water mass", ..)
    cell_id            int64[n_sd]
    cell_order         int64[n_cell]    order of the cells in the sorted permutation
    cell_start         int64[n_cell+1]  first position of each cell in the sorted permutation
    cell_origin, position_in_cell       (n_dims, n_sd), only with a grid
    ctl                int64[8]         control block of the fused entry points
    mirror             int64[4 n_sd]    per-droplet records kept by the fused step (HIP)
    healthy            int64[1]         "no zero multiplicity among the live" flag of the chain route

The same columns are what the reference's `ParticleAttributes` holds behind its attribute
objects (PySDM/impl/particle_attributes.py:13-46; layout: particle_attributes_factory.py:42-44),
which is what makes a Population interchangeable with it at the backend boundary.  Host-side
bookkeeping: `live` (number of valid super-droplets), `working` (length the current sub-step
works on), `ordered` (permutation sorted by cell), `state_version` / `cells_version` (bumped when
multiplicities-attributes / cell ids change; derived columns and the mirror are cached by them).
"""
import ctypes

import numpy as np

from .engine import FLOAT, INT
from .physics import constants as const

MASS_ROW = "signed water mass"


def to_integer_multiplicities(values):
    """real-valued multiplicities -> int64 (half-even rounding); NaN marks an unused slot and
    becomes 0.  Refuses a discretisation that empties a droplet or changes the total number of
    real droplets by more than 1 % (the checks of PySDM/initialisation/
    discretise_multiplicities.py:8-32)"""
    values = np.asarray(values)
    if values.dtype.kind != "f":
        return values.astype(np.int64)
    unused = np.isnan(values)
    counts = np.rint(np.where(unused, 0.0, values)).astype(np.int64)
    if unused.all():
        return counts
    if (counts[~unused] <= 0).any():
        raise ValueError("int-casting resulted in multiplicity of zero "
                         f"(min(y_float)={np.nanmin(values)})")
    drift = 100 * abs(1 - np.nansum(values) / np.sum(counts.astype(float)))
    if drift > 1:
        raise ValueError(f"{drift}% error in total real-droplet number due to casting "
                         "multiplicities to ints")
    return counts


def grid_strides(grid):
    """C-order strides (in cells) of a grid: cell id = sum_d origin[d] * strides[d]"""
    dims = [int(g) for g in grid]
    return np.asarray([int(np.prod(dims[d + 1:])) for d in range(len(dims))], dtype=np.int64)


def locate(positions, grid):
    """(cell id, cell origin, position in cell) of positions given in grid coordinates, shape
    (n_dims, n_sd) - what PySDM/impl/mesh.py:62-87 `cellular_attributes` returns"""
    positions = np.asarray(positions, dtype=float)
    origin = positions.astype(np.int64)
    within = positions - np.floor(positions)
    cell_id = (grid_strides(grid).reshape(-1, 1) * origin).sum(axis=0).astype(np.int64)
    return cell_id, origin, within


class Population:  # pylint: disable=too-many-instance-attributes
    def __init__(self, engine, *, multiplicity, mass=None, volume=None, cell_id=None, n_cell=1,
                 more_extensive=None, grid=None, cell_origin=None, position_in_cell=None,
                 rho_w=const.rho_w):
        if (mass is None) == (volume is None):
            raise ValueError("give either `mass` or `volume`")
        self.engine = engine
        self.rho_w = rho_w
        multiplicity = to_integer_multiplicities(multiplicity)
        n_sd = self.n_sd = int(multiplicity.shape[0])
        if mass is None:
            mass = rho_w * np.asarray(volume, dtype=float)  # liquid_spheres.py:22-23
        rows = {MASS_ROW: np.asarray(mass, dtype=float)}
        rows.update(more_extensive or {})
        self.rows = {name: row for row, name in enumerate(rows)}
        self.grid = None if grid is None else tuple(int(g) for g in grid)
        if self.grid is not None:
            n_cell = int(np.prod(self.grid))
        self.n_cell = int(n_cell)
        if cell_id is None:
            cell_id = np.zeros(n_sd, dtype=np.int64)
        cell_id = np.asarray(cell_id, dtype=np.int64)
        if cell_id.shape != (n_sd,) or (n_sd and (cell_id.min() < 0
                                                   or cell_id.max() >= self.n_cell)):
            raise ValueError("cell ids must lie in [0, n_cell)")

        up = engine.upload
        self.perm = up(np.arange(n_sd, dtype=np.int64))
        self.perm_spare = up(np.arange(n_sd, dtype=np.int64))
        self.multiplicity = up(multiplicity)
        self.extensive = up(np.stack([np.asarray(v, dtype=float) for v in rows.values()]))
        self.cell_id = up(cell_id)
        self.cell_order = up(np.arange(self.n_cell, dtype=np.int64))
        self.cell_start = engine.zeros(self.n_cell + 1, INT)
        self.healthy = engine.full(1, INT, 1)
        self.ctl = engine.zeros(8, INT)
        self.mirror = engine.empty(4 * max(n_sd, 1), INT)
        self.cell_origin = self.position_in_cell = None
        if cell_origin is not None:
            self.cell_origin = up(np.asarray(cell_origin, dtype=np.int64))
            self.position_in_cell = up(np.asarray(position_in_cell, dtype=float))

        self.live = n_sd
        self.working = n_sd
        self.ordered = False
        self.state_version = 0
        self.cells_version = 0
        self._derived = {}
        # the fused entry points keep {live, working, ordered, healthy} in `ctl` on the device
        # between calls; anything that changes them (or the columns) from the host side sets
        # `host_dirty`, and the next fused call starts from the host's view again
        self.host_dirty = True
        self.bookkeeping_stale = False  # see refresh_bookkeeping
        self.cell_id_by_id = None  # sharded runs with a sharded displacement step (displacement.py)
        self.mirror_version = None   # state_version the mirror records were built from
        if (multiplicity == 0).any():  # unused slots: compact them away before the first step
            self.compact(assume_unhealthy=True)

    @classmethod
    def adopt(cls, engine, *, perm, perm_spare, multiplicity, extensive, rows, cell_id,
              cell_order, cell_start, live, ordered, healthy=None, rho_w=const.rho_w):
        """a Population over columns that already live in the engine's memory and belong to
        someone else (e.g. PySDM's ParticleAttributes, see pysdm_amd.pysdm_plugin); nothing is
        copied"""
        self = cls.__new__(cls)
        self.engine, self.rho_w = engine, rho_w
        self.n_sd = int(multiplicity.shape[0])
        self.n_cell = int(cell_order.shape[0])
        self.grid = None
        self.rows = dict(rows)
        self.perm, self.perm_spare = perm, perm_spare
        self.multiplicity, self.extensive = multiplicity, extensive
        self.cell_id, self.cell_order, self.cell_start = cell_id, cell_order, cell_start
        self.healthy = healthy if healthy is not None else engine.full(1, INT, 1)
        self.ctl = engine.zeros(8, INT)
        self.mirror = engine.empty(4 * max(self.n_sd, 1), INT)
        self.cell_origin = self.position_in_cell = None
        self.live = self.working = int(live)
        self.ordered = bool(ordered)
        self.state_version = self.cells_version = 0
        self._derived = {}
        self.host_dirty = True
        self.bookkeeping_stale = False
        self.cell_id_by_id = None
        self.mirror_version = None
        return self

    # ---- views ----------------------------------------------------------------------------------
    @property
    def mass(self):
        return self.extensive[self.rows[MASS_ROW]]

    def touch_state(self):
        """multiplicities / extensive attributes changed"""
        self.state_version += 1
        self.host_dirty = True

    def touch_cells(self):
        """cell ids changed: the permutation is no longer sorted by cell"""
        self.cells_version += 1
        self.ordered = False
        self.host_dirty = True

    def refresh_bookkeeping(self):
        """fused steps without read-back leave `live` / `working` / `ordered` behind the control
        block on the device; whoever is about to use the host's view (a compaction, a
        displacement step, an upload of the control block) calls this first"""
        if self.bookkeeping_stale:
            words = self.engine.download(self.ctl)
            self.live = self.working = int(words[0])
            self.ordered = bool(words[2])
            self.bookkeeping_stale = False

    def swap_buffers(self):
        self.perm, self.perm_spare = self.perm_spare, self.perm

    # ---- chain-route primitives (each one ABI symbol) ---------------------------------------------
    def compact(self, assume_unhealthy=False):
        """drops super-droplets with zero multiplicity or a flagged slot from the permutation
        (`sanitize`, particle_attributes.py:67-73)"""
        eng = self.engine
        self.refresh_bookkeeping()
        if not assume_unhealthy and int(eng.download(self.healthy)[0]) != 0:
            return
        self.live = eng.scalar_out("sdm_remove_zero_n_or_flagged", ctypes.c_int64,
                                   self.multiplicity, self.perm, self.live, self.n_sd)
        self.working = self.live
        eng.fill(self.healthy, 1)
        self.ordered = False
        self.host_dirty = True

    def sort_by_cell(self):
        """stable counting sort of the working part of the permutation by cell
        (cell caretaker, collisions_methods.py:587-631)"""
        self.engine.call("sdm_counting_sort_by_cell_id", self.perm_spare, self.perm, self.cell_id,
                         self.cell_order, self.working, self.cell_start, self.n_cell)
        self.swap_buffers()
        self.ordered = True
        self.host_dirty = True

    def sorted_cell_start(self):
        """`cell_start`, sorting first if needed (the lazy property, particle_attributes.py:51-55)"""
        if not self.ordered:
            self.sort_by_cell()
        return self.cell_start

    # ---- derived columns (pure functions of the water mass; cached per state version) ------------
    def _cached(self, name, build):
        entry = self._derived.get(name)
        if entry is None or entry[0] != self.state_version:
            array = entry[1] if entry is not None else self.engine.empty(self.n_sd, FLOAT)
            build(array)
            self._derived[name] = (self.state_version, array)
        return self._derived[name][1]

    def volume(self):
        return self._cached("volume", lambda out: self.engine.call(
            "sdm_volume_of_water_mass", out, self.mass, self.n_sd, self.rho_w))

    def _power_of_volume(self, out, exponent):
        eng = self.engine
        eng.call("sdm_elementwise_f64", 2, out, self.volume(), None, 1 / const.PI_4_3, self.n_sd)
        eng.call("sdm_elementwise_f64", 4, out, out, None, exponent, self.n_sd)

    def radius(self):
        """attributes/physics/radius.py:15-17"""
        return self._cached("radius", lambda out: self._power_of_volume(out, 1 / 3))

    def area(self):
        """attributes/physics/area.py"""
        def build(out):
            self._power_of_volume(out, 2 / 3)
            self.engine.call("sdm_elementwise_f64", 2, out, out, None, const.PI_4_3 * 3,
                             self.n_sd)
        return self._cached("area", build)

    def fall_velocity(self, law):
        """terminal velocity of every slot by `law` (a pysdm_amd.terminal_velocity object)"""
        return self._cached("fall velocity", lambda out: law.evaluate(self.engine, out,
                                                                      self.radius(), self.n_sd))

    def column(self, name, law=None):
        """per-droplet column by the names pair programs use"""
        if name == "water mass":
            return self.mass
        if name == "multiplicity":
            return self.multiplicity
        if name == "fall velocity":
            return self.fall_velocity(law)
        return {"volume": self.volume, "radius": self.radius, "area": self.area}[name]()

    # ---- host copies ------------------------------------------------------------------------------
    def snapshot(self):
        """host copies of the canonical state (the sorted `cell_start` included)"""
        down = self.engine.download
        return {
            "idx": down(self.perm),
            "length": np.asarray(self.live),
            "multiplicity": down(self.multiplicity),
            "attributes": down(self.extensive),
            "cell_start": down(self.sorted_cell_start()),
        }

    def live_ids(self):
        return self.engine.download(self.perm)[: self.live]
