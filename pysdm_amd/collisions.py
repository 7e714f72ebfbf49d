"""`CollisionRunner`: time steps of the SDM collision dynamic on a `Population`.

Two routes through the same library:

* "fused" (default): one `sdm_collision_step` / `sdm_collision_run` call per time step / run.  The
  control state (live and working length, sortedness, health) stays in the population's `ctl`
  block on the device between calls; random numbers are generated inside the kernels from the
  NumPy-PCG64 stream at the positions this object tracks (`offset`, `offset_breakup`).
* "chain": stage by stage through the fine-grained symbols (pysdm_amd.chain).

Both produce the state the reference's `Collision.__call__` (PySDM/dynamics/collisions/
collision.py:174-194) produces for the same seed; per-cell diagnostics (`collision_rate`,
`collision_rate_deficit`, `coalescence_rate`, `breakup_rate`, `breakup_rate_deficit`,
`stats_n_substep`, `stats_dt_min`) are arrays of this object, named as in the reference.
"""
import warnings

import numpy as np

from . import abi
from .engine import FLOAT, INT
from .physics import constants as const
from .terminal_velocity import LAWS

READ_BACK, FRESH_CTL, MIRROR_VALID = 1, 2, 4
_DEVICE_ERRORS = {1: "a cell is larger than the per-cell kernel's capacity",
                  2: "the grid barrier of the compaction kernel timed out (is the GPU shared "
                     "with another process?)"}


class CollisionRunner:  # pylint: disable=too-many-instance-attributes
    def __init__(self, population, setup, *, dt, dv, route="fused",
                 terminal_velocity="GunnKinzer1949", constants=None, read_back=True):
        if population.n_sd < 2:
            raise ValueError("No one to collide with!")
        if route not in ("fused", "chain"):
            raise ValueError(route)
        if population.n_cell > 1 and setup.croupier == "global" and setup.adaptive:
            # the one configuration in which this package does NOT reproduce the reference, by
            # design ("Several cells, global croupier, adaptive")
            warnings.warn(
                "several cells + global croupier + adaptive sub-stepping: once a working length "
                "is cut, the reference's counting sort duplicates ids beyond the cut "
                "(collisions_methods.py:587-631) and the outcome depends on the order of "
                "execution; this backend keeps every id exactly once instead - results agree with "
                "the reference only up to the first cut", stacklevel=2)
        self.population = population
        self.engine = eng = population.engine
        self.setup = setup
        self.dt, self.dv = float(dt), float(dv)
        self.route = route
        self.read_back = read_back
        self.constants = constants or const.namespace()
        self.dt_range = setup.clamped_dt_range(self.dt)
        n_cell = population.n_cell
        self.collision_rate = eng.zeros(n_cell, INT)
        self.collision_rate_deficit = eng.zeros(n_cell, INT)
        self.coalescence_rate = eng.zeros(n_cell, INT)
        self.breakup_rate = eng.zeros(n_cell, INT) if setup.breakup else None
        self.breakup_rate_deficit = eng.zeros(n_cell, INT) if setup.breakup else None
        self.stats_n_substep = eng.full(n_cell, INT, 0 if setup.adaptive else setup.substeps)
        self.stats_dt_min = eng.full(n_cell, FLOAT, np.nan)
        self.dt_left = eng.full(n_cell, FLOAT, np.nan)
        self.sub_steps_done = 0
        self.pairs_done = 0
        self.steps_done = 0
        self.descriptor = setup.descriptor(self.constants)
        self.gamma_hook = None  # chain route only, see pysdm_amd.chain
        self.shard = None       # set by pysdm_amd.sharding.attach
        self.counts_global_pairs = False
        self._law_name = terminal_velocity
        self._law = None
        self._chain = None
        self._cfg = self._state = None
        self._result = abi.StepResult()
        self.offset = self.offset_breakup = 0

    # ---- parts of the set-up ------------------------------------------------------------------------
    @property
    def law(self):
        """terminal-velocity law feeding the "fall velocity" column (built on first use)"""
        if self._law is None:
            self._law = LAWS[self._law_name](self.engine)
        return self._law

    def step_cfg(self):
        """the set-up as `sdm_step_cfg`"""
        if self._cfg is not None:
            return self._cfg
        pop, setup, k, desc = self.population, self.setup, self.constants, self.descriptor
        cfg = abi.StepCfg()
        cfg.n_sd, cfg.n_cell, cfg.n_attr = pop.n_sd, pop.n_cell, len(pop.rows)
        cfg.dt, cfg.dv = self.dt, self.dv
        cfg.dt_min, cfg.dt_max = self.dt_range
        cfg.adaptive, cfg.substeps = int(setup.adaptive), int(setup.substeps)
        cfg.croupier_local = int(setup.croupier == "local")
        cfg.optimized_random = int(setup.optimized_random)
        cfg.enable_breakup = int(setup.breakup)
        cfg.handle_all_breakups = int(setup.handle_all_breakups)
        cfg.kernel, cfg.ec, cfg.frag = desc.get("kernel", 0), desc.get("ec", 0), desc.get(
            "frag", 0)
        cfg.mass_attr = pop.rows["signed water mass"]
        for name, width in (("kernel_param", 2), ("ec_param", 2), ("frag_param", 2),
                            ("berry_params", 13), ("kernel_berry_params", 13)):
            setattr(cfg, name, (abi.c_f64 * width)(*desc.get(name, (0.0,) * width)))
        cfg.eb_const = desc.get("eb_const", 0.0)
        cfg.frag_vmin, cfg.frag_nfmax = desc.get("frag_vmin", 0.0), desc.get("frag_nfmax", -1.0)
        cfg.rho_w, cfg.sgm_w = k.rho_w, k.sgm_w
        cfg.straub_consts = (abi.c_f64 * 6)(k.CM, k.STRAUB_E_D1, k.STRAUB_MU2, k.VEDDER_1987_A,
                                            k.VEDDER_1987_b, k.PI)
        cfg.berry_unit = desc.get("berry_unit", 1.0)
        cfg.kernel_berry_unit = desc.get("kernel_berry_unit", 1.0)
        cfg.max_multiplicity = int(setup.max_multiplicity)
        cfg.rng_state_inc = (abi.c_u64 * 4)(*abi.pcg64_state_inc(setup.seed))
        if desc["needs_gk"]:
            if self._law_name != "GunnKinzer1949":
                raise NotImplementedError("the fused step evaluates fall velocities from the "
                                          "Gunn-Kinzer table; use route='chain' for other laws")
            cfg.gk_table_len = self.law.length
            cfg.gk_factor = float(self.law.factor)
        self._cfg = cfg
        return cfg

    def _step_state(self):
        pop = self.population
        state = self._state
        if state is None:
            def address(array):
                return None if array is None else abi.c_ptr(
                    array.data_ptr() if hasattr(array, "data_ptr") else array.ctypes.data)

            state = abi.StepState()
            for name, array in (
                    ("multiplicity", pop.multiplicity), ("attributes", pop.extensive),
                    ("cell_id", pop.cell_id), ("cell_idx", pop.cell_order),
                    ("cell_start", pop.cell_start), ("dt_left", self.dt_left),
                    ("stats_dt_min", self.stats_dt_min),
                    ("stats_n_substep", self.stats_n_substep),
                    ("collision_rate", self.collision_rate),
                    ("collision_rate_deficit", self.collision_rate_deficit),
                    ("coalescence_rate", self.coalescence_rate),
                    ("breakup_rate", self.breakup_rate),
                    ("breakup_rate_deficit", self.breakup_rate_deficit),
                    ("ctl", pop.ctl), ("nm", pop.mirror)):
                setattr(state, name, address(array))
            if self.descriptor["needs_gk"]:
                state.gk_a, state.gk_b = address(self.law.a), address(self.law.b)
            state.known_valid = -1
            if self.shard is not None:
                self.shard.fill(state, address)
            self._address = address
            self._state = state
        # (sharded run with a sharded displacement step: every id's own cell, see sdm_hip.h)
        state.cell_id_by_id = self._address(getattr(pop, "cell_id_by_id", None))
        state.idx = self._address(pop.perm)
        state.tmp_idx = self._address(pop.perm_spare)
        state.rng_offset = self.offset
        state.rng_offset_breakup = self.offset_breakup
        return state

    # ---- running ------------------------------------------------------------------------------------
    def run(self, n_steps=1):
        if n_steps <= 0:
            return
        if self.route == "chain":
            self._run_chain(n_steps)
        else:
            self._run_fused(n_steps)
        self.steps_done += n_steps

    __call__ = run

    def _run_chain(self, n_steps):
        if self._chain is None:
            from .chain import ChainedCollision  # pylint: disable=import-outside-toplevel

            self._chain = ChainedCollision(self)
            self._chain.draws.offset = self.offset
            self._chain.draws.offset_breakup = self.offset_breakup
        for _ in range(n_steps):
            self._chain.time_step()
        self.offset = self._chain.draws.offset
        self.offset_breakup = self._chain.draws.offset_breakup

    def _run_fused(self, n_steps):
        pop, eng = self.population, self.engine
        flags = READ_BACK if self.read_back else 0
        if pop.host_dirty and pop.bookkeeping_stale:
            # steps without read-back left live / working / ordered behind the device's control
            # block; the host view is about to be uploaded: bring it up to date first
            self.sync()
        if pop.host_dirty:
            pop.compact()  # the fused compaction looks for flagged slots only
            eng.assign(pop.ctl, eng.upload(np.asarray(
                [pop.live, pop.working, int(pop.ordered), 1, 0, 0, 0, 0], dtype=np.int64)))
            flags |= FRESH_CTL
            if pop.mirror_version is not None and pop.mirror_version == pop.state_version:
                flags |= MIRROR_VALID  # only the permutation / cell ids changed since
            self._state = None
        state, result = self._step_state(), self._result
        try:
            if n_steps == 1:
                eng.call("sdm_collision_step", self.step_cfg(), state, result, flags)
            else:
                eng.call("sdm_collision_run", self.step_cfg(), state, result, flags,
                         int(n_steps))
        except RuntimeError as failure:
            if self.shard is not None and self.shard.error is not None:
                raise RuntimeError("exchange between the processes failed") from self.shard.error
            raise failure
        if result.idx_swapped:
            pop.swap_buffers()
        self.offset, self.offset_breakup = result.rng_offset, result.rng_offset_breakup
        self.sub_steps_done += result.n_substeps
        if result.n_pairs >= 0:
            self.pairs_done += result.n_pairs
        pop.state_version += 1
        pop.mirror_version = pop.state_version
        pop.host_dirty = False
        pop.bookkeeping_stale = not self.read_back
        if self.read_back:
            self._adopt(list(result.ctl))

    def _adopt(self, words):
        pop = self.population
        pop.live, pop.working, pop.ordered = int(words[0]), int(words[0]), bool(words[2])
        pop.bookkeeping_stale = False
        error = int(words[7]) & 0xff
        if error != 0:
            raise RuntimeError("libsdm_hip: device-side failure in the fused collision step: "
                               + _DEVICE_ERRORS.get(error, f"code {error}"))
        if int(words[7]) & 0x100:
            # a cell's stats_dt_min became equal to dt_min in some sub-step: the reference's
            # condition (collision.py:276-277; NaN entries, the initial state, silence it)
            self.population.ctl[7] = 0
            smallest = self.engine.scalar_out("sdm_reduce_f64", abi.c_f64, 0, self.stats_dt_min,
                                              self.population.n_cell)
            if smallest == self.dt_range[0]:
                warnings.warn("adaptive time-step reached dt_min")
        if words[4] > 0 and self.setup.warn_overflows:
            warnings.warn("overflow")
            self.population.ctl[4] = 0

    def sync(self):
        """brings the host-side bookkeeping up to date after `read_back=False` steps"""
        self._adopt(self.engine.download(self.population.ctl))

    # ---- results ------------------------------------------------------------------------------------
    def snapshot(self):
        """population + diagnostics as host arrays"""
        down = self.engine.download
        snap = self.population.snapshot()
        for name in ("collision_rate", "collision_rate_deficit", "coalescence_rate",
                     "stats_n_substep", "stats_dt_min", "breakup_rate", "breakup_rate_deficit"):
            array = getattr(self, name)
            if array is not None:
                snap[name] = down(array)
        return snap
