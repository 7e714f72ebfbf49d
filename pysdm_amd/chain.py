"""The fine-grained route of a collision time step: one ABI symbol per stage.

This is the sequence a front-end drives when it talks to the backend method by method (the route
PySDM's own `Collision` takes through the plug-in): draw - permute - pair - probabilities -
(efficiencies, fragments) - adaptive scaling - gamma - update - compaction, with the adaptive
sub-step loop around it.  It exists next to the fused route (`sdm_collision_step`, one call per
time step) to exercise every fine-grained kernel of the library in situ and to cover parts that
have no fused descriptor.  Stage order and the lazy counting sort follow
PySDM/dynamics/collisions/collision.py:174-290 and PySDM/impl/particle_attributes.py:47-110; the
parts (kernels, efficiencies, fragmentation functions) are *pair programs* (pysdm_amd.recipe)
run by the small interpreter below.
"""
import ctypes
import math
import warnings

import numpy as np

from . import abi
from .engine import BOOL, FLOAT

PAIR_OPS = {"sum": 0, "max": 1, "min": 2, "distance": 3, "multiply": 4}
EW = {"add": 0, "sub": 1, "mul": 2, "div": 3, "pow": 4, "divnz": 5, "exp": 7, "fill": 9}


class Draws:
    """the reference's stream layout (dynamics/impl/random_generator_optimizer*.py): per draw
    `n_sd (+ shift)` doubles for the permutation, then `n_sd // 2` for gamma, from one PCG64
    stream; breakup adds two generators with the same seed, i.e. one further stream read twice.
    With `optimized_random` a time step draws once and its sub-steps slide a window over it."""

    def __init__(self, engine, n_sd, seed, *, optimized, dt, dt_min, breakup):
        self.engine = engine
        self.n_sd, self.n_pairs = n_sd, n_sd // 2
        self.optimized = optimized
        self.shift_room = math.ceil(dt / dt_min) if optimized else 0
        self.state_inc = abi.pcg64_state_inc(seed)
        self.permutation = engine.empty(n_sd + self.shift_room, FLOAT)
        self.gamma = engine.empty(self.n_pairs, FLOAT)
        self.process = self.fragment = None
        if breakup:
            self.process = engine.empty(self.n_pairs, FLOAT)
            self.fragment = engine.empty(self.n_pairs, FLOAT)
        self.offset = self.offset_breakup = 0
        self.sub_step = 0

    def _fill(self, array, offset):
        n = self.engine.size(array)
        self.engine.call("sdm_pcg64_uniform", array, n, self.state_inc, offset)
        return offset + n

    def next(self):
        if not self.optimized or self.sub_step == 0:
            self.offset = self._fill(self.permutation, self.offset)
            self.offset = self._fill(self.gamma, self.offset)
            if self.process is not None:
                self._fill(self.process, self.offset_breakup)
                self.offset_breakup = self._fill(self.fragment, self.offset_breakup)
        first = self.sub_step if self.optimized else 0
        self.sub_step += 1
        return self.permutation[first: self.n_sd + first]

    def end_of_time_step(self):
        self.sub_step = 0


class ChainedCollision:  # pylint: disable=too-many-instance-attributes
    """executes time steps of `runner` (a CollisionRunner) stage by stage"""

    def __init__(self, runner):
        self.runner = runner
        pop, setup, eng = runner.population, runner.setup, runner.engine
        self.n_pairs = pop.n_sd // 2
        self.flag = eng.empty(pop.n_sd, BOOL)
        self.kernel_value = self._pair_array()
        self.prob = self._pair_array()          # probability, then gamma, in place
        self.norm_factor = eng.empty(pop.n_cell, FLOAT)
        self.registers = {}
        if setup.breakup:
            self.ec, self.eb = self._pair_array(), self._pair_array()
            self.n_fragment, self.fragment_mass = self._pair_array(), self._pair_array()
            self.overflow = eng.zeros(1, np.int64)
        self.draws = Draws(eng, pop.n_sd, setup.seed, optimized=setup.optimized_random,
                           dt=runner.dt, dt_min=runner.dt_range[0], breakup=setup.breakup)

    def _pair_array(self):
        # a fresh Storage of the reference is NaN-filled (storage.py:121-134): programs that
        # scale a scratch array before writing it see the same values here
        return self.runner.engine.full(self.n_pairs, FLOAT, np.nan)

    # ---- the interpreter ---------------------------------------------------------------------------
    def execute(self, program, **bound):
        eng, pop = self.runner.engine, self.runner.population
        n = self.n_pairs
        regs = dict(bound)

        def reg(name):
            if name not in regs:
                if name not in self.registers:
                    self.registers[name] = self._pair_array()
                regs[name] = self.registers[name]
            return regs[name]

        def elementwise(code, dst, other):
            if isinstance(other, str):
                eng.call("sdm_elementwise_f64", code, reg(dst), reg(dst), reg(other), 0.0, n)
            else:
                eng.call("sdm_elementwise_f64", code, reg(dst), reg(dst), None, float(other), n)

        for ins in program:
            op = ins[0]
            if op == "pair":
                column = pop.column(ins[3], self.runner.law)
                eng.call("sdm_pair_op", PAIR_OPS[ins[1]], reg(ins[2]), n, column,
                         int(ins[3] == "multiplicity"), self.flag, pop.perm, pop.working)
            elif op in ("mul", "add", "div", "sub", "divnz", "pow"):
                elementwise(EW[op], ins[1], ins[2])
            elif op == "exp":
                eng.call("sdm_elementwise_f64", EW["exp"], reg(ins[1]), reg(ins[1]), None, 0.0, n)
            elif op == "fill":
                eng.call("sdm_elementwise_f64", EW["fill"], reg(ins[1]), None, None,
                         float(ins[2]), n)
            elif op == "copy":
                eng.assign(reg(ins[1]), reg(ins[2]))
            elif op == "lce":
                eng.call("sdm_linear_collection_efficiency", [float(p) for p in ins[2]],
                         reg(ins[1]), n, pop.radius(), self.flag, pop.perm, pop.working,
                         float(ins[3]))
            elif op == "volume_to_mass":
                eng.call("sdm_mass_of_water_volume", reg(ins[1]), reg(ins[1]), n, pop.rho_w)
            elif op == "call":
                args = [n if a == "#pairs" else reg(a) if isinstance(a, str) else a
                        for a in ins[2:]]
                eng.call(ins[1], *args)
            else:
                raise ValueError(f"unknown pair-program instruction {ins!r}")

    # ---- one sub-step -------------------------------------------------------------------------------
    def sub_step(self):  # pylint: disable=too-many-locals
        run, pop, setup, eng = self.runner, self.runner.population, self.runner.setup, \
            self.runner.engine
        k = run.constants
        u01 = self.draws.next()
        # (the lazy sort exchanges the permutation buffers: ask for cell_start first)
        if setup.croupier == "local":
            cell_start = pop.sorted_cell_start()
            eng.call("sdm_shuffle_local", pop.perm, u01, cell_start, pop.n_cell)
        else:
            eng.call("sdm_shuffle_global", pop.perm, pop.working, u01)
            pop.ordered = False
        cell_start = pop.sorted_cell_start()
        eng.call("sdm_find_pairs", cell_start, self.flag, pop.cell_id, pop.cell_order, pop.perm,
                 pop.working)
        eng.call("sdm_sort_within_pair_by_attr", pop.perm, pop.working, self.flag,
                 pop.multiplicity, 1)
        run.pairs_done += pop.working // 2
        # probability of collision, eq. (20) of Shima et al. 2009
        self.execute(setup.kernel.program(k), out=self.kernel_value)
        self.execute([("pair", "max", "out", "multiplicity"), ("mul", "out", "kernel")],
                     out=self.prob, kernel=self.kernel_value)
        eng.call("sdm_normalize", self.prob, self.n_pairs, pop.cell_id, pop.cell_order,
                 cell_start, self.norm_factor, pop.n_cell, run.dt, run.dv)
        if setup.breakup:
            self.execute(setup.coalescence_efficiency.program(k), out=self.ec)
            self.execute(setup.breakup_efficiency.program(k), out=self.eb)
            self.execute(setup.fragmentation.program(k), nf=self.n_fragment,
                         fm=self.fragment_mass, u01=self.draws.fragment)
        if run.gamma_hook is not None:
            # test hook: stands in for the adaptive scaling + gamma stage (the scenarios of the
            # reference's unit tests force gamma); gets (chain, probability array, rand array)
            run.gamma_hook(self, self.prob, self.draws.gamma)
        elif setup.adaptive:
            eng.call("sdm_scale_prob_for_adaptive_sdm_gamma", self.prob, pop.perm, pop.working,
                     pop.multiplicity, pop.cell_id, run.dt_left, pop.n_cell, run.dt,
                     run.dt_range[0], run.dt_range[1], self.flag, run.stats_n_substep,
                     run.stats_dt_min)
            smallest = eng.scalar_out("sdm_reduce_f64", ctypes.c_double, 0, run.stats_dt_min,
                                      pop.n_cell)
            if smallest == run.dt_range[0]:
                warnings.warn("adaptive time-step reached dt_min")
        else:
            eng.call("sdm_elementwise_f64", EW["div"], self.prob, self.prob, None,
                     float(setup.substeps), self.n_pairs)
        if run.gamma_hook is None:
            self.compute_gamma()
        if setup.breakup:
            eng.fill(self.overflow, 0)
            eng.call("sdm_collision_coalescence_breakup", pop.multiplicity, pop.perm,
                     pop.working, pop.extensive, len(pop.rows), pop.n_sd, self.prob,
                     self.draws.process, self.ec, self.eb, self.fragment_mass, pop.healthy,
                     pop.cell_id, run.coalescence_rate, run.breakup_rate,
                     run.breakup_rate_deficit, self.flag, int(setup.max_multiplicity), pop.mass,
                     int(setup.handle_all_breakups),
                     self.overflow if setup.warn_overflows else None)
            if setup.warn_overflows and int(eng.download(self.overflow)[0]) > 0:
                warnings.warn("overflow")
        else:
            eng.call("sdm_collision_coalescence", pop.multiplicity, pop.perm, pop.working,
                     pop.extensive, len(pop.rows), pop.n_sd, self.prob, pop.healthy, pop.cell_id,
                     run.coalescence_rate, self.flag)
        pop.compact()
        pop.touch_state()
        run.sub_steps_done += 1

    def compute_gamma(self):
        run, pop = self.runner, self.runner.population
        run.engine.call("sdm_compute_gamma", self.prob, self.draws.gamma, pop.perm, pop.working,
                        pop.multiplicity, pop.cell_id, run.collision_rate_deficit,
                        run.collision_rate, self.flag, self.prob)

    # ---- one time step ------------------------------------------------------------------------------
    def time_step(self):
        run, pop, setup, eng = self.runner, self.runner.population, self.runner.setup, \
            self.runner.engine
        if not setup.adaptive:
            for _ in range(setup.substeps):
                self.sub_step()
        else:
            eng.fill(run.dt_left, run.dt)
            while pop.working != 0:
                eng.call("sdm_sort_by_key", pop.cell_order, run.dt_left, pop.n_cell)
                self.sub_step()
                cell_start = pop.sorted_cell_start()
                pop.working = eng.scalar_out("sdm_adaptive_sdm_end", ctypes.c_int64, run.dt_left,
                                             pop.n_cell, cell_start)
            pop.working = pop.live
            eng.call("sdm_identity_index", pop.cell_order, pop.n_cell)
            pop.sort_by_cell()
        self.draws.end_of_time_step()
