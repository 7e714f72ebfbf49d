"""Host side of the fused per-time-step route: packs the dynamic's configuration and the device
state into the C structs of include/sdm_hip.h and calls `sdm_collision_step` once per time step.
The control words (valid / working length, sorted, healthy) stay on the device between calls.
"""
import ctypes
import warnings

import torch

from .. import _lib
from .._lib import StepCfg, StepResult, StepState, c_ptr, check
from . import state_access
from .hip import _Context, pcg64_state_inc


def _p(tensor):
    return None if tensor is None else ctypes.c_void_p(tensor.data_ptr())


class FusedStep:  # pylint: disable=too-many-instance-attributes
    def __init__(self, backend, dynamic, parts):
        self.backend = backend
        self.dynamic = dynamic
        self.particulator = part = dynamic.particulator
        attrs = part.attributes
        const = backend.formulae.constants
        self.read_back = True
        self.total_pairs = 0
        self.total_substeps = 0

        cfg = StepCfg()
        cfg.n_sd, cfg.n_cell = part.n_sd, part.mesh.n_cell
        storage = attrs.get_extensive_attribute_storage()
        cfg.n_attr = storage.shape[0]
        cfg.dt, cfg.dv = part.dt, part.mesh.dv
        cfg.dt_min, cfg.dt_max = dynamic.dt_coal_range
        cfg.adaptive = int(dynamic.adaptive)
        cfg.substeps = int(dynamic.substeps)
        cfg.croupier_local = int(dynamic.croupier == "local")
        cfg.optimized_random = int(dynamic.optimized_random)
        cfg.enable_breakup = int(dynamic.enable_breakup)
        cfg.handle_all_breakups = int(backend.formulae.handle_all_breakups)
        cfg.kernel = parts.get("kernel", 0)
        cfg.ec = parts.get("ec", 0)
        cfg.frag = parts.get("frag", 0)
        keys = list(attrs.get_extensive_attribute_keys())
        cfg.mass_attr = keys.index("signed water mass")
        cfg.kernel_param = (ctypes.c_double * 2)(*parts.get("kernel_param", (0.0, 0.0)))
        cfg.ec_param = (ctypes.c_double * 2)(*parts.get("ec_param", (0.0, 0.0)))
        cfg.eb_const = parts.get("eb_const", 0.0)
        cfg.frag_param = (ctypes.c_double * 2)(*parts.get("frag_param", (0.0, 0.0)))
        cfg.frag_vmin = parts.get("frag_vmin", 0.0)
        cfg.frag_nfmax = parts.get("frag_nfmax", -1.0)
        cfg.rho_w, cfg.sgm_w = const.rho_w, const.sgm_w
        cfg.straub_consts = backend.straub_consts()
        cfg.berry_params = (ctypes.c_double * 13)(*parts.get("berry_params", (0.0,) * 13))
        cfg.berry_unit = parts.get("berry_unit", 1.0)
        cfg.kernel_berry_params = (ctypes.c_double * 13)(
            *parts.get("kernel_berry_params", (0.0,) * 13))
        cfg.kernel_berry_unit = parts.get("kernel_berry_unit", 1.0)
        cfg.max_multiplicity = int(dynamic.max_multiplicity)
        cfg.rng_state_inc = (ctypes.c_uint64 * 4)(*pcg64_state_inc(backend.formulae.seed))
        self.gk = None
        if parts.get("needs_gk", False):
            self.gk = state_access.attribute_object(
                attrs, "relative fall velocity").approximation
            cfg.gk_table_len = self.gk.a.data.numel()
            cfg.gk_factor = float(self.gk.factor)
        self.cfg = cfg

        view = state_access.view(attrs)
        self.idx = view["idx"]
        self.tmp_idx = view["caretaker"].tmp_idx
        self.ctl = torch.zeros(8, dtype=torch.int64, device=self.idx.data.device)
        # per-droplet mirror records, up to 32 B each (see sdm_step_state.nm)
        self.nm = torch.empty(4 * part.n_sd, dtype=torch.int64, device=self.idx.data.device)
        self._ctl_initialised = False
        self._stamps = None
        self._mirror_built = False
        self._state_cache = None
        self.result = StepResult()

    def _push_host_state(self):
        # the fused step's compaction only looks for flagged positions: hand it a state without
        # unflagged zero multiplicities (what the reference's super_droplet_count asserts anyway)
        self.particulator.attributes.sanitize()
        view = state_access.view(self.particulator.attributes)
        host = torch.tensor(
            [view["valid_n_sd"], len(view["idx"]), int(view["sorted"]),
             int(bool(view["healthy"])), 0, 0, 0, 0], dtype=torch.int64,
        )
        self.ctl.copy_(host)
        self._ctl_initialised = True

    def _state(self):
        """the C view of the device state (rebuilt per call: cheap next to a time step only if
        kept short, hence the cached constant part)"""
        dyn, attrs = self.dynamic, self.particulator.attributes
        state = self._state_cache
        if state is None:
            state = StepState()
            state.multiplicity = _p(attrs["multiplicity"].data)
            state.attributes = _p(attrs.get_extensive_attribute_storage().data)
            state.cell_id = _p(attrs["cell id"].data)
            state.cell_idx = _p(attrs.cell_idx.data)
            state.cell_start = _p(state_access.view(attrs)["cell_start"].data)
            state.dt_left = _p(dyn.dt_left.data)
            state.stats_dt_min = _p(dyn.stats_dt_min.data)
            state.stats_n_substep = _p(dyn.stats_n_substep.data)
            state.collision_rate = _p(dyn.collision_rate.data)
            state.collision_rate_deficit = _p(dyn.collision_rate_deficit.data)
            state.coalescence_rate = _p(dyn.coalescence_rate.data)
            if dyn.enable_breakup:
                state.breakup_rate = _p(dyn.breakup_rate.data)
                state.breakup_rate_deficit = _p(dyn.breakup_rate_deficit.data)
            if self.gk is not None:
                state.gk_a, state.gk_b = _p(self.gk.a.data), _p(self.gk.b.data)
            state.ctl = _p(self.ctl)
            state.nm = _p(self.nm)
            state.known_valid = -1
            self._state_cache = state
        state.idx = _p(self.idx.data)
        state.tmp_idx = _p(self.tmp_idx.data)
        state.rng_offset = dyn.rnd_opt_coll.rnd.offset
        if dyn.enable_breakup:
            state.rng_offset_breakup = dyn.rnd_opt_proc.rnd.offset
        return state

    def __call__(self, n_steps=1):
        dyn = self.dynamic
        flags = int(self.read_back)
        # anything else that touched multiplicities / attributes since the last fused call
        # (method-by-method calls, uploads) invalidates the device-side bookkeeping
        stamps = self._timestamps()
        mirror_valid = self._stamps is not None and self._stamps[0] == stamps[0]
        if self._stamps != stamps:
            self._ctl_initialised = False
            self._state_cache = None
        if not self._ctl_initialised:
            self._push_host_state()
            flags |= 2
            if mirror_valid and self._mirror_built:  # only the permutation / cell ids changed
                flags |= 4
        self._mirror_built = True
        state = self._state()
        ctx = _Context.get()
        if n_steps == 1:
            check(ctx.lib.sdm_collision_step(ctx.handle, ctypes.byref(self.cfg),
                                             ctypes.byref(state), ctypes.byref(self.result),
                                             flags))
        else:
            check(ctx.lib.sdm_collision_run(ctx.handle, ctypes.byref(self.cfg),
                                            ctypes.byref(state), ctypes.byref(self.result),
                                            flags, ctypes.c_int64(n_steps)))
        res = self.result
        if res.idx_swapped:
            self.idx.data, self.tmp_idx.data = self.tmp_idx.data, self.idx.data
        dyn.rnd_opt_coll.rnd.offset = res.rng_offset
        if dyn.enable_breakup:
            dyn.rnd_opt_proc.rnd.offset = res.rng_offset_breakup
            dyn.rnd_opt_frag.rnd.offset = res.rng_offset_breakup
        self.total_substeps += res.n_substeps
        if res.n_pairs >= 0:
            self.total_pairs += res.n_pairs
        if self.read_back:
            self._commit(list(self.result.ctl))  # read back by the library at the end of the call
        state_access.mark_collision_outputs_updated(self.particulator.attributes)
        self._stamps = self._timestamps()

    def _timestamps(self):
        """(stamps of what the mirror holds, stamp of the cell ids)"""
        attrs = self.particulator.attributes
        names = ["multiplicity"] + list(attrs.get_extensive_attribute_keys())
        stamp = lambda name: state_access.attribute_object(attrs, name).timestamp  # noqa: E731
        return tuple(stamp(name) for name in names), stamp("cell id")

    def _commit(self, words):
        state_access.commit(self.particulator.attributes, valid_n_sd=int(words[0]),
                            sorted_flag=bool(words[2]))
        if words[7] != 0:
            raise RuntimeError(
                "libsdm_hip: device-side failure in the fused collision step "
                + {1: "(cell larger than the per-cell kernel's capacity)",
                   2: "(grid barrier of the compaction kernel timed out)"}.get(int(words[7]), "")
            )
        if words[4] > 0 and self.dynamic.warn_overflows:
            warnings.warn("overflow")
            self.ctl[4] = 0

    def sync(self):
        """bring the host-side bookkeeping up to date (needed after read_back=False steps)"""
        self._commit(self.ctl.cpu().numpy())
