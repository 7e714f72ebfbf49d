"""`HIP`: the MI355X backend object -- PySDM's backend interface for the collision path with every
method executed by hand-written HIP kernels (libsdm_hip.so, C ABI in include/sdm_hip.h).

Drop-in contract (reference: PySDM/backends/numba.py:18-67,
PySDM/backends/thrust_rtc.py): an *instance* with `.formulae`, `.Storage`, `.Random`,
`.default_croupier`, constructed as `HIP(formulae=None, double_precision=True)`.  torch is used for
device memory and streams only.  There is no CPU fallback: without a GPU or without the library
construction fails.
"""
import ctypes
import warnings

import numpy as np
import torch

from .. import _lib
from .._lib import c_f64, c_i64, c_int, c_ptr, check
from ..formulae import Formulae
from . import storage_base as sb
from .impl_common import BackendMethods, RandomCommon, advection_scheme_id

_TORCH_DTYPE = {np.float64: torch.float64, np.int64: torch.int64, np.bool_: torch.bool}


def _ptr(tensor):
    if tensor is None:
        return c_ptr(0)
    if not tensor.is_contiguous():
        raise ValueError("non-contiguous device array passed to libsdm_hip")
    return c_ptr(tensor.data_ptr())


class _Context:
    """one sdm_ctx per process and device; follows torch's current stream"""

    _instances = {}

    def __init__(self, device_index):
        self.lib = _lib.load()
        self.handle = c_ptr()
        check(self.lib.sdm_ctx_create(ctypes.byref(self.handle), c_int(device_index)))
        self.device = torch.device("cuda", device_index)
        self._stream = None

    @classmethod
    def get(cls, device_index=None):
        if not torch.cuda.is_available():
            raise RuntimeError("pysdm_amd.backends.HIP needs a GPU (torch.cuda.is_available() is "
                               "False); there is no CPU fallback")
        if device_index is None:
            device_index = torch.cuda.current_device()
        if device_index not in cls._instances:
            cls._instances[device_index] = _Context(device_index)
        ctx = cls._instances[device_index]
        stream = torch.cuda.current_stream(ctx.device).cuda_stream
        if stream != ctx._stream:
            check(ctx.lib.sdm_ctx_set_stream(ctx.handle, c_ptr(stream)))
            ctx._stream = stream
        return ctx


def _call(name, *args):
    ctx = _Context.get()
    check(getattr(ctx.lib, name)(ctx.handle, *args))


class Storage(sb.StorageBase):
    _IS_BACKEND_STORAGE = True

    @classmethod
    def _alloc(cls, shape, dtype):
        return torch.empty(shape, dtype=_TORCH_DTYPE[dtype], device=_Context.get().device)

    @classmethod
    def _upload_raw(cls, array):
        return torch.from_numpy(np.ascontiguousarray(array)).to(_Context.get().device)

    @staticmethod
    def _download_raw(raw):
        return raw.detach().cpu().numpy()

    @staticmethod
    def _assign_raw(raw, key, value):
        raw[key] = value

    @staticmethod
    def _scalar(raw_element):
        return raw_element.item()

    def _ew(self, op, a, b=None, scalar=0.0):
        out = self.data
        n = out.numel()
        if op == sb.EW_FLOOR and self.dtype is Storage.INT and a.dtype == torch.float64:
            _call("sdm_floor_to_i64", _ptr(out), _ptr(a), c_i64(n))
        elif (op == sb.EW_SUB and self.dtype is Storage.FLOAT and b is not None
              and b.dtype == torch.int64):
            assert a.data_ptr() == out.data_ptr()
            _call("sdm_subtract_i64", _ptr(out), _ptr(b), c_i64(n))
        elif self.dtype is Storage.FLOAT:
            _call("sdm_elementwise_f64", c_int(op), _ptr(out), _ptr(a), _ptr(b),
                  c_f64(float(scalar)), c_i64(n))
        elif self.dtype is Storage.INT:
            _call("sdm_elementwise_i64", c_int(op), _ptr(out), _ptr(a), _ptr(b),
                  c_i64(int(scalar)), c_i64(n))
        else:
            raise NotImplementedError("arithmetic on bool storage")

    def _reduce(self, kind):
        if self.dtype is not Storage.FLOAT:
            host = self.to_ndarray()
            return host.min() if kind == 0 else host.max()
        result = c_f64()
        _call("sdm_reduce_f64", c_int(kind), _ptr(self.data), c_i64(self.data.numel()),
              ctypes.byref(result))
        return result.value


def pcg64_state_inc(seed):
    """{state_hi, state_lo, inc_hi, inc_lo} of numpy.random.PCG64(seed) -- NumPy defines the
    stream (impl_numba/random.py:16); the device reproduces it with jump-ahead"""
    state = np.random.PCG64(seed).state["state"]
    mask = (1 << 64) - 1
    return (state["state"] >> 64, state["state"] & mask, state["inc"] >> 64, state["inc"] & mask)


class Random(RandomCommon):  # pylint: disable=too-few-public-methods
    """device-side NumPy-PCG64 stream: each call continues where the previous one stopped"""

    def __init__(self, size, seed):
        super().__init__(size, seed)
        self.state_inc = (ctypes.c_uint64 * 4)(*pcg64_state_inc(seed))
        self.offset = 0

    def __call__(self, storage):
        n = storage.data.numel()
        _call("sdm_pcg64_uniform", _ptr(storage.data), c_i64(n), self.state_inc,
              ctypes.c_uint64(self.offset))
        self.offset += n


_PAIR_OPS = {"sum": 0, "max": 1, "min": 2, "distance": 3, "multiply": 4}


def _flag(is_first_in_pair):
    return _ptr(is_first_in_pair.indicator.data)


def _is_int(storage):
    return c_int(1 if storage.dtype is Storage.INT else 0)


class HIP(BackendMethods):  # pylint: disable=too-many-public-methods
    Storage = Storage
    Random = Random
    default_croupier = "local"

    def __init__(self, formulae=None, double_precision=True, device=None):
        if not double_precision:
            raise NotImplementedError("the HIP backend computes in float64 only")
        self.formulae = formulae or Formulae()
        _Context.get(device)
        super().__init__()

    @staticmethod
    def synchronize():
        ctx = _Context.get()
        check(ctx.lib.sdm_ctx_synchronize(ctx.handle))

    # ---- index methods (index_methods.py) ----------------------------------------------------
    @staticmethod
    def identity_index(idx):
        _call("sdm_identity_index", _ptr(idx), c_i64(idx.numel()))

    @staticmethod
    def shuffle_global(idx, length, u01):
        _call("sdm_shuffle_global", _ptr(idx), c_i64(int(length)), _ptr(u01))

    @staticmethod
    def shuffle_local(idx, u01, cell_start):
        _call("sdm_shuffle_local", _ptr(idx), _ptr(u01), _ptr(cell_start),
              c_i64(cell_start.numel() - 1))

    @staticmethod
    def sort_by_key(idx, attr):
        _call("sdm_sort_by_key", _ptr(idx.data), _ptr(attr.data), c_i64(attr.data.numel()))

    @staticmethod
    def remove_zero_n_or_flagged(multiplicity, idx, length):
        new_length = c_i64()
        _call("sdm_remove_zero_n_or_flagged", _ptr(multiplicity), _ptr(idx), c_i64(int(length)),
              c_i64(idx.numel()), ctypes.byref(new_length))
        return new_length.value

    @staticmethod
    def make_cell_caretaker(idx_shape, idx_dtype, cell_start_len, scheme="default"):
        tmp_idx = Storage.empty(idx_shape, idx_dtype)

        def caretaker(cell_id, cell_idx, cell_start, idx):
            _call("sdm_counting_sort_by_cell_id", _ptr(tmp_idx.data), _ptr(idx.data),
                  _ptr(cell_id.data), _ptr(cell_idx.data), c_i64(len(idx)),
                  _ptr(cell_start.data), c_i64(cell_start_len - 1))
            idx.data, tmp_idx.data = tmp_idx.data, idx.data

        caretaker.tmp_idx = tmp_idx
        return caretaker

    @staticmethod
    def cell_id(cell_id, cell_origin, strides):
        flat = strides.data.reshape(-1).contiguous()
        _call("sdm_cell_id", _ptr(cell_id.data), _ptr(cell_origin.data), _ptr(flat),
              c_i64(flat.numel()), c_i64(cell_id.data.numel()))

    # ---- pair methods (pair_methods.py) ------------------------------------------------------
    @staticmethod
    def find_pairs(cell_start, is_first_in_pair, cell_id, cell_idx, idx):
        _call("sdm_find_pairs", _ptr(cell_start.data), _flag(is_first_in_pair),
              _ptr(cell_id.data), _ptr(cell_idx.data), _ptr(idx.data), c_i64(len(idx)))

    @staticmethod
    def sort_within_pair_by_attr(idx, is_first_in_pair, attr):
        _call("sdm_sort_within_pair_by_attr", _ptr(idx.data), c_i64(len(idx)),
              _flag(is_first_in_pair), _ptr(attr.data), _is_int(attr))

    @staticmethod
    def _pair_op(name, data_out, data_in, is_first_in_pair, idx):
        _call("sdm_pair_op", c_int(_PAIR_OPS[name]), _ptr(data_out.data),
              c_i64(data_out.data.numel()), _ptr(data_in.data), _is_int(data_in),
              _flag(is_first_in_pair), _ptr(idx.data), c_i64(len(idx)))

    def sum_pair(self, data_out, data_in, is_first_in_pair, idx):
        self._pair_op("sum", data_out, data_in, is_first_in_pair, idx)

    def max_pair(self, data_out, data_in, is_first_in_pair, idx):
        self._pair_op("max", data_out, data_in, is_first_in_pair, idx)

    def min_pair(self, data_out, data_in, is_first_in_pair, idx):
        self._pair_op("min", data_out, data_in, is_first_in_pair, idx)

    def distance_pair(self, data_out, data_in, is_first_in_pair, idx):
        self._pair_op("distance", data_out, data_in, is_first_in_pair, idx)

    def multiply_pair(self, data_out, data_in, is_first_in_pair, idx):
        self._pair_op("multiply", data_out, data_in, is_first_in_pair, idx)

    @staticmethod
    def sort_pair(data_out, data_in, is_first_in_pair, idx):
        _call("sdm_sort_pair", _ptr(data_out.data), c_i64(data_out.data.numel()),
              _ptr(data_in.data), _flag(is_first_in_pair), _ptr(idx.data), c_i64(len(idx)))

    # ---- collisions methods (collisions_methods.py) -------------------------------------------
    @staticmethod
    def normalize(prob, cell_id, cell_idx, cell_start, norm_factor, timestep, dv):
        _call("sdm_normalize", _ptr(prob.data), c_i64(prob.data.numel()), _ptr(cell_id.data),
              _ptr(cell_idx.data), _ptr(cell_start.data), _ptr(norm_factor.data),
              c_i64(cell_start.data.numel() - 1), c_f64(timestep), c_f64(dv))

    @staticmethod
    def scale_prob_for_adaptive_sdm_gamma(*, prob, multiplicity, cell_id, dt_left, dt, dt_range,
                                          is_first_in_pair, stats_n_substep, stats_dt_min):
        _call("sdm_scale_prob_for_adaptive_sdm_gamma", _ptr(prob.data),
              _ptr(multiplicity.idx.data), c_i64(len(multiplicity)), _ptr(multiplicity.data),
              _ptr(cell_id.data), _ptr(dt_left.data), c_i64(dt_left.data.numel()), c_f64(dt),
              c_f64(dt_range[0]), c_f64(dt_range[1]), _flag(is_first_in_pair),
              _ptr(stats_n_substep.data), _ptr(stats_dt_min.data))

    @staticmethod
    def compute_gamma(*, prob, rand, multiplicity, cell_id, collision_rate_deficit,
                      collision_rate, is_first_in_pair, out):
        _call("sdm_compute_gamma", _ptr(prob.data), _ptr(rand.data), _ptr(multiplicity.idx.data),
              c_i64(len(multiplicity)), _ptr(multiplicity.data), _ptr(cell_id.data),
              _ptr(collision_rate_deficit.data), _ptr(collision_rate.data),
              _flag(is_first_in_pair), _ptr(out.data))

    @staticmethod
    def adaptive_sdm_end(dt_left, cell_start):
        end = c_i64()
        _call("sdm_adaptive_sdm_end", _ptr(dt_left.data), c_i64(len(dt_left)),
              _ptr(cell_start.data), ctypes.byref(end))
        return end.value

    @staticmethod
    def collision_coalescence(*, multiplicity, idx, attributes, gamma, healthy, cell_id,
                              coalescence_rate, is_first_in_pair):
        _call("sdm_collision_coalescence", _ptr(multiplicity.data), _ptr(idx.data),
              c_i64(len(idx)), _ptr(attributes.data), c_i64(attributes.shape[0]),
              c_i64(attributes.shape[1]), _ptr(gamma.data), _ptr(healthy.data),
              _ptr(cell_id.data), _ptr(coalescence_rate.data), _flag(is_first_in_pair))

    def collision_coalescence_breakup(self, *, multiplicity, idx, attributes, gamma, rand, Ec, Eb,
                                      fragment_mass, healthy, cell_id, coalescence_rate,
                                      breakup_rate, breakup_rate_deficit, is_first_in_pair,
                                      warn_overflows, particle_mass, max_multiplicity):
        n_overflow = None
        if warn_overflows:
            n_overflow = torch.zeros(1, dtype=torch.int64, device=multiplicity.data.device)
        _call("sdm_collision_coalescence_breakup", _ptr(multiplicity.data), _ptr(idx.data),
              c_i64(len(idx)), _ptr(attributes.data), c_i64(attributes.shape[0]),
              c_i64(attributes.shape[1]), _ptr(gamma.data), _ptr(rand.data), _ptr(Ec.data),
              _ptr(Eb.data), _ptr(fragment_mass.data), _ptr(healthy.data), _ptr(cell_id.data),
              _ptr(coalescence_rate.data), _ptr(breakup_rate.data),
              _ptr(breakup_rate_deficit.data), _flag(is_first_in_pair),
              c_i64(int(max_multiplicity)), _ptr(particle_mass.data),
              c_int(int(self.formulae.handle_all_breakups)), _ptr(n_overflow))
        if warn_overflows and int(n_overflow.item()) > 0:
            warnings.warn("overflow")

    @staticmethod
    def linear_collection_efficiency(*, params, output, radii, is_first_in_pair, unit):
        par = (c_f64 * 13)(*[float(p) for p in params])
        _call("sdm_linear_collection_efficiency", par, _ptr(output.data),
              c_i64(output.data.numel()), _ptr(radii.data), _flag(is_first_in_pair),
              _ptr(radii.idx.data), c_i64(len(is_first_in_pair)), c_f64(unit))

    # ---- derived attributes -------------------------------------------------------------------
    @staticmethod
    def interpolation(*, output, radius, factor, b, c):
        _call("sdm_interpolation", _ptr(output.data), _ptr(radius.data),
              c_i64(radius.data.numel()), c_f64(factor), _ptr(b.data), _ptr(c.data),
              c_i64(b.data.numel()))

    def volume_of_water_mass(self, volume, mass):
        _call("sdm_volume_of_water_mass", _ptr(volume.data), _ptr(mass.data),
              c_i64(volume.data.numel()), c_f64(self.formulae.constants.rho_w))

    def mass_of_water_volume(self, mass, volume):
        _call("sdm_mass_of_water_volume", _ptr(mass.data), _ptr(volume.data),
              c_i64(volume.data.numel()), c_f64(self.formulae.constants.rho_w))

    # ---- fragmentation (fragmentation_methods.py) -----------------------------------------------
    @staticmethod
    def exp_fragmentation(*, n_fragment, scale, frag_volume, x_plus_y, rand, vmin, nfmax,
                          tol=1e-5):
        _call("sdm_exp_fragmentation", _ptr(n_fragment.data), c_f64(scale),
              _ptr(frag_volume.data), _ptr(x_plus_y.data), _ptr(rand.data),
              c_i64(frag_volume.data.numel()), c_f64(vmin),
              c_f64(-1.0 if nfmax is None else nfmax), c_f64(tol))

    def gauss_fragmentation(self, *, n_fragment, mu, sigma, frag_volume, x_plus_y, rand, vmin,
                            nfmax):
        const = self.formulae.constants
        _call("sdm_gauss_fragmentation", _ptr(n_fragment.data), c_f64(mu), c_f64(sigma),
              _ptr(frag_volume.data), _ptr(x_plus_y.data), _ptr(rand.data),
              c_i64(frag_volume.data.numel()), c_f64(vmin),
              c_f64(-1.0 if nfmax is None else nfmax),
              (c_f64 * 2)(const.VEDDER_1987_A, const.VEDDER_1987_b))

    @staticmethod
    def feingold1988_fragmentation(*, n_fragment, scale, frag_volume, x_plus_y, rand, fragtol,
                                   vmin, nfmax):
        _call("sdm_feingold1988_fragmentation", _ptr(n_fragment.data), c_f64(scale),
              _ptr(frag_volume.data), _ptr(x_plus_y.data), _ptr(rand.data),
              c_i64(frag_volume.data.numel()), c_f64(fragtol), c_f64(vmin),
              c_f64(-1.0 if nfmax is None else nfmax))

    @staticmethod
    def slams_fragmentation(n_fragment, frag_volume, x_plus_y, probs, rand, vmin, nfmax):
        _call("sdm_slams_fragmentation", _ptr(n_fragment.data), _ptr(frag_volume.data),
              _ptr(x_plus_y.data), _ptr(probs.data), _ptr(rand.data),
              c_i64(frag_volume.data.numel()), c_f64(vmin),
              c_f64(-1.0 if nfmax is None else nfmax))

    def ll82_fragmentation(self, *, n_fragment, CKE, W, W2, St, ds, dl, dcoal, frag_volume,
                           x_plus_y, rand, vmin, nfmax, Rf, Rs, Rd, tol=1e-8):
        const = self.formulae.constants
        _call("sdm_ll82_fragmentation", _ptr(n_fragment.data), _ptr(CKE.data), _ptr(W.data),
              _ptr(W2.data), _ptr(St.data), _ptr(ds.data), _ptr(dl.data), _ptr(dcoal.data),
              _ptr(frag_volume.data), _ptr(x_plus_y.data), _ptr(rand.data),
              c_i64(frag_volume.data.numel()), c_f64(vmin),
              c_f64(-1.0 if nfmax is None else nfmax), _ptr(Rf.data), _ptr(Rs.data),
              _ptr(Rd.data), c_f64(tol),
              (c_f64 * 4)(const.CM, const.PI, const.VEDDER_1987_A, const.VEDDER_1987_b))

    @staticmethod
    def ll82_coalescence_check(*, Ec, dl):
        _call("sdm_ll82_coalescence_check", _ptr(Ec.data), _ptr(dl.data), c_i64(Ec.data.numel()))

    def straub_consts(self):
        const = self.formulae.constants
        return (c_f64 * 6)(const.CM, const.STRAUB_E_D1, const.STRAUB_MU2, const.VEDDER_1987_A,
                           const.VEDDER_1987_b, const.PI)

    def straub_fragmentation(self, *, n_fragment, CW, gam, ds, frag_volume, v_max, x_plus_y, rand,
                             vmin, nfmax, Nr1, Nr2, Nr3, Nr4, Nrt, d34):
        _call("sdm_straub_fragmentation", _ptr(n_fragment.data), _ptr(CW.data), _ptr(gam.data),
              _ptr(ds.data), _ptr(frag_volume.data), _ptr(v_max.data), _ptr(x_plus_y.data),
              _ptr(rand.data), c_i64(frag_volume.data.numel()), c_f64(vmin),
              c_f64(-1.0 if nfmax is None else nfmax), _ptr(Nr1.data), _ptr(Nr2.data),
              _ptr(Nr3.data), _ptr(Nr4.data), _ptr(Nrt.data), _ptr(d34.data),
              self.straub_consts())

    # ---- terminal velocities other than the Gunn-Kinzer table (terminal_velocity_methods.py) ------
    def terminal_velocity(self, *, values, radius):
        # (raw device arrays, as the reference passes `.data`)
        const = self.formulae.constants
        _call("sdm_terminal_velocity", _ptr(values), _ptr(radius), c_i64(values.numel()),
              (c_f64 * 5)(const.ROGERS_YAU_TERM_VEL_SMALL_K, const.ROGERS_YAU_TERM_VEL_MEDIUM_K,
                          const.ROGERS_YAU_TERM_VEL_LARGE_K,
                          const.ROGERS_YAU_TERM_VEL_SMALL_R_LIMIT,
                          const.ROGERS_YAU_TERM_VEL_MEDIUM_R_LIMIT))

    @staticmethod
    def power_series(*, values, radius, num_terms, prefactors, powers):
        n_terms = int(num_terms)
        _call("sdm_power_series", _ptr(values), _ptr(radius), c_i64(values.numel()),
              c_int(n_terms), (c_f64 * n_terms)(*[float(v) for v in prefactors]),
              (c_f64 * n_terms)(*[float(v) for v in powers]))

    # ---- displacement (displacement_methods.py) ------------------------------------------------
    def calculate_displacement(self, *, dim, displacement, courant, cell_origin, position_in_cell,
                               n_substeps):
        n_dims = len(courant.shape)
        if n_dims not in (1, 2, 3):
            raise NotImplementedError()
        _call("sdm_calculate_displacement", c_int(dim), c_int(n_dims),
              c_int(advection_scheme_id(self.formulae)), _ptr(displacement.data),
              _ptr(courant.data), (c_i64 * 3)(*courant.shape, *([1] * (3 - n_dims))),
              _ptr(cell_origin.data), _ptr(position_in_cell.data),
              c_i64(displacement.shape[1]), c_f64(n_substeps))

    @staticmethod
    def flag_precipitated(*, cell_origin, position_in_cell, water_mass, multiplicity, idx, length,
                          healthy, precipitation_counting_level_index, displacement) -> float:
        result = c_f64()
        _call("sdm_flag_precipitated", _ptr(cell_origin.data), _ptr(position_in_cell.data),
              _ptr(water_mass.data), _ptr(multiplicity.data), _ptr(idx.data), c_i64(int(length)),
              c_i64(idx.data.numel()), c_int(cell_origin.shape[0]), _ptr(healthy.data),
              c_f64(precipitation_counting_level_index), _ptr(displacement.data),
              ctypes.byref(result))
        return result.value

    @staticmethod
    def flag_out_of_column(cell_origin, position_in_cell, idx, length, healthy,
                           domain_top_level_index):
        _call("sdm_flag_out_of_column", _ptr(cell_origin.data), _ptr(position_in_cell.data),
              _ptr(idx.data), c_i64(int(length)), c_i64(idx.data.numel()),
              c_int(cell_origin.shape[0]), _ptr(healthy.data), c_f64(domain_top_level_index))

    # ---- moments (moments_methods.py) ---------------------------------------------------------
    @staticmethod
    def moments(*, moment_0, moments, multiplicity, attr_data, cell_id, idx, length, ranks, min_x,
                max_x, x_attr, weighting_attribute, weighting_rank, skip_division_by_m0):
        _call("sdm_moments", _ptr(moment_0.data), _ptr(moments.data), _ptr(multiplicity.data),
              _ptr(attr_data.data), _ptr(cell_id.data), _ptr(idx.data), c_i64(int(length)),
              _ptr(ranks.data), c_i64(ranks.data.numel()), c_i64(moment_0.data.numel()),
              c_f64(min_x), c_f64(max_x), _ptr(x_attr.data), _ptr(weighting_attribute.data),
              c_f64(weighting_rank), c_int(int(skip_division_by_m0)))

    @staticmethod
    def spectrum_moments(*, moment_0, moments, multiplicity, attr_data, cell_id, idx, length, rank,
                         x_bins, x_attr, weighting_attribute, weighting_rank):
        assert moments.shape[0] == x_bins.shape[0] - 1
        assert moment_0.shape == moments.shape
        _call("sdm_spectrum_moments", _ptr(moment_0.data), _ptr(moments.data),
              _ptr(multiplicity.data), _ptr(attr_data.data), _ptr(cell_id.data), _ptr(idx.data),
              c_i64(int(length)), c_f64(rank), _ptr(x_bins.data), c_i64(moments.shape[0]),
              c_i64(moments.shape[1]), _ptr(x_attr.data), _ptr(weighting_attribute.data),
              c_f64(weighting_rank))

    # ---- the fused displacement step (sdm_displacement_step) ---------------------------------------
    def displacement_step(self, dynamic):
        """one `Displacement.__call__` in one library call; returns the precipitated mass"""
        from . import state_access  # pylint: disable=import-outside-toplevel

        part = dynamic.particulator
        attrs = part.attributes
        view = state_access.view(attrs)
        n_valid = attrs.super_droplet_count  # asserts a healthy state, as flag_precipitated does
        mesh = part.mesh
        n_dims = len(mesh.grid)
        cfg = _lib.DispCfg()
        cfg.n_sd, cfg.n_dims = part.n_sd, n_dims
        cfg.scheme = advection_scheme_id(self.formulae)
        cfg.enable_sedimentation = int(dynamic.enable_sedimentation)
        cfg.n_substeps = int(dynamic._n_substeps)  # pylint: disable=protected-access
        cfg.grid = (c_i64 * 3)(*[int(g) for g in mesh.grid], *([1] * (3 - n_dims)))
        strides = np.asarray(mesh.strides).ravel()
        cfg.strides = (c_i64 * 3)(*[int(v) for v in strides], *([0] * (3 - n_dims)))
        if dynamic.enable_sedimentation:
            cfg.dt_over_dz = part.dt / cfg.n_substeps / mesh.dz
        cfg.level = float(dynamic.precipitation_counting_level_index)
        state = _lib.DispState()
        for dim in range(n_dims):
            state.courant[dim] = dynamic.courant[dim].data.data_ptr()
        state.displacement = dynamic.displacement.data.data_ptr()
        state.position_in_cell = attrs["position in cell"].data.data_ptr()
        state.cell_origin = attrs["cell origin"].data.data_ptr()
        state.cell_id = attrs["cell id"].data.data_ptr()
        if dynamic.enable_sedimentation:
            state.fall_velocity = attrs["relative fall velocity"].data.data_ptr()
        state.water_mass = attrs["water mass"].data.data_ptr()
        state.multiplicity = attrs["multiplicity"].data.data_ptr()
        idx = view["idx"]
        state.idx = idx.data.data_ptr()
        ctl = torch.tensor([n_valid, n_valid, 0, 1, 0, 0, 0, 0], dtype=torch.int64).to(
            idx.data.device)
        state.ctl = ctl.data_ptr()
        rainfall, survivors = c_f64(), c_i64()
        _call("sdm_displacement_step", ctypes.byref(cfg), ctypes.byref(state),
              ctypes.byref(rainfall), ctypes.byref(survivors))
        state_access.commit(attrs, valid_n_sd=survivors.value, sorted_flag=False)
        return rainfall.value

    # ---- the fused per-time-step route ----------------------------------------------------------
    def make_collision_step(self, dynamic, parts):
        from .hip_fused import FusedStep  # pylint: disable=import-outside-toplevel

        return FusedStep(self, dynamic, parts)

    @staticmethod
    def collision_step(fused_step, n_steps=1):
        fused_step(n_steps)
