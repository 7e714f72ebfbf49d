"""TEST INFRASTRUCTURE, NOT PRODUCT CODE: a CPU backend object over the C oracle.

`OracleBackend` exposes the reference's backend-method names (PySDM/backends/numba.py:18-33 mixins,
collision path only) on numpy-backed `Storage`s and forwards each to oracle/sdm_oracle.c (serial,
strict IEEE) -- or, for the Storage element-wise ops, to the very numpy calls the reference uses
(PySDM/backends/impl_numba/storage_impl.py).  It exists so that tests can (i) pin the oracle
against the goldens generated from the reference and (ii) run the package's backend-neutral host
logic (pysdm_amd.*) without a GPU.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (pysdm_amd.backends.HIP) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

from pysdm_amd.backends import storage_base as sb
from pysdm_amd.backends.impl_common import BackendMethods, RandomCommon, advection_scheme_id

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsdm_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "sdm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(_LIB_PATH):
        subprocess.check_call(
            ["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
             "-fvisibility=hidden", "-o", _LIB_PATH, src, "-lm"]
        )
    return _LIB_PATH


_lib = None


def lib():
    global _lib  # pylint: disable=global-statement
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_remove_zero_n_or_flagged.restype = ctypes.c_int64
        _lib.oracle_adaptive_sdm_end.restype = ctypes.c_int64
        _lib.oracle_collision_coalescence_breakup.restype = ctypes.c_int64
    return _lib


def _p(array):
    if array is None:
        return None
    assert array.flags["C_CONTIGUOUS"], "oracle needs contiguous arrays"
    return array.ctypes.data_as(ctypes.c_void_p)


_i64 = ctypes.c_int64
_f64 = ctypes.c_double
_int = ctypes.c_int


class Storage(sb.StorageBase):
    _IS_BACKEND_STORAGE = True

    @classmethod
    def _alloc(cls, shape, dtype):
        return np.empty(shape, dtype=dtype)

    @classmethod
    def _upload_raw(cls, array):
        return np.array(array, copy=True)

    @staticmethod
    def _download_raw(raw):
        return raw.copy()

    @staticmethod
    def _assign_raw(raw, key, value):
        raw[key] = value

    def _ew(self, op, a, b=None, scalar=0.0):
        # the numpy expressions of PySDM/backends/impl_numba/storage_impl.py
        out = self.data
        y = b if b is not None else scalar
        if op == sb.EW_ADD:
            out[:] = a + y
        elif op == sb.EW_SUB:
            out[:] = a - y
        elif op == sb.EW_MUL:
            out[:] = a * y
        elif op == sb.EW_DIV:
            out[:] = a / y
        elif op == sb.EW_POW:
            out[:] = np.sign(a) * np.power(np.abs(a), scalar)
        elif op == sb.EW_DIV_IF_NOT_ZERO:
            mask = b != 0.0
            out[mask] = a[mask] / b[mask]
        elif op == sb.EW_FLOOR:
            out[:] = np.floor(a)
        elif op == sb.EW_EXP:
            out[:] = np.exp(a)
        elif op == sb.EW_ABS:
            out[:] = np.abs(a)
        elif op == sb.EW_FILL:
            out[:] = scalar
        elif op == sb.EW_ADD_MUL:
            out[:] = a + scalar * b
        elif op == sb.EW_MOD:
            out[:] = a % y
        else:
            raise NotImplementedError(op)

    def _reduce(self, kind):
        return np.amin(self.data) if kind == 0 else np.amax(self.data)


class Random(RandomCommon):  # PySDM/backends/impl_numba/random.py:13-19
    """PCG64 through the C restatement (pinned against numpy's own generator in the tests)"""

    def __init__(self, size, seed):
        super().__init__(size, seed)
        state = np.random.PCG64(seed).state["state"]
        mask = (1 << 64) - 1
        self.state = (ctypes.c_uint64 * 4)(
            state["state"] >> 64, state["state"] & mask, state["inc"] >> 64, state["inc"] & mask
        )

    def __call__(self, storage):
        flat = np.empty(int(np.prod(storage.shape)), dtype=np.float64)
        lib().oracle_pcg64_fill(self.state, _p(flat), _i64(flat.size))
        storage.data[:] = flat.reshape(storage.shape)


_PAIR_OPS = {"sum": 0, "max": 1, "min": 2, "distance": 3, "multiply": 4}


class OracleBackend(BackendMethods):  # pylint: disable=too-many-public-methods
    Storage = Storage
    Random = Random
    default_croupier = "local"  # PySDM/backends/numba.py:37

    def __init__(self, formulae=None, double_precision=True):
        if not double_precision:
            raise NotImplementedError()
        from pysdm_amd.formulae import Formulae  # pylint: disable=import-outside-toplevel

        self.formulae = formulae or Formulae()
        super().__init__()

    # ---- index methods ------------------------------------------------------------------------
    @staticmethod
    def identity_index(idx):
        lib().oracle_identity_index(_p(idx), _i64(len(idx)))

    @staticmethod
    def shuffle_global(idx, length, u01):
        lib().oracle_shuffle_global(_p(idx), _i64(int(length)), _p(u01))

    @staticmethod
    def shuffle_local(idx, u01, cell_start):
        lib().oracle_shuffle_local(_p(idx), _p(u01), _p(cell_start), _i64(len(cell_start) - 1))

    @staticmethod
    def sort_by_key(idx, attr):
        lib().oracle_sort_by_key(_p(idx.data), _p(attr.data), _i64(len(attr.data)))

    @staticmethod
    def remove_zero_n_or_flagged(multiplicity, idx, length):
        return int(
            lib().oracle_remove_zero_n_or_flagged(
                _p(multiplicity), _p(idx), _i64(int(length)), _i64(len(idx))
            )
        )

    @staticmethod
    def make_cell_caretaker(idx_shape, idx_dtype, cell_start_len, scheme="default"):
        tmp_idx = Storage.empty(idx_shape, idx_dtype)

        def caretaker(cell_id, cell_idx, cell_start, idx):
            lib().oracle_counting_sort_by_cell_id(
                _p(tmp_idx.data), _p(idx.data), _p(cell_id.data), _p(cell_idx.data),
                _i64(len(idx)), _p(cell_start.data), _i64(cell_start_len),
            )
            idx.data, tmp_idx.data = tmp_idx.data, idx.data

        return caretaker

    @staticmethod
    def cell_id(cell_id, cell_origin, strides):
        s = np.ascontiguousarray(strides.data.ravel())
        lib().oracle_cell_id(
            _p(cell_id.data), _p(cell_origin.data), _p(s), _i64(len(s)), _i64(len(cell_id.data))
        )

    # ---- pair methods -------------------------------------------------------------------------
    @staticmethod
    def find_pairs(cell_start, is_first_in_pair, cell_id, cell_idx, idx):
        flag = is_first_in_pair.indicator.data.view(np.uint8)
        lib().oracle_find_pairs(
            _p(cell_start.data), _p(flag), _p(cell_id.data), _p(cell_idx.data), _p(idx.data),
            _i64(len(idx)),
        )

    @staticmethod
    def sort_within_pair_by_attr(idx, is_first_in_pair, attr):
        flag = is_first_in_pair.indicator.data.view(np.uint8)
        fun = (
            lib().oracle_sort_within_pair_by_attr_i64
            if attr.data.dtype == np.int64
            else lib().oracle_sort_within_pair_by_attr_f64
        )
        fun(_p(idx.data), _i64(len(idx)), _p(flag), _p(attr.data))

    @staticmethod
    def _pair_op(name, data_out, data_in, is_first_in_pair, idx):
        flag = is_first_in_pair.indicator.data.view(np.uint8)
        fun = lib().oracle_pair_op_i64 if data_in.data.dtype == np.int64 else lib().oracle_pair_op_f64
        fun(_int(_PAIR_OPS[name]), _p(data_out.data), _i64(len(data_out.data)), _p(data_in.data),
            _p(flag), _p(idx.data), _i64(len(idx)))

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
        flag = is_first_in_pair.indicator.data.view(np.uint8)
        lib().oracle_sort_pair_f64(_p(data_out.data), _i64(len(data_out.data)), _p(data_in.data),
                                   _p(flag), _p(idx.data), _i64(len(idx)))

    # ---- collisions methods -------------------------------------------------------------------
    @staticmethod
    def normalize(prob, cell_id, cell_idx, cell_start, norm_factor, timestep, dv):
        lib().oracle_normalize(
            _p(prob.data), _i64(len(prob.data)), _p(cell_id.data), _p(cell_idx.data),
            _p(cell_start.data), _p(norm_factor.data), _i64(len(cell_start.data) - 1),
            _f64(timestep), _f64(dv),
        )

    @staticmethod
    def scale_prob_for_adaptive_sdm_gamma(*, prob, multiplicity, cell_id, dt_left, dt, dt_range,
                                          is_first_in_pair, stats_n_substep, stats_dt_min):
        flag = is_first_in_pair.indicator.data.view(np.uint8)
        lib().oracle_scale_prob_for_adaptive_sdm_gamma(
            _p(prob.data), _p(multiplicity.idx.data), _i64(len(multiplicity)),
            _p(multiplicity.data), _p(cell_id.data), _p(dt_left.data), _i64(len(dt_left.data)),
            _f64(dt), _f64(dt_range[0]), _f64(dt_range[1]), _p(flag), _p(stats_n_substep.data),
            _p(stats_dt_min.data),
        )

    @staticmethod
    def compute_gamma(*, prob, rand, multiplicity, cell_id, collision_rate_deficit,
                      collision_rate, is_first_in_pair, out):
        flag = is_first_in_pair.indicator.data.view(np.uint8)
        lib().oracle_compute_gamma(
            _p(prob.data), _p(rand.data), _p(multiplicity.idx.data), _i64(len(multiplicity)),
            _p(multiplicity.data), _p(cell_id.data), _p(collision_rate_deficit.data),
            _p(collision_rate.data), _p(flag), _p(out.data),
        )

    @staticmethod
    def adaptive_sdm_end(dt_left, cell_start):
        return int(
            lib().oracle_adaptive_sdm_end(_p(dt_left.data), _i64(len(dt_left)), _p(cell_start.data))
        )

    @staticmethod
    def collision_coalescence(*, multiplicity, idx, attributes, gamma, healthy, cell_id,
                              coalescence_rate, is_first_in_pair):
        flag = is_first_in_pair.indicator.data.view(np.uint8)
        lib().oracle_collision_coalescence(
            _p(multiplicity.data), _p(idx.data), _i64(len(idx)), _p(attributes.data),
            _i64(attributes.shape[0]), _i64(attributes.shape[1]), _p(gamma.data), _p(healthy.data),
            _p(cell_id.data), _p(coalescence_rate.data), _p(flag),
        )

    def collision_coalescence_breakup(self, *, multiplicity, idx, attributes, gamma, rand, Ec, Eb,
                                      fragment_mass, healthy, cell_id, coalescence_rate,
                                      breakup_rate, breakup_rate_deficit, is_first_in_pair,
                                      warn_overflows, particle_mass, max_multiplicity):
        flag = is_first_in_pair.indicator.data.view(np.uint8)
        n_overflow = lib().oracle_collision_coalescence_breakup(
            _p(multiplicity.data), _p(idx.data), _i64(len(idx)), _p(attributes.data),
            _i64(attributes.shape[0]), _i64(attributes.shape[1]), _p(gamma.data), _p(rand.data),
            _p(Ec.data), _p(Eb.data), _p(fragment_mass.data), _p(healthy.data), _p(cell_id.data),
            _p(coalescence_rate.data), _p(breakup_rate.data), _p(breakup_rate_deficit.data),
            _p(flag), _i64(int(max_multiplicity)), _p(particle_mass.data),
            _int(int(self.formulae.handle_all_breakups)),
        )
        if warn_overflows and n_overflow:
            import warnings  # pylint: disable=import-outside-toplevel

            warnings.warn("overflow")

    @staticmethod
    def linear_collection_efficiency(*, params, output, radii, is_first_in_pair, unit):
        flag = is_first_in_pair.indicator.data.view(np.uint8)
        par = np.asarray(params, dtype=np.float64)
        lib().oracle_linear_collection_efficiency(
            _p(par), _p(output.data), _i64(len(output.data)), _p(radii.data), _p(flag),
            _p(radii.idx.data), _i64(len(is_first_in_pair)), _f64(unit),
        )

    # ---- derived attributes -------------------------------------------------------------------
    @staticmethod
    def interpolation(*, output, radius, factor, b, c):
        lib().oracle_interpolation(
            _p(output.data), _p(radius.data), _i64(len(radius.data)), _f64(factor), _p(b.data),
            _p(c.data),
        )

    def volume_of_water_mass(self, volume, mass):
        lib().oracle_volume_of_water_mass(
            _p(volume.data), _p(mass.data), _i64(len(volume.data)),
            _f64(self.formulae.constants.rho_w),
        )

    def mass_of_water_volume(self, mass, volume):
        lib().oracle_mass_of_water_volume(
            _p(mass.data), _p(volume.data), _i64(len(volume.data)),
            _f64(self.formulae.constants.rho_w),
        )

    # ---- fragmentation ------------------------------------------------------------------------
    @staticmethod
    def exp_fragmentation(*, n_fragment, scale, frag_volume, x_plus_y, rand, vmin, nfmax,
                          tol=1e-5):
        n = len(frag_volume.data)
        lib().oracle_exp_fragmentation(_f64(scale), _p(frag_volume.data), _p(rand.data), _i64(n),
                                       _f64(tol))
        lib().oracle_fragmentation_limiters(
            _p(n_fragment.data), _p(frag_volume.data), _i64(n), _f64(vmin),
            _f64(-1.0 if nfmax is None else nfmax), _p(x_plus_y.data),
        )

    def _limiters(self, n_fragment, frag_volume, x_plus_y, vmin, nfmax):
        lib().oracle_fragmentation_limiters(
            _p(n_fragment.data), _p(frag_volume.data), _i64(len(frag_volume.data)), _f64(vmin),
            _f64(-1.0 if nfmax is None else nfmax), _p(x_plus_y.data),
        )

    def gauss_fragmentation(self, *, n_fragment, mu, sigma, frag_volume, x_plus_y, rand, vmin,
                            nfmax):
        const = self.formulae.constants
        consts = np.asarray([const.VEDDER_1987_A, const.VEDDER_1987_b], dtype=np.float64)
        lib().oracle_gauss_fragmentation(_f64(mu), _f64(sigma), _p(frag_volume.data),
                                         _p(rand.data), _i64(len(frag_volume.data)), _p(consts))
        self._limiters(n_fragment, frag_volume, x_plus_y, vmin, nfmax)

    def feingold1988_fragmentation(self, *, n_fragment, scale, frag_volume, x_plus_y, rand,
                                   fragtol, vmin, nfmax):
        lib().oracle_feingold1988_fragmentation(
            _f64(scale), _p(frag_volume.data), _p(x_plus_y.data), _p(rand.data),
            _i64(len(frag_volume.data)), _f64(fragtol))
        self._limiters(n_fragment, frag_volume, x_plus_y, vmin, nfmax)

    def slams_fragmentation(self, n_fragment, frag_volume, x_plus_y, probs, rand, vmin, nfmax):
        lib().oracle_slams_fragmentation(
            _p(n_fragment.data), _p(frag_volume.data), _p(x_plus_y.data), _p(probs.data),
            _p(rand.data), _i64(len(frag_volume.data)))
        self._limiters(n_fragment, frag_volume, x_plus_y, vmin, nfmax)

    def ll82_fragmentation(self, *, n_fragment, CKE, W, W2, St, ds, dl, dcoal, frag_volume,
                           x_plus_y, rand, vmin, nfmax, Rf, Rs, Rd, tol=1e-8):
        const = self.formulae.constants
        consts = np.asarray([const.CM, const.PI, const.VEDDER_1987_A, const.VEDDER_1987_b],
                            dtype=np.float64)
        lib().oracle_ll82_fragmentation(
            _p(CKE.data), _p(W.data), _p(W2.data), _p(St.data), _p(ds.data), _p(dl.data),
            _p(dcoal.data), _p(frag_volume.data), _p(rand.data), _p(Rf.data), _p(Rs.data),
            _p(Rd.data), _i64(len(frag_volume.data)), _f64(tol), _p(consts))
        self._limiters(n_fragment, frag_volume, x_plus_y, vmin, nfmax)

    @staticmethod
    def ll82_coalescence_check(*, Ec, dl):
        lib().oracle_ll82_coalescence_check(_p(Ec.data), _p(dl.data), _i64(len(Ec.data)))

    def straub_fragmentation(self, *, n_fragment, CW, gam, ds, frag_volume, v_max, x_plus_y, rand,
                             vmin, nfmax, Nr1, Nr2, Nr3, Nr4, Nrt, d34):
        n = len(frag_volume.data)
        const = self.formulae.constants
        consts = np.asarray(
            [const.CM, const.STRAUB_E_D1, const.STRAUB_MU2, const.VEDDER_1987_A,
             const.VEDDER_1987_b, const.PI], dtype=np.float64,
        )
        lib().oracle_straub_fragmentation(
            _p(CW.data), _p(gam.data), _p(ds.data), _p(v_max.data), _p(frag_volume.data),
            _p(rand.data), _p(Nr1.data), _p(Nr2.data), _p(Nr3.data), _p(Nr4.data), _p(Nrt.data),
            _p(d34.data), _i64(n), _p(consts),
        )
        lib().oracle_fragmentation_limiters(
            _p(n_fragment.data), _p(frag_volume.data), _i64(n), _f64(vmin),
            _f64(-1.0 if nfmax is None else nfmax), _p(x_plus_y.data),
        )

    # ---- terminal velocities other than the Gunn-Kinzer table (terminal_velocity_methods.py) ------
    def _rogers_yau_consts(self):
        const = self.formulae.constants
        return np.asarray([const.ROGERS_YAU_TERM_VEL_SMALL_K, const.ROGERS_YAU_TERM_VEL_MEDIUM_K,
                           const.ROGERS_YAU_TERM_VEL_LARGE_K,
                           const.ROGERS_YAU_TERM_VEL_SMALL_R_LIMIT,
                           const.ROGERS_YAU_TERM_VEL_MEDIUM_R_LIMIT], dtype=np.float64)

    def terminal_velocity(self, *, values, radius):
        # (raw arrays, as the reference passes `.data`)
        lib().oracle_terminal_velocity(_p(values), _p(radius), _i64(len(values)),
                                       _p(self._rogers_yau_consts()))

    @staticmethod
    def power_series(*, values, radius, num_terms, prefactors, powers):
        lib().oracle_power_series(
            _p(values), _p(radius), _i64(len(values)), _int(int(num_terms)),
            _p(np.ascontiguousarray(prefactors, dtype=np.float64)),
            _p(np.ascontiguousarray(powers, dtype=np.float64)))

    # ---- displacement (displacement_methods.py) --------------------------------------------------
    def calculate_displacement(self, *, dim, displacement, courant, cell_origin, position_in_cell,
                               n_substeps):
        n_dims = len(courant.shape)
        if n_dims not in (1, 2, 3):
            raise NotImplementedError()
        shape = np.asarray(courant.shape, dtype=np.int64)
        lib().oracle_calculate_displacement(
            _int(dim), _int(n_dims), _int(advection_scheme_id(self.formulae)),
            _p(displacement.data), _p(courant.data), _p(shape), _p(cell_origin.data),
            _p(position_in_cell.data), _i64(displacement.shape[1]), _f64(n_substeps))

    @staticmethod
    def flag_precipitated(*, cell_origin, position_in_cell, water_mass, multiplicity, idx, length,
                          healthy, precipitation_counting_level_index, displacement) -> float:
        fun = lib().oracle_flag_precipitated
        fun.restype = ctypes.c_double
        return fun(
            _p(cell_origin.data), _p(position_in_cell.data), _p(water_mass.data),
            _p(multiplicity.data), _p(idx.data), _i64(int(length)), _i64(len(idx.data)),
            _int(cell_origin.shape[0]), _p(healthy.data),
            _f64(precipitation_counting_level_index), _p(displacement.data))

    @staticmethod
    def flag_out_of_column(cell_origin, position_in_cell, idx, length, healthy,
                           domain_top_level_index):
        lib().oracle_flag_out_of_column(
            _p(cell_origin.data), _p(position_in_cell.data), _p(idx.data), _i64(int(length)),
            _i64(len(idx.data)), _int(cell_origin.shape[0]), _p(healthy.data),
            _f64(domain_top_level_index))

    # ---- moments ------------------------------------------------------------------------------
    @staticmethod
    def moments(*, moment_0, moments, multiplicity, attr_data, cell_id, idx, length, ranks, min_x,
                max_x, x_attr, weighting_attribute, weighting_rank, skip_division_by_m0):
        lib().oracle_moments(
            _p(moment_0.data), _p(moments.data), _p(multiplicity.data), _p(attr_data.data),
            _p(cell_id.data), _p(idx.data), _i64(int(length)), _p(ranks.data),
            _i64(len(ranks.data)), _i64(len(moment_0.data)), _f64(min_x), _f64(max_x),
            _p(x_attr.data), _p(weighting_attribute.data), _f64(weighting_rank),
            _int(int(skip_division_by_m0)),
        )

    @staticmethod
    def spectrum_moments(*, moment_0, moments, multiplicity, attr_data, cell_id, idx, length, rank,
                         x_bins, x_attr, weighting_attribute, weighting_rank):
        assert moments.shape[0] == x_bins.shape[0] - 1
        assert moment_0.shape == moments.shape
        lib().oracle_spectrum_moments(
            _p(moment_0.data), _p(moments.data), _p(multiplicity.data), _p(attr_data.data),
            _p(cell_id.data), _p(idx.data), _i64(int(length)), _f64(rank), _p(x_bins.data),
            _i64(moments.shape[0]), _i64(moments.shape[1]), _p(x_attr.data),
            _p(weighting_attribute.data), _f64(weighting_rank),
        )
