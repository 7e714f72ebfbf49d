"""The backend object PySDM's front-end talks to, over any `Engine`.

Drop-in contract (reference: PySDM/backends/numba.py:18-67 and the method mix-ins
under PySDM/backends/impl_numba/methods/): an *instance* with `.formulae`, `.Storage`, `.Random`,
`.default_croupier`, constructed as `Backend(formulae=None, double_precision=True)`, whose methods
carry the reference's names, keyword signatures and argument objects (Storages; `idx` objects with
a live `len()`; indexed storages exposing `.idx`; pair indicators exposing `.indicator`).  Every
method is one call of the ABI symbol of the same name: arguments are unpacked to raw arrays and
sizes here, the binding (pysdm_amd.abi) checks dtypes and contiguity, the library does the work.

`backend_class_for(engine_getter)` builds the class for an engine: `pysdm_amd.backends.HIP` binds
it to the HIP engine; the test suite binds the same code to the CPU oracle.
"""
import ctypes
import warnings

import numpy as np

from ..abi import pcg64_state_inc
from ..displacement import SCHEMES
from ..formulae import Formulae
from .storage import storage_class_for

PAIR = {"sum": 0, "max": 1, "min": 2, "distance": 3, "multiply": 4}


def _scheme_code(formulae):
    """SDM scheme code for `formulae.particle_advection` (this package's or PySDM's object:
    PySDM wraps the chosen class in a namespace that keeps its name, formulae.py:144-160)"""
    scheme = formulae.particle_advection
    code = getattr(scheme, "scheme_id", None)
    if code is None:
        code = SCHEMES[getattr(scheme, "__name__", type(scheme).__name__)]
    return code


def _nf(nfmax):
    return -1.0 if nfmax is None else float(nfmax)


def backend_class_for(engine_getter, name, doc=None):  # pylint: disable=too-many-statements
    Storage = storage_class_for(engine_getter)

    def call(symbol, *args):
        engine_getter().call(symbol, *args)

    def flag(is_first_in_pair):
        return is_first_in_pair.indicator.data

    def is_int(storage):
        return int(storage.dtype is Storage.INT)

    class Typed:
        """Storages handed over with another element type than the C parameter names.  The
        reference's Numba bodies are duck-typed (compiled per argument types), and its own unit
        tests use that: an int `out` for `sum_pair` (tests/unit_tests/backends/
        test_pair_methods.py:81-101), int probabilities for `scale_prob_for_adaptive_sdm_gamma`
        (test_collisions_methods.py:261-336), one float array as both counters of `compute_gamma`
        (dynamics/collisions/test_sdm_single_cell.py:216-258).  The C ABI is typed, so such an
        argument is converted into a temporary of the right type and - `with` exit - cast back into
        the caller's array (C-style truncation = NumPy assignment).  Off the hot path: PySDM's own
        dynamics always pass the types the ABI names, and then this is a no-op."""

        def __init__(self):
            self.pending = {}

        def __call__(self, storage, tag):
            if storage.dtype is tag:
                return storage.data
            key = id(storage.data)
            if key not in self.pending:
                eng = engine_getter()
                temporary = eng.upload(eng.download(storage.data).astype(tag))
                self.pending[key] = (storage, temporary)
            return self.pending[key][1]

        def __enter__(self):
            return self

        def __exit__(self, kind, value, trace):
            if kind is None:
                eng = engine_getter()
                for storage, temporary in self.pending.values():
                    eng.assign(storage.data,
                               eng.upload(eng.download(temporary).astype(storage.dtype)))
            return False

    class Random:  # pylint: disable=too-few-public-methods
        """NumPy-PCG64 stream; each call continues where the previous one stopped
        (impl_numba/random.py:13-19)"""

        def __init__(self, size, seed):
            assert isinstance(size, int) and isinstance(seed, int)
            self.size = size
            self.state_inc = pcg64_state_inc(seed)
            self.offset = 0

        def __call__(self, storage):
            n = int(np.prod(storage.shape))
            call("sdm_pcg64_uniform", storage.data, n, self.state_inc, self.offset)
            self.offset += n

    class Backend:  # pylint: disable=too-many-public-methods
        default_croupier = "local"  # PySDM/backends/numba.py:37

        def __init__(self, formulae=None, double_precision=True, **engine_options):
            if not double_precision:
                raise NotImplementedError("this backend computes in float64 only")
            self.formulae = formulae or Formulae()
            self.engine = engine_getter(**engine_options)
            base_init = getattr(super(), "__init__", None)
            if base_init is not None:
                base_init()

        def synchronize(self):
            self.engine.synchronize()

        # ---- index methods (index_methods.py) ---------------------------------------------------
        @staticmethod
        def identity_index(idx):
            call("sdm_identity_index", idx, int(np.prod(idx.shape)))

        @staticmethod
        def shuffle_global(idx, length, u01):
            call("sdm_shuffle_global", idx, int(length), u01)

        @staticmethod
        def shuffle_local(idx, u01, cell_start):
            call("sdm_shuffle_local", idx, u01, cell_start, int(cell_start.shape[0]) - 1)

        @staticmethod
        def sort_by_key(idx, attr):
            call("sdm_sort_by_key", idx.data, attr.data, int(attr.shape[0]))

        def remove_zero_n_or_flagged(self, multiplicity, idx, length):
            return self.engine.scalar_out("sdm_remove_zero_n_or_flagged", ctypes.c_int64,
                                          multiplicity, idx, int(length), int(idx.shape[0]))

        @staticmethod
        def make_cell_caretaker(idx_shape, idx_dtype, cell_start_len, scheme="default"):  # pylint: disable=unused-argument
            spare = Storage.empty(idx_shape, idx_dtype)

            def caretaker(cell_id, cell_idx, cell_start, idx):
                call("sdm_counting_sort_by_cell_id", spare.data, idx.data, cell_id.data,
                     cell_idx.data, len(idx), cell_start.data, cell_start_len - 1)
                idx.data, spare.data = spare.data, idx.data

            caretaker.tmp_idx = spare
            return caretaker

        @staticmethod
        def cell_id(cell_id, cell_origin, strides):
            flat = Storage.from_ndarray(np.asarray(strides.to_ndarray()).ravel())
            call("sdm_cell_id", cell_id.data, cell_origin.data, flat.data, int(flat.shape[0]),
                 int(cell_id.shape[0]))

        # ---- pair methods (pair_methods.py) -----------------------------------------------------
        @staticmethod
        def find_pairs(cell_start, is_first_in_pair, cell_id, cell_idx, idx):
            call("sdm_find_pairs", cell_start.data, flag(is_first_in_pair), cell_id.data,
                 cell_idx.data, idx.data, len(idx))

        @staticmethod
        def sort_within_pair_by_attr(idx, is_first_in_pair, attr):
            call("sdm_sort_within_pair_by_attr", idx.data, len(idx), flag(is_first_in_pair),
                 attr.data, is_int(attr))

        @staticmethod
        def _pair(op, data_out, data_in, is_first_in_pair, idx):
            with Typed() as typed:
                call("sdm_pair_op", PAIR[op], typed(data_out, Storage.FLOAT),
                     int(data_out.shape[0]), data_in.data, is_int(data_in),
                     flag(is_first_in_pair), idx.data, len(idx))

        def sum_pair(self, data_out, data_in, is_first_in_pair, idx):
            self._pair("sum", data_out, data_in, is_first_in_pair, idx)

        def max_pair(self, data_out, data_in, is_first_in_pair, idx):
            self._pair("max", data_out, data_in, is_first_in_pair, idx)

        def min_pair(self, data_out, data_in, is_first_in_pair, idx):
            self._pair("min", data_out, data_in, is_first_in_pair, idx)

        def distance_pair(self, data_out, data_in, is_first_in_pair, idx):
            self._pair("distance", data_out, data_in, is_first_in_pair, idx)

        def multiply_pair(self, data_out, data_in, is_first_in_pair, idx):
            self._pair("multiply", data_out, data_in, is_first_in_pair, idx)

        @staticmethod
        def sort_pair(data_out, data_in, is_first_in_pair, idx):
            call("sdm_sort_pair", data_out.data, int(data_out.shape[0]), data_in.data,
                 flag(is_first_in_pair), idx.data, len(idx))

        # ---- collisions methods (collisions_methods.py) -----------------------------------------
        @staticmethod
        def normalize(prob, cell_id, cell_idx, cell_start, norm_factor, timestep, dv):
            call("sdm_normalize", prob.data, int(prob.shape[0]), cell_id.data, cell_idx.data,
                 cell_start.data, norm_factor.data, int(cell_start.shape[0]) - 1, float(timestep),
                 float(dv))

        @staticmethod
        def scale_prob_for_adaptive_sdm_gamma(*, prob, multiplicity, cell_id, dt_left, dt,
                                              dt_range, is_first_in_pair, stats_n_substep,
                                              stats_dt_min):
            with Typed() as typed:
                call("sdm_scale_prob_for_adaptive_sdm_gamma", typed(prob, Storage.FLOAT),
                     multiplicity.idx.data, len(multiplicity), multiplicity.data, cell_id.data,
                     typed(dt_left, Storage.FLOAT), int(dt_left.shape[0]), float(dt),
                     float(dt_range[0]), float(dt_range[1]), flag(is_first_in_pair),
                     typed(stats_n_substep, Storage.INT), typed(stats_dt_min, Storage.FLOAT))

        @staticmethod
        def compute_gamma(*, prob, rand, multiplicity, cell_id, collision_rate_deficit,
                          collision_rate, is_first_in_pair, out):
            with Typed() as typed:
                call("sdm_compute_gamma", typed(prob, Storage.FLOAT), typed(rand, Storage.FLOAT),
                     multiplicity.idx.data, len(multiplicity), multiplicity.data, cell_id.data,
                     typed(collision_rate_deficit, Storage.INT),
                     typed(collision_rate, Storage.INT), flag(is_first_in_pair),
                     typed(out, Storage.FLOAT))

        def adaptive_sdm_end(self, dt_left, cell_start):
            return self.engine.scalar_out("sdm_adaptive_sdm_end", ctypes.c_int64, dt_left.data,
                                          len(dt_left), cell_start.data)

        @staticmethod
        def collision_coalescence(*, multiplicity, idx, attributes, gamma, healthy, cell_id,
                                  coalescence_rate, is_first_in_pair):
            with Typed() as typed:
                call("sdm_collision_coalescence", multiplicity.data, idx.data, len(idx),
                     attributes.data, int(attributes.shape[0]), int(attributes.shape[1]),
                     typed(gamma, Storage.FLOAT), healthy.data, cell_id.data,
                     typed(coalescence_rate, Storage.INT), flag(is_first_in_pair))

        def collision_coalescence_breakup(self, *, multiplicity, idx, attributes, gamma, rand, Ec,
                                          Eb, fragment_mass, healthy, cell_id, coalescence_rate,
                                          breakup_rate, breakup_rate_deficit, is_first_in_pair,
                                          warn_overflows, particle_mass, max_multiplicity):
            overflows = self.engine.zeros(1, np.int64) if warn_overflows else None
            with Typed() as typed:
                flt, cnt = Storage.FLOAT, Storage.INT
                call("sdm_collision_coalescence_breakup", multiplicity.data, idx.data, len(idx),
                     attributes.data, int(attributes.shape[0]), int(attributes.shape[1]),
                     typed(gamma, flt), typed(rand, flt), typed(Ec, flt), typed(Eb, flt),
                     typed(fragment_mass, flt), healthy.data, cell_id.data,
                     typed(coalescence_rate, cnt), typed(breakup_rate, cnt),
                     typed(breakup_rate_deficit, cnt), flag(is_first_in_pair),
                     int(max_multiplicity), particle_mass.data,
                     int(self.formulae.handle_all_breakups), overflows)
            if warn_overflows and int(self.engine.download(overflows)[0]) > 0:
                warnings.warn("overflow")

        @staticmethod
        def linear_collection_efficiency(*, params, output, radii, is_first_in_pair, unit):
            call("sdm_linear_collection_efficiency", [float(p) for p in params], output.data,
                 int(output.shape[0]), radii.data, flag(is_first_in_pair), radii.idx.data,
                 len(is_first_in_pair), float(unit))

        # ---- derived attributes ------------------------------------------------------------------
        @staticmethod
        def interpolation(*, output, radius, factor, b, c):
            call("sdm_interpolation", output.data, radius.data, int(radius.shape[0]),
                 float(factor), b.data, c.data, int(b.shape[0]))

        def volume_of_water_mass(self, volume, mass):
            call("sdm_volume_of_water_mass", volume.data, mass.data, int(volume.shape[0]),
                 self.formulae.constants.rho_w)

        def mass_of_water_volume(self, mass, volume):
            call("sdm_mass_of_water_volume", mass.data, volume.data, int(volume.shape[0]),
                 self.formulae.constants.rho_w)

        # ---- fragmentation (fragmentation_methods.py) ----------------------------------------------
        @staticmethod
        def exp_fragmentation(*, n_fragment, scale, frag_volume, x_plus_y, rand, vmin, nfmax,
                              tol=1e-5):
            call("sdm_exp_fragmentation", n_fragment.data, float(scale), frag_volume.data,
                 x_plus_y.data, rand.data, int(frag_volume.shape[0]), float(vmin), _nf(nfmax),
                 float(tol))

        def gauss_fragmentation(self, *, n_fragment, mu, sigma, frag_volume, x_plus_y, rand,
                                vmin, nfmax):
            k = self.formulae.constants
            call("sdm_gauss_fragmentation", n_fragment.data, float(mu), float(sigma),
                 frag_volume.data, x_plus_y.data, rand.data, int(frag_volume.shape[0]),
                 float(vmin), _nf(nfmax), (k.VEDDER_1987_A, k.VEDDER_1987_b))

        @staticmethod
        def feingold1988_fragmentation(*, n_fragment, scale, frag_volume, x_plus_y, rand,
                                       fragtol, vmin, nfmax):
            call("sdm_feingold1988_fragmentation", n_fragment.data, float(scale),
                 frag_volume.data, x_plus_y.data, rand.data, int(frag_volume.shape[0]),
                 float(fragtol), float(vmin), _nf(nfmax))

        @staticmethod
        def slams_fragmentation(n_fragment, frag_volume, x_plus_y, probs, rand, vmin, nfmax):
            call("sdm_slams_fragmentation", n_fragment.data, frag_volume.data, x_plus_y.data,
                 probs.data, rand.data, int(frag_volume.shape[0]), float(vmin), _nf(nfmax))

        def ll82_fragmentation(self, *, n_fragment, CKE, W, W2, St, ds, dl, dcoal, frag_volume,
                               x_plus_y, rand, vmin, nfmax, Rf, Rs, Rd, tol=1e-8):
            k = self.formulae.constants
            call("sdm_ll82_fragmentation", n_fragment.data, CKE.data, W.data, W2.data, St.data,
                 ds.data, dl.data, dcoal.data, frag_volume.data, x_plus_y.data, rand.data,
                 int(frag_volume.shape[0]), float(vmin), _nf(nfmax), Rf.data, Rs.data, Rd.data,
                 float(tol), (k.CM, k.PI, k.VEDDER_1987_A, k.VEDDER_1987_b))

        @staticmethod
        def ll82_coalescence_check(*, Ec, dl):
            call("sdm_ll82_coalescence_check", Ec.data, dl.data, int(Ec.shape[0]))

        def straub_fragmentation(self, *, n_fragment, CW, gam, ds, frag_volume, v_max, x_plus_y,
                                 rand, vmin, nfmax, Nr1, Nr2, Nr3, Nr4, Nrt, d34):
            k = self.formulae.constants
            call("sdm_straub_fragmentation", n_fragment.data, CW.data, gam.data, ds.data,
                 frag_volume.data, v_max.data, x_plus_y.data, rand.data,
                 int(frag_volume.shape[0]), float(vmin), _nf(nfmax), Nr1.data, Nr2.data, Nr3.data,
                 Nr4.data, Nrt.data, d34.data,
                 (k.CM, k.STRAUB_E_D1, k.STRAUB_MU2, k.VEDDER_1987_A, k.VEDDER_1987_b, k.PI))

        # ---- terminal velocities besides the Gunn-Kinzer table --------------------------------------
        def terminal_velocity(self, *, values, radius):
            # (raw arrays, as the reference passes `.data`)
            k = self.formulae.constants
            call("sdm_terminal_velocity", values, radius, int(np.prod(values.shape)),
                 (k.ROGERS_YAU_TERM_VEL_SMALL_K, k.ROGERS_YAU_TERM_VEL_MEDIUM_K,
                  k.ROGERS_YAU_TERM_VEL_LARGE_K, k.ROGERS_YAU_TERM_VEL_SMALL_R_LIMIT,
                  k.ROGERS_YAU_TERM_VEL_MEDIUM_R_LIMIT))

        @staticmethod
        def power_series(*, values, radius, num_terms, prefactors, powers):
            call("sdm_power_series", values, radius, int(np.prod(values.shape)), int(num_terms),
                 [float(v) for v in prefactors], [float(v) for v in powers])

        # ---- displacement (displacement_methods.py) ---------------------------------------------------
        def calculate_displacement(self, *, dim, displacement, courant, cell_origin,
                                   position_in_cell, n_substeps):
            n_dims = len(courant.shape)
            if n_dims not in (1, 2, 3):
                raise NotImplementedError()
            shape = (ctypes.c_int64 * 3)(*courant.shape, *([1] * (3 - n_dims)))
            call("sdm_calculate_displacement", int(dim), n_dims, _scheme_code(self.formulae),
                 displacement.data, courant.data, shape, cell_origin.data, position_in_cell.data,
                 int(displacement.shape[1]), float(n_substeps))

        def flag_precipitated(self, *, cell_origin, position_in_cell, water_mass, multiplicity,
                              idx, length, healthy, precipitation_counting_level_index,
                              displacement) -> float:
            return self.engine.scalar_out(
                "sdm_flag_precipitated", ctypes.c_double, cell_origin.data,
                position_in_cell.data, water_mass.data, multiplicity.data, idx.data, int(length),
                int(idx.shape[0]), int(cell_origin.shape[0]), healthy.data,
                float(precipitation_counting_level_index), displacement.data)

        @staticmethod
        def flag_out_of_column(cell_origin, position_in_cell, idx, length, healthy,
                               domain_top_level_index):
            call("sdm_flag_out_of_column", cell_origin.data, position_in_cell.data, idx.data,
                 int(length), int(idx.shape[0]), int(cell_origin.shape[0]), healthy.data,
                 float(domain_top_level_index))

        # ---- moments (moments_methods.py) -----------------------------------------------------------
        @staticmethod
        def moments(*, moment_0, moments, multiplicity, attr_data, cell_id, idx, length, ranks,
                    min_x, max_x, x_attr, weighting_attribute, weighting_rank,
                    skip_division_by_m0):
            with Typed() as typed:
                flt = Storage.FLOAT
                call("sdm_moments", moment_0.data, moments.data, multiplicity.data,
                     typed(attr_data, flt), cell_id.data, idx.data, int(length),
                     typed(ranks, flt), int(ranks.shape[0]), int(moment_0.shape[0]),
                     float(min_x), float(max_x), typed(x_attr, flt),
                     typed(weighting_attribute, flt), float(weighting_rank),
                     int(skip_division_by_m0))

        @staticmethod
        def spectrum_moments(*, moment_0, moments, multiplicity, attr_data, cell_id, idx, length,
                             rank, x_bins, x_attr, weighting_attribute, weighting_rank):
            assert moments.shape[0] == x_bins.shape[0] - 1
            assert moment_0.shape == moments.shape
            with Typed() as typed:
                flt = Storage.FLOAT
                call("sdm_spectrum_moments", moment_0.data, moments.data, multiplicity.data,
                     typed(attr_data, flt), cell_id.data, idx.data, int(length), float(rank),
                     typed(x_bins, flt), int(moments.shape[0]), int(moments.shape[1]),
                     typed(x_attr, flt), typed(weighting_attribute, flt), float(weighting_rank))

    Backend.Storage = Storage
    Backend.Random = Random
    Backend.__name__ = Backend.__qualname__ = name
    Backend.__doc__ = doc
    Storage.__module__ = Backend.__module__ = __name__
    return Backend
