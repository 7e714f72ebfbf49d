"""The SDM collision / coalescence / breakup dynamic and its pluggable parts.

Host-side mirror of PySDM/dynamics/collisions/collision.py:41-349 (`Collision`, `Coalescence`,
`Breakup`), .../collision_kernels/{golovin.py:14-16, geometric.py:15-22, constantK.py},
.../coalescence_efficiencies/{constEc.py, berry1967.py, _parameterized.py:17-25,
straub2010.py:27-50}, .../breakup_efficiencies/constEb.py,
.../breakup_fragmentations/{always_n.py, exponential.py:23-37, straub2010.py:42-101,
impl/volume_based.py:10-17} and PySDM/dynamics/impl/random_generator_optimizer*.py.

Two execution routes, both on the backend's device code:
  * the method-by-method route: the chain of backend calls the reference makes;
  * the fused route (`backend.collision_step`, one C call per time step) taken when the backend
    offers it and every plugged part has a device-side descriptor (`fused_descriptor`).
"""
import math
import warnings
from collections import namedtuple

import numpy as np

from ..attributes import Multiplicity
from . import builder_owned
from ..physics import constants as const
from ..physics.constants import si

DEFAULTS = namedtuple("_", ("dt_coal_range", "adaptive", "substeps", "max_multiplicity"))(
    dt_coal_range=(0.1 * si.second, 100.0 * si.second),
    adaptive=True,
    substeps=1,
    max_multiplicity=Multiplicity.MAX_VALUE // int(2e5),
)


# ---- random-number reuse helpers ------------------------------------------------------------
class RandomGeneratorOptimizer:  # pylint: disable=too-many-instance-attributes
    """one generator feeding `pairs_rand` (n_sd [+ shift]) then `rand` (n_sd // 2) per draw;
    with `optimized_random` one draw per time step and a window sliding by one per sub-step"""

    PAIRS = True

    def __init__(self, optimized_random, dt_min, seed):
        self.particulator = None
        self.optimized_random = optimized_random
        self.dt_min = dt_min
        self.seed = seed
        self.substep = 0
        self.pairs_rand = None
        self.rand = None
        self.rnd = None

    def register(self, builder):
        self.particulator = builder.particulator
        n_sd = self.particulator.n_sd
        shift = math.ceil(self.particulator.dt / self.dt_min) if self.optimized_random else 0
        if self.PAIRS:
            self.pairs_rand = self.particulator.Storage.empty(n_sd + shift, dtype=float)
        self.rand = self.particulator.Storage.empty(n_sd // 2, dtype=float)
        self.rnd = self.particulator.Random(n_sd + shift, self.seed)

    def reset(self):
        self.substep = 0

    def get_random_arrays(self):
        if not self.optimized_random or self.substep == 0:
            if self.PAIRS:
                self.pairs_rand.urand(self.rnd)
            self.rand.urand(self.rnd)
        shift = self.substep if self.optimized_random else 0
        self.substep += 1
        if not self.PAIRS:
            return self.rand
        return self.pairs_rand[shift : self.particulator.n_sd + shift], self.rand


class RandomGeneratorOptimizerNoPair(RandomGeneratorOptimizer):
    PAIRS = False


# ---- collision kernels ----------------------------------------------------------------------
class Golovin:
    def __init__(self, b):
        self.b = b
        self.particulator = None

    def register(self, builder):
        self.particulator = builder.particulator
        builder.request_attribute("volume")

    def __call__(self, output, is_first_in_pair):
        output.sum(self.particulator.attributes["volume"], is_first_in_pair)
        output *= self.b

    def fused_descriptor(self):
        return {"kernel": 0, "kernel_param": (float(self.b), 0.0)}

    def analytic_solution(self, x, t, x_0, N_0):
        """number density n(x, t) / N_0 of Golovin (1963) for an exponential initial spectrum with
        mean volume x_0 and concentration N_0 (collision_kernels/golovin.py:23-47); the scaled
        Bessel function keeps large arguments finite"""
        from scipy import special  # pylint: disable=import-outside-toplevel

        tau = 1 - np.exp(-N_0 * self.b * x_0 * t)
        sqrt_tau = np.sqrt(tau)
        x = np.asarray(x, dtype=float)
        result = ((1 - tau) / (x * sqrt_tau) * special.ive(1, 2 * x / x_0 * sqrt_tau)
                  * np.exp(-(1 + tau - 2 * sqrt_tau) * x / x_0))
        return result if result.ndim else float(result)


class ConstantK:
    def __init__(self, a):
        self.a = a
        self.particulator = None

    def register(self, builder):
        self.particulator = builder.particulator

    def __call__(self, output, is_first_in_pair):
        output.fill(self.a)

    def fused_descriptor(self):
        return None  # fill() also covers pair-less slots: keep the method-by-method route


class Geometric:
    def __init__(self, collection_efficiency=1.0, x="volume"):
        self.collection_efficiency = collection_efficiency
        self.x = x
        self.particulator = None
        self.pair_tmp = None

    def register(self, builder):
        self.particulator = builder.particulator
        builder.request_attribute("radius")
        builder.request_attribute("relative fall velocity")
        self.pair_tmp = self.particulator.PairwiseStorage.empty(
            self.particulator.n_sd // 2, dtype=float
        )

    def __call__(self, output, is_first_in_pair):
        output.sum(self.particulator.attributes["radius"], is_first_in_pair)
        output **= 2
        output *= const.PI * self.collection_efficiency
        self.pair_tmp.distance(
            self.particulator.attributes["relative fall velocity"], is_first_in_pair
        )
        output *= self.pair_tmp

    def fused_descriptor(self):
        return {"kernel": 1, "kernel_param": (const.PI * self.collection_efficiency, 0.0),
                "needs_gk": True}


class ParameterizedKernel:
    """gravitational kernel with Berry's parameterised collection efficiency
    (collision_kernels/impl/parameterized.py:8-30)"""

    def __init__(self, params):
        self.params = params
        self.particulator = None
        self.pair_tmp = None

    def register(self, builder):
        self.particulator = builder.particulator
        builder.request_attribute("radius")
        builder.request_attribute("relative fall velocity")
        self.pair_tmp = self.particulator.PairwiseStorage.empty(
            self.particulator.n_sd // 2, dtype=float
        )

    def __call__(self, output, is_first_in_pair):
        attributes = self.particulator.attributes
        self.particulator.backend.linear_collection_efficiency(
            params=self.params,
            output=output,
            radii=attributes["radius"],
            is_first_in_pair=is_first_in_pair,
            unit=si.um,
        )
        output **= 2
        output *= const.PI
        self.pair_tmp.max(attributes["radius"], is_first_in_pair)
        self.pair_tmp **= 2
        output *= self.pair_tmp
        self.pair_tmp.distance(attributes["relative fall velocity"], is_first_in_pair)
        output *= self.pair_tmp


    def fused_descriptor(self):
        return {"kernel": 3, "kernel_berry_params": tuple(float(p) for p in self.params),
                "kernel_berry_unit": si.um, "needs_gk": True}


class Electric(ParameterizedKernel):  # collision_kernels/electric.py (3000 V/cm, Berry 1967)
    def __init__(self):
        super().__init__((1, 1, -7, 1.78, -20.5, 1.73, 0.26, 1.47, 1, 0.82, -0.003, 4.4, 8))


class Hydrodynamic(ParameterizedKernel):  # collision_kernels/hydrodynamic.py
    def __init__(self):
        super().__init__((1, 1, -27, 1.65, -58, 1.9, 15, 1.13, 16.7, 1, 0.004, 4, 8))


class Linear:
    """K = a + b (v_j + v_k), collision_kernels/linear.py.  The reference file is an unfinished
    stub (its TODO #744): it calls `output.sum_pair`, which PairwiseStorage does not have, so it
    cannot be run there; this is the evident intent with the pairwise `sum`."""

    def fused_descriptor(self):
        return {"kernel": 5, "kernel_param": (float(self.a), float(self.b))}

    def __init__(self, a, b):
        self.a = a
        self.b = b
        self.particulator = None

    def register(self, builder):
        self.particulator = builder.particulator
        builder.request_attribute("volume")

    def __call__(self, output, is_first_in_pair):
        output.sum(self.particulator.attributes["volume"], is_first_in_pair)
        output *= self.b
        output += self.a


class SimpleGeometric:  # collision_kernels/simple_geometric.py (no fall velocity)
    def __init__(self, C):
        self.particulator = None
        self.pair_tmp = None
        self.C = C

    def register(self, builder):
        self.particulator = builder.particulator
        builder.request_attribute("radius")
        builder.request_attribute("area")
        self.pair_tmp = self.particulator.PairwiseStorage.empty(
            self.particulator.n_sd // 2, dtype=float
        )

    def __call__(self, output, is_first_in_pair):
        output[:] = self.C
        self.pair_tmp.sum(self.particulator.attributes["radius"], is_first_in_pair)
        self.pair_tmp **= 2
        output *= self.pair_tmp
        self.pair_tmp.distance(self.particulator.attributes["area"], is_first_in_pair)
        output *= self.pair_tmp

    def fused_descriptor(self):
        return {"kernel": 4, "kernel_param": (float(self.C), 0.0)}


# ---- efficiencies ---------------------------------------------------------------------------
class ConstEc:
    def __init__(self, Ec=1.0):
        self.Ec = Ec
        self.particulator = None

    def register(self, builder):
        self.particulator = builder.particulator

    def __call__(self, output, is_first_in_pair):
        output.fill(self.Ec)

    def fused_descriptor(self):
        return {"ec": 0, "ec_param": (float(self.Ec), 0.0)}


class ConstEb:
    def __init__(self, Eb=1.0):
        self.Eb = Eb
        self.particulator = None

    def register(self, builder):
        self.particulator = builder.particulator

    def __call__(self, output, is_first_in_pair):
        output.fill(self.Eb)

    def fused_descriptor(self):
        return {"eb_const": float(self.Eb)}


class Parameterized:
    def __init__(self, params):
        self.particulator = None
        self.params = params

    def register(self, builder):
        self.particulator = builder.particulator
        builder.request_attribute("radius")

    def __call__(self, output, is_first_in_pair):
        self.particulator.backend.linear_collection_efficiency(
            params=self.params,
            output=output,
            radii=self.particulator.attributes["radius"],
            is_first_in_pair=is_first_in_pair,
            unit=si.um,
        )
        output **= 2

    def fused_descriptor(self):
        return {"ec": 1, "berry_params": tuple(float(p) for p in self.params),
                "berry_unit": si.um}


class Berry1967(Parameterized):  # pylint: disable=too-few-public-methods
    def __init__(self):
        super().__init__((1, 1, -27, 1.65, -58, 1.9, 15, 1.13, 16.7, 1, 0.004, 4, 8))


class SpecifiedEff(Parameterized):  # coalescence_efficiencies/specified_eff.py
    # pylint: disable=too-many-arguments
    def __init__(self, *, A=1, B=1, D1=-27, D2=1.65, E1=-58, E2=1.9, F1=15, F2=1.13, G1=16.7,
                 G2=1, G3=0.004, Mf=4, Mg=8):
        super().__init__((A, B, D1, D2, E1, E2, F1, F2, G1, G2, G3, Mf, Mg))


class Straub2010Ec:
    def __init__(self):
        self.particulator = None
        self.arrays = {}
        self.const = None

    def register(self, builder):
        self.particulator = builder.particulator
        self.const = self.particulator.formulae.constants
        builder.request_attribute("volume")
        builder.request_attribute("relative fall velocity")
        for key in ("Sc", "tmp", "tmp2", "We"):
            self.arrays[key] = self.particulator.PairwiseStorage.empty(
                self.particulator.n_sd // 2, dtype=float
            )

    def __call__(self, output, is_first_in_pair):
        arr, attrs = self.arrays, self.particulator.attributes
        arr["tmp"].sum(attrs["volume"], is_first_in_pair)
        arr["Sc"].fill(arr["tmp"])
        arr["Sc"] *= 6 / self.const.PI
        arr["tmp"] *= 2
        arr["tmp2"].distance(attrs["relative fall velocity"], is_first_in_pair)
        arr["tmp2"] **= 2
        arr["We"].multiply(attrs["volume"], is_first_in_pair)
        arr["We"].divide_if_not_zero(arr["tmp"])
        arr["We"] *= arr["tmp2"]
        arr["We"] *= self.const.rho_w
        arr["Sc"] **= 2 / 3
        arr["Sc"] *= self.const.PI * self.const.sgm_w
        arr["We"].divide_if_not_zero(arr["Sc"])
        arr["We"] *= -1.15
        arr["We"].exp()
        output.fill(arr["We"])

    def fused_descriptor(self):
        return {"ec": 2, "needs_gk": True}


def _ll82_surface_and_kinetic(arr, attrs, extensive, is_first_in_pair, const):
    """Sc, St and CKE of a colliding pair as both Low & List classes compute them
    (coalescence_efficiencies/lowlist1982.py:46-80, breakup_fragmentations/lowlist82.py:53-78);
    `extensive` is "water mass" for Ec and "volume" for Nf"""
    arr["Sc"].sum(attrs[extensive], is_first_in_pair)
    arr["Sc"] **= 2 / 3
    arr["Sc"] *= const.PI * const.sgm_w * (6 / const.PI) ** (2 / 3)
    arr["St"].min(attrs["radius"], is_first_in_pair)
    arr["St"] *= 2
    arr["St"] **= 2
    arr["tmp"].max(attrs["radius"], is_first_in_pair)
    arr["tmp"] *= 2
    arr["tmp"] **= 2
    arr["St"] += arr["tmp"]
    arr["St"] *= const.PI * const.sgm_w
    arr["tmp"].sum(attrs[extensive], is_first_in_pair)
    arr["tmp2"].distance(attrs["relative fall velocity"], is_first_in_pair)
    arr["tmp2"] **= 2
    arr["CKE"].multiply(attrs[extensive], is_first_in_pair)
    arr["CKE"].divide_if_not_zero(arr["tmp"])
    arr["CKE"] *= arr["tmp2"]
    arr["CKE"] *= const.rho_w / 2


class LowList1982Ec:  # coalescence_efficiencies/lowlist1982.py
    def __init__(self):
        self.particulator = None
        self.arrays = {}
        self.const = None

    def register(self, builder):
        self.particulator = builder.particulator
        self.const = self.particulator.formulae.constants
        builder.request_attribute("radius")
        builder.request_attribute("water mass")
        builder.request_attribute("relative fall velocity")
        for key in ("Sc", "St", "dS", "tmp", "tmp2", "CKE", "Et", "ds", "dl"):
            self.arrays[key] = self.particulator.PairwiseStorage.empty(
                self.particulator.n_sd // 2, dtype=float
            )

    def __call__(self, output, is_first_in_pair):
        arr, attrs = self.arrays, self.particulator.attributes
        arr["ds"].min(attrs["radius"], is_first_in_pair)
        arr["ds"] *= 2
        arr["dl"].max(attrs["radius"], is_first_in_pair)
        arr["dl"] *= 2
        _ll82_surface_and_kinetic(arr, attrs, "water mass", is_first_in_pair, self.const)
        arr["dS"].fill(arr["St"])
        arr["dS"] -= arr["Sc"]
        arr["Et"].fill(arr["CKE"])
        arr["Et"] += arr["dS"]
        a = 0.778
        b = 2.61e6 / si.J**2 * si.m**2
        arr["tmp2"].fill(arr["Et"])
        arr["tmp2"] **= 2
        arr["tmp2"] *= -1.0 * b * self.const.sgm_w
        arr["tmp2"] /= arr["Sc"]
        output.fill(arr["ds"])
        output /= arr["dl"]
        output += 1.0
        output **= -2.0
        output *= a
        arr["tmp2"].exp()
        output *= arr["tmp2"]
        self.particulator.backend.ll82_coalescence_check(Ec=output, dl=arr["dl"])

    def fused_descriptor(self):
        factor = self.const.PI * self.const.sgm_w * (6 / self.const.PI) ** (2 / 3)
        return {"ec": 3, "ec_param": (0.0, factor), "needs_gk": True}


# ---- fragmentation functions ----------------------------------------------------------------
class AlwaysN:
    def __init__(self, n):
        self.particulator = None
        self.N = n

    def register(self, builder):
        self.particulator = builder.particulator

    def __call__(self, nf, frag_mass, u01, is_first_in_pair):
        nf.fill(self.N)
        frag_mass.sum(self.particulator.attributes["water mass"], is_first_in_pair)
        frag_mass /= self.N

    def fused_descriptor(self):
        return {"frag": 0, "frag_param": (float(self.N), 0.0)}


class ConstantMass:  # breakup_fragmentations/constant_mass.py
    def __init__(self, c):
        self.particulator = None
        self.C = c

    def register(self, builder):
        self.particulator = builder.particulator

    def __call__(self, nf, frag_mass, u01, is_first_in_pair):
        frag_mass[:] = self.C
        nf.sum(self.particulator.attributes["water mass"], is_first_in_pair)
        nf /= self.C

    def fused_descriptor(self):
        return {"frag": 6, "frag_param": (float(self.C), 0.0)}


class VolumeBasedFragmentationFunction:
    def __init__(self):
        self.particulator = None

    def register(self, builder):
        self.particulator = builder.particulator
        builder.request_attribute("volume")

    def __call__(self, nf, frag_mass, u01, is_first_in_pair):
        # volumes are written into the mass array, then converted in place
        self.compute_fragment_number_and_volumes(nf, frag_mass, u01, is_first_in_pair)
        self.particulator.backend.mass_of_water_volume(frag_mass, frag_mass)

    def compute_fragment_number_and_volumes(self, nf, frag_volume, u01, is_first_in_pair):
        raise NotImplementedError()


class Exponential(VolumeBasedFragmentationFunction):
    def __init__(self, scale, vmin=0.0, nfmax=None):
        super().__init__()
        self.scale = scale
        self.vmin = vmin
        self.nfmax = nfmax
        self.sum_of_volumes = None

    def register(self, builder):
        super().register(builder)
        self.sum_of_volumes = self.particulator.PairwiseStorage.empty(
            self.particulator.n_sd // 2, dtype=float
        )

    def compute_fragment_number_and_volumes(self, nf, frag_volume, u01, is_first_in_pair):
        self.sum_of_volumes.sum(self.particulator.attributes["volume"], is_first_in_pair)
        self.particulator.backend.exp_fragmentation(
            n_fragment=nf,
            scale=self.scale,
            frag_volume=frag_volume,
            x_plus_y=self.sum_of_volumes,
            rand=u01,
            vmin=self.vmin,
            nfmax=self.nfmax,
        )

    def fused_descriptor(self):
        return {"frag": 1, "frag_param": (float(self.scale), 0.0), "frag_vmin": float(self.vmin),
                "frag_nfmax": -1.0 if self.nfmax is None else float(self.nfmax)}


class _SumOfVolumes(VolumeBasedFragmentationFunction):
    def __init__(self, vmin, nfmax):
        super().__init__()
        self.vmin = vmin
        self.nfmax = nfmax
        self.sum_of_volumes = None

    def register(self, builder):
        super().register(builder)
        self.sum_of_volumes = self.particulator.PairwiseStorage.empty(
            self.particulator.n_sd // 2, dtype=float
        )

    def _sum(self, is_first_in_pair):
        self.sum_of_volumes.sum(self.particulator.attributes["volume"], is_first_in_pair)
        return self.sum_of_volumes


class Gaussian(_SumOfVolumes):  # breakup_fragmentations/gaussian.py (mu, sigma: volumes)
    def __init__(self, mu, sigma, vmin=0.0, nfmax=None):
        super().__init__(vmin, nfmax)
        self.mu = mu
        self.sigma = sigma

    def compute_fragment_number_and_volumes(self, nf, frag_volume, u01, is_first_in_pair):
        self.particulator.backend.gauss_fragmentation(
            n_fragment=nf, mu=self.mu, sigma=self.sigma, frag_volume=frag_volume,
            x_plus_y=self._sum(is_first_in_pair), rand=u01, vmin=self.vmin, nfmax=self.nfmax,
        )

    def fused_descriptor(self):
        return {"frag": 3, "frag_param": (float(self.mu), float(self.sigma)),
                "frag_vmin": float(self.vmin),
                "frag_nfmax": -1.0 if self.nfmax is None else float(self.nfmax)}


class Feingold1988(_SumOfVolumes):  # breakup_fragmentations/feingold1988.py
    def __init__(self, scale, fragtol=1e-3, vmin=0.0, nfmax=None):
        super().__init__(vmin, nfmax)
        self.scale = scale
        self.fragtol = fragtol

    def compute_fragment_number_and_volumes(self, nf, frag_volume, u01, is_first_in_pair):
        self.particulator.backend.feingold1988_fragmentation(
            n_fragment=nf, scale=self.scale, frag_volume=frag_volume,
            x_plus_y=self._sum(is_first_in_pair), rand=u01, fragtol=self.fragtol,
            vmin=self.vmin, nfmax=self.nfmax,
        )

    def fused_descriptor(self):
        return {"frag": 4, "frag_param": (float(self.scale), float(self.fragtol)),
                "frag_vmin": float(self.vmin),
                "frag_nfmax": -1.0 if self.nfmax is None else float(self.nfmax)}


class SLAMS(_SumOfVolumes):  # breakup_fragmentations/slams.py
    def __init__(self, vmin=0.0, nfmax=None):
        super().__init__(vmin, nfmax)
        self.p_vec = None

    def register(self, builder):
        super().register(builder)
        self.p_vec = self.particulator.PairwiseStorage.empty(
            self.particulator.n_sd // 2, dtype=float
        )

    def compute_fragment_number_and_volumes(self, nf, frag_volume, u01, is_first_in_pair):
        self.particulator.backend.slams_fragmentation(
            n_fragment=nf, frag_volume=frag_volume, x_plus_y=self._sum(is_first_in_pair),
            probs=self.p_vec, rand=u01, vmin=self.vmin, nfmax=self.nfmax,
        )

    def fused_descriptor(self):
        return {"frag": 5, "frag_vmin": float(self.vmin),
                "frag_nfmax": -1.0 if self.nfmax is None else float(self.nfmax)}


class LowList1982Nf(_SumOfVolumes):  # breakup_fragmentations/lowlist82.py
    def __init__(self, vmin=0.0, nfmax=None):
        super().__init__(vmin, nfmax)
        self.arrays = {}
        self.ll82_tmp = {}
        self.const = None

    def register(self, builder):
        super().register(builder)
        self.const = self.particulator.formulae.constants
        builder.request_attribute("radius")
        builder.request_attribute("relative fall velocity")
        half = self.particulator.n_sd // 2
        for key in ("Sc", "St", "tmp", "tmp2", "CKE", "We", "W2", "ds", "dl", "dcoal"):
            self.arrays[key] = self.particulator.PairwiseStorage.empty(half, dtype=float)
        for key in ("Rf", "Rs", "Rd"):
            self.ll82_tmp[key] = self.particulator.PairwiseStorage.empty(half, dtype=float)

    def compute_fragment_number_and_volumes(self, nf, frag_volume, u01, is_first_in_pair):
        arr, attrs = self.arrays, self.particulator.attributes
        arr["ds"].min(attrs["radius"], is_first_in_pair)
        arr["ds"] *= 2
        arr["dl"].max(attrs["radius"], is_first_in_pair)
        arr["dl"] *= 2
        arr["dcoal"].sum(attrs["volume"], is_first_in_pair)
        arr["dcoal"] /= self.const.PI / 6
        arr["dcoal"] **= 1 / 3
        _ll82_surface_and_kinetic(arr, attrs, "volume", is_first_in_pair, self.const)
        arr["We"].fill(arr["CKE"])
        arr["W2"].fill(arr["CKE"])
        arr["We"].divide_if_not_zero(arr["Sc"])
        arr["W2"].divide_if_not_zero(arr["St"])
        for key in ("Rf", "Rs", "Rd"):
            self.ll82_tmp[key] *= 0.0
        self.particulator.backend.ll82_fragmentation(
            n_fragment=nf, CKE=arr["CKE"], W=arr["We"], W2=arr["W2"], St=arr["St"],
            ds=arr["ds"], dl=arr["dl"], dcoal=arr["dcoal"], frag_volume=frag_volume,
            x_plus_y=self._sum(is_first_in_pair), rand=u01, vmin=self.vmin, nfmax=self.nfmax,
            Rf=self.ll82_tmp["Rf"], Rs=self.ll82_tmp["Rs"], Rd=self.ll82_tmp["Rd"],
        )

    def fused_descriptor(self):
        factor = self.const.PI * self.const.sgm_w * (6 / self.const.PI) ** (2 / 3)
        return {"frag": 7, "frag_param": (factor, 0.0), "needs_gk": True, "frag_vmin": float(self.vmin),
                "frag_nfmax": -1.0 if self.nfmax is None else float(self.nfmax)}


class Straub2010Nf(VolumeBasedFragmentationFunction):
    # pylint: disable=too-many-instance-attributes
    def __init__(self, vmin=0.0, nfmax=None):
        super().__init__()
        self.vmin = vmin
        self.nfmax = nfmax
        self.arrays = {}
        self.straub_tmp = {}
        self.max_size = None
        self.sum_of_volumes = None
        self.const = None

    def register(self, builder):
        super().register(builder)
        pairwise = self.particulator.PairwiseStorage
        n_pairs = self.particulator.n_sd // 2
        self.max_size = pairwise.empty(n_pairs, dtype=float)
        self.sum_of_volumes = pairwise.empty(n_pairs, dtype=float)
        self.const = self.particulator.formulae.constants
        builder.request_attribute("radius")
        builder.request_attribute("relative fall velocity")
        for key in ("Sc", "tmp", "tmp2", "CKE", "We", "gam", "CW", "ds"):
            self.arrays[key] = pairwise.empty(n_pairs, dtype=float)
        for key in ("Nr1", "Nr2", "Nr3", "Nr4", "Nrt", "d34"):
            self.straub_tmp[key] = pairwise.empty(n_pairs, dtype=float)

    def sc_factor(self):
        return self.const.PI * self.const.sgm_w * (6 / self.const.PI) ** (2 / 3)

    def compute_fragment_number_and_volumes(self, nf, frag_volume, u01, is_first_in_pair):
        arr, attrs = self.arrays, self.particulator.attributes
        self.max_size.max(attrs["volume"], is_first_in_pair)
        self.sum_of_volumes.sum(attrs["volume"], is_first_in_pair)
        arr["ds"].min(attrs["radius"], is_first_in_pair)
        arr["ds"] *= 2
        # dimensionless numbers; CW = CKE * We
        arr["tmp"].sum(attrs["volume"], is_first_in_pair)
        arr["Sc"].fill(arr["tmp"])
        arr["Sc"] **= 2 / 3
        arr["Sc"] *= self.sc_factor()
        arr["tmp2"].distance(attrs["relative fall velocity"], is_first_in_pair)
        arr["tmp2"] **= 2
        arr["CKE"].multiply(attrs["volume"], is_first_in_pair)
        arr["CKE"].divide_if_not_zero(arr["tmp"])
        arr["CKE"] *= arr["tmp2"]
        arr["CKE"] *= self.const.rho_w / 2
        arr["We"].fill(arr["CKE"])
        arr["We"].divide_if_not_zero(arr["Sc"])
        arr["CW"].fill(arr["We"])
        arr["CW"] *= arr["CKE"]
        arr["CW"] /= si.uJ
        arr["gam"].max(attrs["radius"], is_first_in_pair)
        arr["tmp"].min(attrs["radius"], is_first_in_pair)
        arr["gam"].divide_if_not_zero(arr["tmp"])
        for key in ("Nr1", "Nr2", "Nr3", "Nr4", "Nrt"):
            self.straub_tmp[key].fill(0)
        self.particulator.backend.straub_fragmentation(
            n_fragment=nf, CW=arr["CW"], gam=arr["gam"], ds=arr["ds"], frag_volume=frag_volume,
            v_max=self.max_size, x_plus_y=self.sum_of_volumes, rand=u01, vmin=self.vmin,
            nfmax=self.nfmax, **self.straub_tmp,
        )

    def fused_descriptor(self):
        return {"frag": 2, "frag_param": (0.0, self.sc_factor()), "frag_vmin": float(self.vmin),
                "frag_nfmax": -1.0 if self.nfmax is None else float(self.nfmax),
                "needs_gk": True}


# ---- the dynamic ----------------------------------------------------------------------------
@builder_owned
class Collision:  # pylint: disable=too-many-instance-attributes
    DYNAMIC_KEY = "Collision"

    def __init__(self, *, collision_kernel, coalescence_efficiency, breakup_efficiency,
                 fragmentation_function, croupier=None, optimized_random=False,
                 substeps: int = DEFAULTS.substeps, adaptive: bool = DEFAULTS.adaptive,
                 dt_coal_range=DEFAULTS.dt_coal_range, enable_breakup: bool = True,
                 warn_overflows: bool = True, fused=None):
        assert substeps == 1 or adaptive is False
        assert dt_coal_range[0] > 0
        self.particulator = None
        self.enable = True
        self.enable_breakup = enable_breakup
        self.warn_overflows = warn_overflows
        self.max_multiplicity = DEFAULTS.max_multiplicity
        self.collision_kernel = collision_kernel
        self.compute_coalescence_efficiency = coalescence_efficiency
        self.compute_breakup_efficiency = breakup_efficiency
        self.compute_number_of_fragments = fragmentation_function
        self.rnd_opt_frag = self.rnd_opt_coll = self.rnd_opt_proc = None
        self.croupier = croupier
        self.optimized_random = optimized_random
        self.__substeps = substeps
        self.adaptive = adaptive
        self.dt_coal_range = tuple(dt_coal_range)
        self.fused = fused  # None: use the fused route when available; False: never
        self._fused_state = None
        for name in ("stats_n_substep", "stats_dt_min", "kernel_temp", "n_fragment",
                     "fragment_mass", "Ec_temp", "Eb_temp", "norm_factor_temp", "gamma",
                     "is_first_in_pair", "dt_left", "collision_rate", "collision_rate_deficit",
                     "coalescence_rate", "breakup_rate", "breakup_rate_deficit"):
            setattr(self, name, None)

    @property
    def substeps(self):
        return self.__substeps

    def register(self, builder):
        part = self.particulator = builder.particulator
        rnd_args = {"optimized_random": self.optimized_random, "dt_min": self.dt_coal_range[0],
                    "seed": builder.formulae.seed}
        self.rnd_opt_coll = RandomGeneratorOptimizer(**rnd_args)
        if self.enable_breakup:
            self.rnd_opt_proc = RandomGeneratorOptimizerNoPair(**rnd_args)
            self.rnd_opt_frag = RandomGeneratorOptimizerNoPair(**rnd_args)
        if part.n_sd < 2:
            raise ValueError("No one to collide with!")
        if self.dt_coal_range[1] > part.dt:
            self.dt_coal_range = (self.dt_coal_range[0], part.dt)
        assert self.dt_coal_range[0] <= self.dt_coal_range[1]

        n_pairs, n_cell = part.n_sd // 2, part.mesh.n_cell

        def pairwise():
            return part.PairwiseStorage.empty(n_pairs, dtype=float)

        def counter():
            return part.Storage.from_ndarray(np.zeros(n_cell, dtype=int))

        self.kernel_temp = pairwise()
        self.norm_factor_temp = part.Storage.empty(n_cell, dtype=float)
        self.gamma = pairwise()
        self.is_first_in_pair = part.PairIndicator(part.n_sd)
        self.dt_left = part.Storage.empty(n_cell, dtype=float)
        self.stats_n_substep = part.Storage.empty(n_cell, dtype=int)
        self.stats_n_substep[:] = 0 if self.adaptive else self.__substeps
        self.stats_dt_min = part.Storage.empty(n_cell, dtype=float)
        self.stats_dt_min[:] = np.nan
        self.rnd_opt_coll.register(builder)
        self.collision_kernel.register(builder)
        if self.croupier is None:
            self.croupier = part.backend.default_croupier
        self.collision_rate = counter()
        self.collision_rate_deficit = counter()
        self.coalescence_rate = counter()
        if self.enable_breakup:
            self.n_fragment = pairwise()
            self.fragment_mass = pairwise()
            self.Ec_temp = pairwise()
            self.Eb_temp = pairwise()
            self.rnd_opt_proc.register(builder)
            self.rnd_opt_frag.register(builder)
            self.compute_coalescence_efficiency.register(builder)
            self.compute_breakup_efficiency.register(builder)
            self.compute_number_of_fragments.register(builder)
            self.breakup_rate = counter()
            self.breakup_rate_deficit = counter()

    # -- fused route --------------------------------------------------------------------------
    def fused_config(self):
        """device-side description of this dynamic, or None if some part has none"""
        parts = [self.collision_kernel]
        if self.enable_breakup:
            parts += [self.compute_coalescence_efficiency, self.compute_breakup_efficiency,
                      self.compute_number_of_fragments]
        cfg = {}
        for part in parts:
            desc = part.fused_descriptor() if hasattr(part, "fused_descriptor") else None
            if desc is None:
                return None
            cfg["needs_gk"] = cfg.get("needs_gk", False) or desc.pop("needs_gk", False)
            cfg.update(desc)
        return cfg

    def _use_fused(self):
        if self.fused is False or not hasattr(self.particulator.backend, "collision_step"):
            return False
        if self._fused_state is None:
            cfg = self.fused_config()
            self._fused_state = (
                False if cfg is None else self.particulator.backend.make_collision_step(self, cfg)
            )
        return self._fused_state is not False

    def run_steps(self, n_steps):
        """n_steps consecutive `__call__`s in one backend call; False if not available"""
        if not self.enable or not self._use_fused():
            return False
        with self.particulator.timers[self.DYNAMIC_KEY]:
            self.particulator.backend.collision_step(self._fused_state, n_steps)
        return True

    def __call__(self):
        if not self.enable:
            return
        if self._use_fused():
            self.particulator.backend.collision_step(self._fused_state)
            return
        attributes = self.particulator.attributes
        if not self.adaptive:
            for _ in range(self.__substeps):
                self.step()
        else:
            self.dt_left[:] = self.particulator.dt
            while attributes.get_working_length() != 0:
                attributes.cell_idx.sort_by_key(self.dt_left)
                self.step()
                attributes.cut_working_length(self.particulator.adaptive_sdm_end(self.dt_left))
            attributes.reset_working_length()
            attributes.reset_cell_idx()
        self.rnd_opt_coll.reset()
        if self.enable_breakup:
            self.rnd_opt_proc.reset()
            self.rnd_opt_frag.reset()

    def step(self):
        pairs_rand, rand = self.rnd_opt_coll.get_random_arrays()
        self.toss_candidate_pairs_and_sort_within_pair_by_multiplicity(
            self.is_first_in_pair, pairs_rand
        )
        prob = self.gamma
        self.compute_probabilities_of_collision(self.is_first_in_pair, out=prob)
        proc_rand = None
        if self.enable_breakup:
            proc_rand = self.rnd_opt_proc.get_random_arrays()
            rand_frag = self.rnd_opt_frag.get_random_arrays()
            self.compute_coalescence_efficiency(self.Ec_temp, self.is_first_in_pair)
            self.compute_breakup_efficiency(self.Eb_temp, self.is_first_in_pair)
            self.compute_number_of_fragments(
                self.n_fragment, self.fragment_mass, rand_frag, self.is_first_in_pair
            )
        self.compute_gamma(prob=prob, rand=rand, is_first_in_pair=self.is_first_in_pair,
                           out=self.gamma)
        self.particulator.collision_coalescence_breakup(
            enable_breakup=self.enable_breakup,
            gamma=self.gamma,
            rand=proc_rand,
            Ec=self.Ec_temp,
            Eb=self.Eb_temp,
            fragment_mass=self.fragment_mass,
            coalescence_rate=self.coalescence_rate,
            breakup_rate=self.breakup_rate,
            breakup_rate_deficit=self.breakup_rate_deficit,
            is_first_in_pair=self.is_first_in_pair,
            warn_overflows=self.warn_overflows,
            max_multiplicity=self.max_multiplicity,
        )

    def toss_candidate_pairs_and_sort_within_pair_by_multiplicity(self, is_first_in_pair, u01):
        attributes = self.particulator.attributes
        attributes.permutation(u01, self.croupier == "local")
        is_first_in_pair.update(attributes.cell_start, attributes.cell_idx, attributes["cell id"])
        self.particulator.sort_within_pair_by_attr(is_first_in_pair, attr_name="multiplicity")

    def compute_probabilities_of_collision(self, is_first_in_pair, out):
        """eq. (20) of Shima et al. 2009"""
        self.collision_kernel(self.kernel_temp, is_first_in_pair)
        out.max(self.particulator.attributes["multiplicity"], is_first_in_pair)
        out *= self.kernel_temp
        self.particulator.normalize(out, self.norm_factor_temp)

    def compute_gamma(self, prob, rand, is_first_in_pair, out):
        attributes = self.particulator.attributes
        if self.adaptive:
            self.particulator.backend.scale_prob_for_adaptive_sdm_gamma(
                prob=prob,
                multiplicity=attributes["multiplicity"],
                cell_id=attributes["cell id"],
                dt_left=self.dt_left,
                dt=self.particulator.dt,
                dt_range=self.dt_coal_range,
                is_first_in_pair=is_first_in_pair,
                stats_n_substep=self.stats_n_substep,
                stats_dt_min=self.stats_dt_min,
            )
            if self.stats_dt_min.amin() == self.dt_coal_range[0]:
                warnings.warn("adaptive time-step reached dt_min")
        else:
            prob /= self.__substeps
        self.particulator.backend.compute_gamma(
            prob=prob,
            rand=rand,
            multiplicity=attributes["multiplicity"],
            cell_id=attributes["cell id"],
            collision_rate_deficit=self.collision_rate_deficit,
            collision_rate=self.collision_rate,
            is_first_in_pair=is_first_in_pair,
            out=out,
        )


class Coalescence(Collision):
    def __init__(self, *, collision_kernel, coalescence_efficiency=None, croupier=None,
                 optimized_random=False, substeps: int = DEFAULTS.substeps,
                 adaptive: bool = DEFAULTS.adaptive, dt_coal_range=DEFAULTS.dt_coal_range,
                 fused=None):
        super().__init__(
            collision_kernel=collision_kernel,
            coalescence_efficiency=coalescence_efficiency or ConstEc(Ec=1),
            breakup_efficiency=ConstEb(Eb=0),
            fragmentation_function=AlwaysN(n=1),
            croupier=croupier,
            optimized_random=optimized_random,
            substeps=substeps,
            adaptive=adaptive,
            dt_coal_range=dt_coal_range,
            enable_breakup=False,
            fused=fused,
        )


class Breakup(Collision):
    def __init__(self, *, collision_kernel, fragmentation_function, croupier=None,
                 optimized_random=False, substeps: int = DEFAULTS.substeps,
                 adaptive: bool = DEFAULTS.adaptive, dt_coal_range=DEFAULTS.dt_coal_range,
                 warn_overflows=True, fused=None):
        super().__init__(
            collision_kernel=collision_kernel,
            coalescence_efficiency=ConstEc(Ec=0.0),
            breakup_efficiency=ConstEb(Eb=1.0),
            fragmentation_function=fragmentation_function,
            croupier=croupier,
            optimized_random=optimized_random,
            substeps=substeps,
            adaptive=adaptive,
            dt_coal_range=dt_coal_range,
            warn_overflows=warn_overflows,
            fused=fused,
        )
