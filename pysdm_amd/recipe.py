"""What a collision step is made of, as plain data.

A `CollisionSetup` names the pluggable parts of the SDM collision dynamic - collision kernel,
coalescence efficiency, breakup efficiency, fragmentation function - and the options of the
Monte-Carlo scheme.  Every part is a frozen dataclass carrying its parameters and two renderings:

* `descriptor(k)`: the fields of `sdm_step_cfg` (include/sdm_hip.h) that select and parameterise
  the part in the fused step (`sdm_collision_step`);
* `program(k)`: a *pair program* - the part as a short list of register instructions over
  pair-long arrays - executed by `pysdm_amd.chain` through the fine-grained symbols of the ABI.
  The instruction order is the order of the reference's element-wise passes (each rounds once), so
  both renderings and the reference agree to the bit.

Reference (file:line, relative to the reference root) of what each part computes:
PySDM/dynamics/collisions/collision_kernels/{golovin.py:14-16, geometric.py:15-22, constantK.py,
linear.py, simple_geometric.py, electric.py, hydrodynamic.py, impl/parameterized.py:8-30},
coalescence_efficiencies/{constEc.py, berry1967.py, specified_eff.py, _parameterized.py:17-25,
straub2010.py:27-50, lowlist1982.py:30-103}, breakup_efficiencies/constEb.py,
breakup_fragmentations/{always_n.py, constant_mass.py, exponential.py:23-37, gaussian.py,
feingold1988.py, slams.py, straub2010.py:42-101, lowlist82.py:37-117, impl/volume_based.py:10-17};
options: PySDM/dynamics/collisions/collision.py:41-172.

Instruction set (registers are names; "out" / "nf" / "fm" are a program's results):
  ("pair", op, dst, column)   dst = pair-wise op (sum max min distance multiply) of a column
  ("fill", dst, x) ("copy", dst, src) ("exp", dst) ("pow", dst, p)
  ("mul" | "add" | "div", dst, x-or-register) ("sub", dst, register) ("divnz", dst, register)
  ("lce", dst, params, unit)  Berry's parameterised linear collection efficiency
  ("call", symbol, *args)     any other ABI symbol; register names are replaced by their arrays
  ("volume_to_mass", dst)
"""
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

from .physics import constants as const
from .physics.constants import si

# sdm_step_cfg codes (include/sdm_hip.h)
KERNEL_CODES = {"golovin": 0, "geometric": 1, "constant": 2, "parameterized": 3,
                "simple_geometric": 4, "linear": 5}
EC_CODES = {"const": 0, "berry1967": 1, "straub2010": 2, "lowlist1982": 3}
FRAG_CODES = {"always_n": 0, "exponential": 1, "straub2010": 2, "gaussian": 3,
              "feingold1988": 4, "slams": 5, "constant_mass": 6, "lowlist1982": 7}

BERRY_HYDRODYNAMIC = (1, 1, -27, 1.65, -58, 1.9, 15, 1.13, 16.7, 1, 0.004, 4, 8)
BERRY_ELECTRIC = (1, 1, -7, 1.78, -20.5, 1.73, 0.26, 1.47, 1, 0.82, -0.003, 4.4, 8)

MAX_MULTIPLICITY = np.iinfo(np.int64).max // int(2e5)  # collision.py:36


def _nfmax(value):
    return -1.0 if value is None else float(value)


def surface_factor(k):
    """pi sigma_w (6/pi)^(2/3): surface energy of the coalesced drop per volume^(2/3)"""
    return k.PI * k.sgm_w * (6 / k.PI) ** (2 / 3)


def straub_consts(k):
    return (k.CM, k.STRAUB_E_D1, k.STRAUB_MU2, k.VEDDER_1987_A, k.VEDDER_1987_b, k.PI)


# ---- collision kernels -----------------------------------------------------------------------
@dataclass(frozen=True)
class Golovin:
    b: float

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"kernel": KERNEL_CODES["golovin"], "kernel_param": (float(self.b), 0.0)}

    def program(self, k):  # pylint: disable=unused-argument
        return [("pair", "sum", "out", "volume"), ("mul", "out", self.b)]

    def analytic_solution(self, x, t, x_0, N_0):
        """n(x, t) / N_0 of Golovin (1963) for an exponential initial spectrum of mean volume x_0
        (the form of collision_kernels/golovin.py:23-47; scaled Bessel function for large x)"""
        from scipy import special  # pylint: disable=import-outside-toplevel

        tau = 1 - np.exp(-N_0 * self.b * x_0 * t)
        root = np.sqrt(tau)
        x = np.asarray(x, dtype=float)
        value = ((1 - tau) / (x * root) * special.ive(1, 2 * x / x_0 * root)
                 * np.exp(-(1 + tau - 2 * root) * x / x_0))
        return value if value.ndim else float(value)


@dataclass(frozen=True)
class Geometric:
    collection_efficiency: float = 1.0

    def descriptor(self, k):
        return {"kernel": KERNEL_CODES["geometric"],
                "kernel_param": (k.PI * self.collection_efficiency, 0.0), "needs_gk": True}

    def program(self, k):
        return [("pair", "sum", "out", "radius"), ("pow", "out", 2),
                ("mul", "out", k.PI * self.collection_efficiency),
                ("pair", "distance", "t", "fall velocity"), ("mul", "out", "t")]


@dataclass(frozen=True)
class ConstantK:
    a: float

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"kernel": KERNEL_CODES["constant"], "kernel_param": (float(self.a), 0.0)}

    def program(self, k):  # pylint: disable=unused-argument
        return [("fill", "out", self.a)]


@dataclass(frozen=True)
class Linear:
    """K = a + b (v_j + v_k); a stub in the reference (its TODO #744), run as evidently meant"""
    a: float
    b: float

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"kernel": KERNEL_CODES["linear"], "kernel_param": (float(self.a), float(self.b))}

    def program(self, k):  # pylint: disable=unused-argument
        return [("pair", "sum", "out", "volume"), ("mul", "out", self.b), ("add", "out", self.a)]


@dataclass(frozen=True)
class SimpleGeometric:
    C: float  # pylint: disable=invalid-name

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"kernel": KERNEL_CODES["simple_geometric"], "kernel_param": (float(self.C), 0.0)}

    def program(self, k):  # pylint: disable=unused-argument
        return [("fill", "out", self.C), ("pair", "sum", "t", "radius"), ("pow", "t", 2),
                ("mul", "out", "t"), ("pair", "distance", "t", "area"), ("mul", "out", "t")]


@dataclass(frozen=True)
class ParameterizedKernel:
    """gravitational kernel with Berry's (1967) parameterised collection efficiency"""
    params: Tuple[float, ...]

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"kernel": KERNEL_CODES["parameterized"],
                "kernel_berry_params": tuple(float(p) for p in self.params),
                "kernel_berry_unit": si.um, "needs_gk": True}

    def program(self, k):
        return [("lce", "out", self.params, si.um), ("pow", "out", 2), ("mul", "out", k.PI),
                ("pair", "max", "t", "radius"), ("pow", "t", 2), ("mul", "out", "t"),
                ("pair", "distance", "t", "fall velocity"), ("mul", "out", "t")]


def Electric():  # pylint: disable=invalid-name
    return ParameterizedKernel(BERRY_ELECTRIC)


def Hydrodynamic():  # pylint: disable=invalid-name
    return ParameterizedKernel(BERRY_HYDRODYNAMIC)


# ---- efficiencies ----------------------------------------------------------------------------
@dataclass(frozen=True)
class ConstEc:
    Ec: float = 1.0  # pylint: disable=invalid-name

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"ec": EC_CODES["const"], "ec_param": (float(self.Ec), 0.0)}

    def program(self, k):  # pylint: disable=unused-argument
        return [("fill", "out", self.Ec)]


@dataclass(frozen=True)
class ConstEb:
    Eb: float = 1.0  # pylint: disable=invalid-name

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"eb_const": float(self.Eb)}

    def program(self, k):  # pylint: disable=unused-argument
        return [("fill", "out", self.Eb)]


@dataclass(frozen=True)
class SpecifiedEff:
    """Ec = (linear collection efficiency)^2 with the 13 parameters (A, B, D1, D2, E1, E2, F1, F2,
    G1, G2, G3, Mf, Mg); the defaults are Berry's (1967) hydrodynamic values"""
    params: Tuple[float, ...] = BERRY_HYDRODYNAMIC

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"ec": EC_CODES["berry1967"],
                "berry_params": tuple(float(p) for p in self.params), "berry_unit": si.um}

    def program(self, k):  # pylint: disable=unused-argument
        return [("lce", "out", self.params, si.um), ("pow", "out", 2)]


def Berry1967():  # pylint: disable=invalid-name
    return SpecifiedEff()


def _weber_like(mass_like, k, scale):
    """registers tmp (sum), tmp2 (dv^2) and CKE = scale * x_j x_k / (x_j + x_k) * dv^2"""
    return [("pair", "sum", "tmp", mass_like), ("pair", "distance", "tmp2", "fall velocity"),
            ("pow", "tmp2", 2), ("pair", "multiply", "CKE", mass_like), ("divnz", "CKE", "tmp"),
            ("mul", "CKE", "tmp2"), ("mul", "CKE", scale)]


@dataclass(frozen=True)
class Straub2010Ec:
    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"ec": EC_CODES["straub2010"], "needs_gk": True}

    def program(self, k):
        return [("pair", "sum", "tmp", "volume"), ("copy", "Sc", "tmp"), ("mul", "Sc", 6 / k.PI),
                ("mul", "tmp", 2), ("pair", "distance", "tmp2", "fall velocity"),
                ("pow", "tmp2", 2), ("pair", "multiply", "We", "volume"), ("divnz", "We", "tmp"),
                ("mul", "We", "tmp2"), ("mul", "We", k.rho_w), ("pow", "Sc", 2 / 3),
                ("mul", "Sc", k.PI * k.sgm_w), ("divnz", "We", "Sc"), ("mul", "We", -1.15),
                ("exp", "We"), ("copy", "out", "We")]


def _lowlist_energies(extensive, k):
    """Sc, St, CKE of a pair as both Low & List parts form them; `extensive` is the column the
    reference uses there: water mass in the efficiency, volume in the fragmentation"""
    return ([("pair", "sum", "Sc", extensive), ("pow", "Sc", 2 / 3),
             ("mul", "Sc", surface_factor(k)),
             ("pair", "min", "St", "radius"), ("mul", "St", 2), ("pow", "St", 2),
             ("pair", "max", "tmp", "radius"), ("mul", "tmp", 2), ("pow", "tmp", 2),
             ("add", "St", "tmp"), ("mul", "St", k.PI * k.sgm_w)]
            + _weber_like(extensive, k, k.rho_w / 2))


@dataclass(frozen=True)
class LowList1982Ec:
    def descriptor(self, k):
        return {"ec": EC_CODES["lowlist1982"], "ec_param": (0.0, surface_factor(k)),
                "needs_gk": True}

    def program(self, k):
        b_coeff = 2.61e6 / si.J**2 * si.m**2
        return ([("pair", "min", "ds", "radius"), ("mul", "ds", 2),
                 ("pair", "max", "dl", "radius"), ("mul", "dl", 2)]
                + _lowlist_energies("water mass", k)
                + [("copy", "dS", "St"), ("sub", "dS", "Sc"), ("copy", "Et", "CKE"),
                   ("add", "Et", "dS"), ("copy", "tmp2", "Et"), ("pow", "tmp2", 2),
                   ("mul", "tmp2", -1.0 * b_coeff * k.sgm_w), ("div", "tmp2", "Sc"),
                   ("copy", "out", "ds"), ("div", "out", "dl"), ("add", "out", 1.0),
                   ("pow", "out", -2.0), ("mul", "out", 0.778), ("exp", "tmp2"),
                   ("mul", "out", "tmp2"), ("call", "sdm_ll82_coalescence_check", "out", "dl",
                                            "#pairs")])


# ---- fragmentation functions: results nf (number of fragments) and fm (fragment mass) -----------
@dataclass(frozen=True)
class AlwaysN:
    n: float

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"frag": FRAG_CODES["always_n"], "frag_param": (float(self.n), 0.0)}

    def program(self, k):  # pylint: disable=unused-argument
        return [("fill", "nf", self.n), ("pair", "sum", "fm", "water mass"),
                ("div", "fm", self.n)]


@dataclass(frozen=True)
class ConstantMass:
    c: float

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"frag": FRAG_CODES["constant_mass"], "frag_param": (float(self.c), 0.0)}

    def program(self, k):  # pylint: disable=unused-argument
        return [("fill", "fm", self.c), ("pair", "sum", "nf", "water mass"),
                ("div", "nf", self.c)]


def _limits(part):
    """fragment volumes are limited from below by vmin and in number by nfmax
    (fragmentation_methods.py:76-95)"""
    return {"frag_vmin": float(part.vmin), "frag_nfmax": _nfmax(part.nfmax)}


def _tail(part):
    return (float(part.vmin), _nfmax(part.nfmax))


@dataclass(frozen=True)
class Exponential:
    scale: float
    vmin: float = 0.0
    nfmax: Optional[float] = None

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"frag": FRAG_CODES["exponential"], "frag_param": (float(self.scale), 0.0),
                **_limits(self)}

    def program(self, k):  # pylint: disable=unused-argument
        return [("pair", "sum", "x_plus_y", "volume"),
                ("call", "sdm_exp_fragmentation", "nf", float(self.scale), "fm", "x_plus_y",
                 "u01", "#pairs", *_tail(self), 1e-5),
                ("volume_to_mass", "fm")]


@dataclass(frozen=True)
class Gaussian:
    mu: float
    sigma: float
    vmin: float = 0.0
    nfmax: Optional[float] = None

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"frag": FRAG_CODES["gaussian"],
                "frag_param": (float(self.mu), float(self.sigma)), **_limits(self)}

    def program(self, k):
        return [("pair", "sum", "x_plus_y", "volume"),
                ("call", "sdm_gauss_fragmentation", "nf", float(self.mu), float(self.sigma),
                 "fm", "x_plus_y", "u01", "#pairs", *_tail(self),
                 (k.VEDDER_1987_A, k.VEDDER_1987_b)),
                ("volume_to_mass", "fm")]


@dataclass(frozen=True)
class Feingold1988:
    scale: float
    fragtol: float = 1e-3
    vmin: float = 0.0
    nfmax: Optional[float] = None

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"frag": FRAG_CODES["feingold1988"],
                "frag_param": (float(self.scale), float(self.fragtol)), **_limits(self)}

    def program(self, k):  # pylint: disable=unused-argument
        return [("pair", "sum", "x_plus_y", "volume"),
                ("call", "sdm_feingold1988_fragmentation", "nf", float(self.scale), "fm",
                 "x_plus_y", "u01", "#pairs", float(self.fragtol), *_tail(self)),
                ("volume_to_mass", "fm")]


@dataclass(frozen=True)
class SLAMS:
    vmin: float = 0.0
    nfmax: Optional[float] = None

    def descriptor(self, k):  # pylint: disable=unused-argument
        return {"frag": FRAG_CODES["slams"], **_limits(self)}

    def program(self, k):  # pylint: disable=unused-argument
        return [("pair", "sum", "x_plus_y", "volume"),
                ("call", "sdm_slams_fragmentation", "nf", "fm", "x_plus_y", "probs", "u01",
                 "#pairs", *_tail(self)),
                ("volume_to_mass", "fm")]


@dataclass(frozen=True)
class Straub2010Nf:
    vmin: float = 0.0
    nfmax: Optional[float] = None

    def descriptor(self, k):
        return {"frag": FRAG_CODES["straub2010"], "frag_param": (0.0, surface_factor(k)),
                "needs_gk": True, **_limits(self)}

    def program(self, k):
        zeroed = [("fill", name, 0) for name in ("Nr1", "Nr2", "Nr3", "Nr4", "Nrt")]
        return ([("pair", "max", "v_max", "volume"), ("pair", "sum", "x_plus_y", "volume"),
                 ("pair", "min", "ds", "radius"), ("mul", "ds", 2),
                 ("pair", "sum", "tmp", "volume"), ("copy", "Sc", "tmp"), ("pow", "Sc", 2 / 3),
                 ("mul", "Sc", surface_factor(k)),
                 ("pair", "distance", "tmp2", "fall velocity"), ("pow", "tmp2", 2),
                 ("pair", "multiply", "CKE", "volume"), ("divnz", "CKE", "tmp"),
                 ("mul", "CKE", "tmp2"), ("mul", "CKE", k.rho_w / 2),
                 ("copy", "We", "CKE"), ("divnz", "We", "Sc"), ("copy", "CW", "We"),
                 ("mul", "CW", "CKE"), ("div", "CW", si.uJ),
                 ("pair", "max", "gam", "radius"), ("pair", "min", "tmp", "radius"),
                 ("divnz", "gam", "tmp")]
                + zeroed
                + [("call", "sdm_straub_fragmentation", "nf", "CW", "gam", "ds", "fm", "v_max",
                    "x_plus_y", "u01", "#pairs", *_tail(self), "Nr1", "Nr2", "Nr3", "Nr4",
                    "Nrt", "d34", straub_consts(k)),
                   ("volume_to_mass", "fm")])


@dataclass(frozen=True)
class LowList1982Nf:
    vmin: float = 0.0
    nfmax: Optional[float] = None

    def descriptor(self, k):
        return {"frag": FRAG_CODES["lowlist1982"], "frag_param": (surface_factor(k), 0.0),
                "needs_gk": True, **_limits(self)}

    def program(self, k):
        return ([("pair", "min", "ds", "radius"), ("mul", "ds", 2),
                 ("pair", "max", "dl", "radius"), ("mul", "dl", 2),
                 ("pair", "sum", "dcoal", "volume"), ("div", "dcoal", k.PI / 6),
                 ("pow", "dcoal", 1 / 3)]
                + _lowlist_energies("volume", k)
                + [("copy", "We", "CKE"), ("copy", "W2", "CKE"), ("divnz", "We", "Sc"),
                   ("divnz", "W2", "St"), ("mul", "Rf", 0.0), ("mul", "Rs", 0.0),
                   ("mul", "Rd", 0.0), ("pair", "sum", "x_plus_y", "volume"),
                   ("call", "sdm_ll82_fragmentation", "nf", "CKE", "We", "W2", "St", "ds", "dl",
                    "dcoal", "fm", "x_plus_y", "u01", "#pairs", *_tail(self), "Rf", "Rs", "Rd",
                    1e-8, (k.CM, k.PI, k.VEDDER_1987_A, k.VEDDER_1987_b)),
                   ("volume_to_mass", "fm")])


# ---- the set-up -------------------------------------------------------------------------------
@dataclass
class CollisionSetup:  # pylint: disable=too-many-instance-attributes
    """options of `Collision` (collision.py:44-172) with their reference defaults"""
    kernel: object
    coalescence_efficiency: object = field(default_factory=ConstEc)
    breakup_efficiency: object = field(default_factory=lambda: ConstEb(0.0))
    fragmentation: object = field(default_factory=lambda: AlwaysN(1))
    breakup: bool = False
    adaptive: bool = True
    substeps: int = 1
    dt_range: Tuple[float, float] = (0.1 * si.second, 100.0 * si.second)
    croupier: str = "local"
    optimized_random: bool = False
    warn_overflows: bool = True
    handle_all_breakups: bool = False
    seed: int = const.default_random_seed
    max_multiplicity: int = MAX_MULTIPLICITY

    def __post_init__(self):
        if not (self.substeps == 1 or self.adaptive is False):
            raise ValueError("substeps > 1 only without adaptivity")
        if self.dt_range[0] <= 0:
            raise ValueError("dt_range must start above zero")
        if self.croupier not in ("local", "global"):
            raise ValueError(self.croupier)

    @classmethod
    def coalescence(cls, kernel, **options):
        """collisions always coalesce (`Coalescence`, collision.py:293-322)"""
        return cls(kernel=kernel, coalescence_efficiency=options.pop("coalescence_efficiency",
                                                                     ConstEc(1.0)),
                   breakup_efficiency=ConstEb(0.0), fragmentation=AlwaysN(1), breakup=False,
                   **options)

    @classmethod
    def collision(cls, kernel, coalescence_efficiency, breakup_efficiency, fragmentation,
                  **options):
        """coalescence, breakup or bounce per colliding pair (`Collision`)"""
        return cls(kernel=kernel, coalescence_efficiency=coalescence_efficiency,
                   breakup_efficiency=breakup_efficiency, fragmentation=fragmentation,
                   breakup=True, **options)

    @classmethod
    def breakup_only(cls, kernel, fragmentation, **options):
        """every collision breaks up (`Breakup`, collision.py:325-349)"""
        return cls(kernel=kernel, coalescence_efficiency=ConstEc(0.0),
                   breakup_efficiency=ConstEb(1.0), fragmentation=fragmentation, breakup=True,
                   **options)

    def clamped_dt_range(self, dt):
        """collision.py:115-116: the upper end never exceeds the time step"""
        lo, hi = self.dt_range
        hi = min(hi, dt)
        if lo > hi:
            raise ValueError("dt_range[0] exceeds the time step")
        return lo, hi

    def parts(self):
        parts = [self.kernel]
        if self.breakup:
            parts += [self.coalescence_efficiency, self.breakup_efficiency, self.fragmentation]
        return parts

    def descriptor(self, k):
        """merged device-side description of all parts"""
        merged = {"needs_gk": False}
        for part in self.parts():
            desc = dict(part.descriptor(k))
            merged["needs_gk"] = merged["needs_gk"] or desc.pop("needs_gk", False)
            merged.update(desc)
        return merged
