"""Plugging this package into an unmodified PySDM installation.

1. The backend.  PySDM picks its backend by object - `Builder(n_sd, backend=<instance>, ...)`
   (PySDM/builder.py:25-38) - insisting only that it is an instance of its own `BackendMethods`
   (PySDM/particulator.py:22).  `HIP` already carries every method, `Storage` and `Random` the
   front-end calls, with PySDM's names and signatures; what remains is the base class:

       from pysdm_amd.pysdm_plugin import install
       HIP = install()                       # also reachable as PySDM.backends.HIP afterwards
       builder = Builder(n_sd, backend=HIP(Formulae(...)), environment=Box(...))

   PySDM's own `Coalescence` / `Collision` / `Breakup` then run stage by stage on the device.

2. The fused step.  `fuse(dynamic)` wraps one of PySDM's collision dynamics so that, registered
   with PySDM's Builder in its place, each call is ONE `sdm_collision_step` on PySDM's own
   attribute arrays (no copies: the particle state is adopted where it lies):

       builder.add_dynamic(fuse(Coalescence(collision_kernel=Golovin(b=1.5e3))))

   The dynamic's parts are translated to this package's `CollisionSetup` by class name and
   parameters (`setup_from_pysdm`); counters (`collision_rate`, ...) stay attributes of the
   returned object, as PySDM's products expect.

PySDM itself is imported lazily: this module loads (and fails loudly) only where PySDM exists.
PySDM's ParticleAttributes keeps the permutation index, `cell_start`, the sorted flag and the
number of valid super-droplets as name-mangled members (PySDM/impl/particle_attributes.py:13-46);
that is how PySDM's `Particulator` itself reaches them (particulator.py:301-313) and how the
fused step hands the state back.
"""
import copy
import importlib

from . import recipe as R
from .collisions import CollisionRunner
from .population import Population

_PRIVATE = "_ParticleAttributes__"


def _pysdm_backend_methods():
    try:
        module = importlib.import_module("PySDM.backends.impl_common.backend_methods")
    except ImportError as error:
        raise ImportError("pysdm_amd.pysdm_plugin needs an importable PySDM") from error
    return module.BackendMethods


def as_pysdm_backend(backend_class):
    """`backend_class` (HIP; the tests pass its CPU-oracle twin) as a class PySDM's Particulator
    accepts"""
    base = _pysdm_backend_methods()
    if issubclass(backend_class, base):
        return backend_class

    def __init__(self, formulae=None, double_precision=True, **options):
        # next to PySDM the default formulae are PySDM's (numba.py:44: `formulae or Formulae()`):
        # its attributes build their physics from members this package's own, smaller Formulae
        # does not have (formulae.terminal_velocity_class, attributes/physics/terminal_velocity.py:19)
        if formulae is None:
            formulae = importlib.import_module("PySDM.formulae").Formulae()
        backend_class.__init__(self, formulae, double_precision, **options)

    return type(backend_class.__name__, (backend_class, base), {
        "__init__": __init__, "__doc__": backend_class.__doc__,
        "__module__": backend_class.__module__})


def install():
    """registers `PySDM.backends.HIP` next to CPU / GPU (PySDM/backends/__init__.py:75-83)"""
    from .backends.hip import HIP  # pylint: disable=import-outside-toplevel

    plugged = as_pysdm_backend(HIP)
    setattr(importlib.import_module("PySDM.backends"), "HIP", plugged)
    return plugged


# ---- PySDM's collision parts -> this package's recipe ---------------------------------------------
def _part(obj):
    """a PySDM kernel / efficiency / fragmentation object as the recipe part of the same name"""
    kind = type(obj).__name__
    limits = {"vmin": getattr(obj, "vmin", 0.0), "nfmax": getattr(obj, "nfmax", None)}
    table = {
        "Golovin": lambda: R.Golovin(b=obj.b),
        "Geometric": lambda: R.Geometric(collection_efficiency=obj.collection_efficiency),
        "ConstantK": lambda: R.ConstantK(a=obj.a),
        "Electric": R.Electric, "Hydrodynamic": R.Hydrodynamic,
        "SimpleGeometric": lambda: R.SimpleGeometric(C=obj.C),
        "ConstEc": lambda: R.ConstEc(Ec=obj.Ec), "ConstEb": lambda: R.ConstEb(Eb=obj.Eb),
        "Berry1967": R.Berry1967,
        "SpecifiedEff": lambda: R.SpecifiedEff(params=tuple(obj.params)),
        "Straub2010Ec": R.Straub2010Ec, "LowList1982Ec": R.LowList1982Ec,
        "AlwaysN": lambda: R.AlwaysN(n=obj.N), "ConstantMass": lambda: R.ConstantMass(c=obj.C),
        "Exponential": lambda: R.Exponential(scale=obj.scale, **limits),
        "Gaussian": lambda: R.Gaussian(mu=obj.mu, sigma=obj.sigma, **limits),
        "Feingold1988": lambda: R.Feingold1988(scale=obj.scale, fragtol=obj.fragtol, **limits),
        "SLAMS": lambda: R.SLAMS(**limits),
        "Straub2010Nf": lambda: R.Straub2010Nf(**limits),
        "LowList1982Nf": lambda: R.LowList1982Nf(**limits),
    }
    if kind not in table:
        raise NotImplementedError(f"{kind} has no device-side description; use PySDM's own "
                                  "dynamic with the plugged backend (stage-by-stage route)")
    return table[kind]()


def setup_from_pysdm(dynamic, formulae):
    """`CollisionSetup` equivalent of a PySDM `Collision` / `Coalescence` / `Breakup` object"""
    return R.CollisionSetup(
        kernel=_part(dynamic.collision_kernel),
        coalescence_efficiency=_part(dynamic.compute_coalescence_efficiency),
        breakup_efficiency=_part(dynamic.compute_breakup_efficiency),
        fragmentation=_part(dynamic.compute_number_of_fragments),
        breakup=bool(dynamic.enable_breakup), adaptive=bool(dynamic.adaptive),
        substeps=int(getattr(dynamic, "_Collision__substeps", 1)), dt_range=tuple(dynamic.dt_coal_range),
        croupier=dynamic.croupier or "local", optimized_random=bool(dynamic.optimized_random),
        warn_overflows=bool(dynamic.warn_overflows),
        handle_all_breakups=bool(formulae.handle_all_breakups), seed=int(formulae.seed),
        max_multiplicity=int(dynamic.max_multiplicity))


class _AdoptedState:
    """PySDM's ParticleAttributes seen as a Population (arrays shared, bookkeeping copied in
    before and back after every fused call)"""

    def __init__(self, particulator):
        self.attributes = attrs = particulator.attributes
        self.idx = getattr(attrs, _PRIVATE + "idx")
        self.caretaker = getattr(attrs, _PRIVATE + "cell_caretaker")
        keys = list(attrs.get_extensive_attribute_keys())
        self.population = Population.adopt(
            particulator.backend.engine, perm=self.idx.data, perm_spare=self.caretaker.tmp_idx.data,
            multiplicity=attrs["multiplicity"].data,
            extensive=attrs.get_extensive_attribute_storage().data,
            rows={name: row for row, name in enumerate(keys)}, cell_id=attrs["cell id"].data,
            cell_order=attrs.cell_idx.data, cell_start=getattr(attrs, _PRIVATE + "cell_start").data,
            live=getattr(attrs, _PRIVATE + "valid_n_sd"), ordered=getattr(attrs, _PRIVATE + "sorted"),
            rho_w=particulator.formulae.constants.rho_w)
        self.stamps = None

    def _stamps(self):
        members = getattr(self.attributes, _PRIVATE + "attributes")
        names = ["multiplicity", "cell id"] + list(self.attributes.get_extensive_attribute_keys())
        return tuple(members[name].timestamp for name in names)

    def before(self):
        attrs, pop = self.attributes, self.population
        attrs.sanitize()
        stamps = self._stamps()
        if stamps != self.stamps:  # someone else touched the state since the last fused call
            pop.perm, pop.perm_spare = self.idx.data, self.caretaker.tmp_idx.data
            pop.live = pop.working = len(self.idx)
            pop.ordered = bool(getattr(attrs, _PRIVATE + "sorted"))
            if self.stamps is None or stamps[0] != self.stamps[0] or stamps[2:] != self.stamps[2:]:
                pop.touch_state()
            pop.host_dirty = True

    def after(self):
        attrs, pop = self.attributes, self.population
        self.idx.data, self.caretaker.tmp_idx.data = pop.perm, pop.perm_spare
        setattr(attrs, _PRIVATE + "valid_n_sd", int(pop.live))
        self.idx.length = self.idx.INT(int(pop.live))
        setattr(attrs, _PRIVATE + "sorted", bool(pop.ordered))
        attrs.mark_updated("multiplicity")
        for key in attrs.get_extensive_attribute_keys():
            attrs.mark_updated(key)
        self.stamps = self._stamps()


class Collision:  # pylint: disable=too-few-public-methods
    """root class: PySDM's Builder files a dynamic under the name of the class right below
    `object` in its MRO (builder.py:55) - the fused dynamic takes PySDM's "Collision" slot"""


class FusedCollision(Collision):
    """a PySDM dynamic (register / instantiate / __call__ protocol of PySDM's Builder) running
    the wrapped PySDM collision dynamic's configuration as the fused step"""

    _OWN = ("inner", "particulator", "runner", "_state")

    def __init__(self, dynamic):
        self.inner = dynamic
        self.particulator = None
        self.runner = None
        self._state = None

    def __setattr__(self, name, value):
        # options belong to the wrapped dynamic: PySDM's SpinUp observer switches collisions off
        # with setattr(particulator.dynamics["Collision"], "enable", False)
        # (examples/PySDM_examples/Arabas_et_al_2015/spin_up.py), and that lands here
        if name in self._OWN or "inner" not in self.__dict__:
            object.__setattr__(self, name, value)
        else:
            setattr(self.__dict__["inner"], name, value)

    def register(self, builder):
        self.particulator = builder.particulator
        self.inner.register(builder)  # requests the attributes the parts need

    def instantiate(self, *, builder):
        own = copy.copy(self)
        own.inner = copy.deepcopy(self.inner)
        own.register(builder)
        return own

    def __getattr__(self, name):
        # diagnostics live on the runner once it exists; options on the wrapped dynamic
        if name.startswith("__") or "inner" not in self.__dict__:
            raise AttributeError(name)
        runner = self.__dict__.get("runner")
        if runner is not None and name in ("collision_rate", "collision_rate_deficit",
                                           "coalescence_rate", "breakup_rate",
                                           "breakup_rate_deficit", "stats_n_substep",
                                           "stats_dt_min"):
            storage = self.particulator.backend.Storage
            array = getattr(runner, name)
            return storage(array, tuple(array.shape), storage.INT if "dt_min" not in name
                           else storage.FLOAT)
        return getattr(self.__dict__["inner"], name)

    def __call__(self):
        if not self.inner.enable:  # collision.py:175
            return
        part = self.particulator
        if self.runner is None:
            self._state = _AdoptedState(part)
            setup = setup_from_pysdm(self.inner, part.formulae)
            self.runner = CollisionRunner(self._state.population, setup, dt=part.dt,
                                          dv=part.mesh.dv, route="fused",
                                          constants=part.formulae.constants)
        self._state.before()
        self.runner.run(1)
        self._state.after()


def fuse(dynamic):
    """`dynamic`: a PySDM Collision / Coalescence / Breakup instance"""
    return FusedCollision(dynamic)
