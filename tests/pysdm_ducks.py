"""Test doubles for the argument objects PySDM's front-end hands to a backend.

A PySDM-shaped backend method receives PySDM's own wrappers: the permutation `Index` (a Storage
with a live `len()`), `IndexedStorage` (a Storage carrying `.idx`), `PairIndicator` (`.indicator`
+ `len()`), `PairwiseStorage` (a Storage).  PySDM does not travel to the GPU box, so the tests
that call backend methods the way PySDM does pass these minimal stand-ins instead: data holders
with exactly the attributes the backend reads, plus a few conveniences for the tests themselves.
(Under a real PySDM the real wrappers are used: tests/test_reference_plugin.py.)"""
import numpy as np


def make(backend):
    Storage = backend.Storage

    class Index(Storage):
        def __init__(self, data, shape=None, dtype=None):
            super().__init__(data, shape, dtype)
            self.length = int(self.shape[0])

        def __len__(self):
            return int(self.length)

        @classmethod
        def identity_index(cls, n):
            return cls.from_ndarray(np.arange(n, dtype=np.int64))

    class IndexedStorage(Storage):
        idx = None

        @classmethod
        def from_ndarray(cls, idx, array):  # pylint: disable=arguments-differ
            made = super().from_ndarray(array)
            made.idx = idx
            return made

        @classmethod
        def empty(cls, idx, shape, dtype):  # pylint: disable=arguments-differ
            made = super().empty(shape, dtype)
            made.idx = idx
            return made

        def __len__(self):
            return len(self.idx)

        def row(self, row):
            """one row of a (rows, n_sd) storage, still indexed"""
            view = IndexedStorage(self.data[row], self.shape[1:], self.dtype)
            view.idx = self.idx
            return view

        def to_ndarray(self, raw=False):  # pylint: disable=arguments-differ
            host = super().to_ndarray()
            return host if raw else host[..., self.idx.to_ndarray()[:len(self.idx)]]

    class PairIndicator:
        def __init__(self, length):
            self.indicator = Storage.empty(length, dtype=bool)
            self.length = length

        def __len__(self):
            return self.length

    class PairwiseStorage(Storage):
        pass

    return Index, IndexedStorage, PairIndicator, PairwiseStorage


# ---- a duck Particulator for pysdm_amd.pysdm_plugin.fuse (no PySDM on the GPU box) -----------------
# Only what `FusedCollision` / `_AdoptedState` touch: PySDM's ParticleAttributes keeps the
# permutation, cell_start, the sorted flag and the valid length as name-mangled members
# (PySDM/impl/particle_attributes.py:13-46) - the class below is therefore NAMED ParticleAttributes,
# so that Python mangles its `__members` to the very names the plug-in reads and writes.
class _Attr:  # pylint: disable=too-few-public-methods
    def __init__(self, data):
        self.data, self.timestamp = data, 0


class _Caretaker:  # pylint: disable=too-few-public-methods
    def __init__(self, tmp_idx):
        self.tmp_idx = tmp_idx


class ParticleAttributes:  # pylint: disable=too-many-instance-attributes
    def __init__(self, backend, multiplicity, mass, cell_id, n_cell):
        index_class = make(backend)[0]
        storage = backend.Storage
        n_sd = len(multiplicity)
        self.__idx = index_class.identity_index(n_sd)
        self.__cell_caretaker = _Caretaker(index_class.identity_index(n_sd))
        self.__cell_start = storage.from_ndarray(np.zeros(n_cell + 1, dtype=np.int64))
        self.__valid_n_sd = n_sd
        self.__sorted = False
        self.cell_idx = index_class.identity_index(n_cell)
        self.__extensive = storage.from_ndarray(np.asarray(mass, dtype=float).reshape(1, -1))
        self.__attributes = {
            "multiplicity": _Attr(storage.from_ndarray(np.asarray(multiplicity, dtype=np.int64)).data),
            "cell id": _Attr(storage.from_ndarray(np.asarray(cell_id, dtype=np.int64)).data),
            "signed water mass": _Attr(self.__extensive.data[0]),
        }
        self.sanitized = 0

    def __getitem__(self, name):
        return self.__attributes[name]

    @staticmethod
    def get_extensive_attribute_keys():
        return ("signed water mass",)

    def get_extensive_attribute_storage(self):
        return self.__extensive

    def sanitize(self):
        self.sanitized += 1

    def mark_updated(self, name):
        self.__attributes[name].timestamp += 1

    # what a test reads back
    def state(self, engine):
        length = int(self.__valid_n_sd)
        assert len(self.__idx) == length
        return {"length": length, "idx": engine.download(self.__idx.data),
                "multiplicity": engine.download(self.__attributes["multiplicity"].data),
                "attributes": engine.download(self.__extensive.data),
                "cell_start": engine.download(self.__cell_start.data), "sorted": self.__sorted}


class _Namespace:  # pylint: disable=too-few-public-methods
    def __init__(self, **members):
        self.__dict__.update(members)


class Particulator:  # pylint: disable=too-few-public-methods
    """`.backend`, `.formulae`, `.attributes`, `.dt`, `.mesh.dv`: what the fused dynamic reads"""

    def __init__(self, backend, *, multiplicity, mass, cell_id, n_cell, dt, dv, seed,
                 handle_all_breakups=False, constants=None):
        from pysdm_amd.physics import constants as const  # pylint: disable=import-outside-toplevel

        self.backend = backend
        self.formulae = _Namespace(constants=constants or const.namespace(), seed=seed,
                                   handle_all_breakups=handle_all_breakups)
        self.attributes = ParticleAttributes(backend, multiplicity, mass, cell_id, n_cell)
        self.dt, self.mesh = dt, _Namespace(dv=dv)
        self.particulator = self  # a Builder, as far as `register` is concerned


def named(name, **members):
    """an object whose class is called `name` (PySDM's parts are recognised by class name)"""
    obj = type(name, (), {})()
    obj.__dict__.update(members)
    return obj


def collision_dynamic(kind, **parts):
    """stand-in for a PySDM Coalescence / Collision object: the option names of
    PySDM/dynamics/collisions/collision.py:57-128 (defaults: :24-28), parts recognised by class
    name, as `pysdm_plugin.setup_from_pysdm` reads them"""
    from pysdm_amd.recipe import MAX_MULTIPLICITY  # pylint: disable=import-outside-toplevel

    options = dict(
        collision_kernel=named("Golovin", b=1.5e3),
        compute_coalescence_efficiency=named("ConstEc", Ec=1.0),
        compute_breakup_efficiency=named("ConstEb", Eb=0.0),
        compute_number_of_fragments=named("AlwaysN", N=1),
        enable_breakup=False, adaptive=True, dt_coal_range=(0.1, 100.0), croupier=None,
        optimized_random=False, warn_overflows=True, max_multiplicity=MAX_MULTIPLICITY,
        enable=True)
    options.update(parts)
    dynamic = type(kind, (), {"register": lambda self, builder: None})()
    dynamic.__dict__.update(options)
    return dynamic


def fused_plugin_run(name, backend_class):
    """`pysdm_plugin.fuse` driven the way PySDM's Builder / Particulator drive a dynamic
    (instantiate -> __call__ per time step) over the duck Particulator, on golden `name`:
    every recorded step must equal the reference's golden"""
    import os  # pylint: disable=import-outside-toplevel

    from pysdm_amd.physics import constants as const  # pylint: disable=import-outside-toplevel
    from pysdm_amd.population import to_integer_multiplicities  # pylint: disable=import-outside-toplevel
    from pysdm_amd.pysdm_plugin import fuse  # pylint: disable=import-outside-toplevel

    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                name + ".npz"))
    cfg = gold["cfg"]
    n_sd, seed, adaptive, dt, dv = int(cfg[0]), int(cfg[1]), bool(cfg[2]), cfg[3], cfg[4]
    if name.startswith("traj_multicell"):
        n_cell = int(np.prod(gold["grid"]))
        cell_id = gold["init/cell_id"]
        kernel = (named("Golovin", b=1.5e3) if "golovin" in name
                  else named("Geometric", collection_efficiency=1))
        options = {"optimized_random": bool(cfg[6])}
    else:
        n_cell, cell_id, kernel, options = 1, np.zeros(n_sd, dtype=np.int64), named(
            "Golovin", b=cfg[5]), {}
    backend = backend_class()
    part = Particulator(backend, multiplicity=to_integer_multiplicities(gold["init/multiplicity"]),
                        mass=const.rho_w * gold["init/volume"], cell_id=cell_id, n_cell=n_cell,
                        dt=dt, dv=dv, seed=seed)
    template = fuse(collision_dynamic("Coalescence", collision_kernel=kernel, adaptive=adaptive,
                                      **options))
    dynamic = template.instantiate(builder=part)  # builder.py:55-63
    assert dynamic is not template and dynamic.inner is not template.inner
    steps = sorted({int(k.split("/")[0][4:]) for k in gold.files if k.startswith("step")})
    done = 0
    for step in steps:
        while done < step:
            dynamic()
            done += 1
        state = part.attributes.state(backend.engine)
        length = state["length"]
        assert length == int(gold[f"step{step}/length"])
        np.testing.assert_array_equal(state["idx"][:length], gold[f"step{step}/idx"][:length])
        for key in ("multiplicity", "attributes", "cell_start"):
            np.testing.assert_array_equal(state[key], gold[f"step{step}/{key}"], err_msg=key)
        for key in ("collision_rate", "collision_rate_deficit", "coalescence_rate",
                    "stats_n_substep"):
            np.testing.assert_array_equal(getattr(dynamic, key).to_ndarray(),
                                          gold[f"step{step}/{key}"], err_msg=key)
    # spin-up protocol (Arabas_et_al_2015/spin_up.py): switched off, a call changes nothing
    setattr(dynamic, "enable", False)
    before = part.attributes.state(backend.engine)
    dynamic()
    after = part.attributes.state(backend.engine)
    for key, value in before.items():
        np.testing.assert_array_equal(after[key], value, err_msg=key)
    assert part.attributes.sanitized == done
    return dynamic
