"""Particle state on the collision path: attribute objects with timestamp-based lazy refresh and
the `ParticleAttributes` manager (permutation index, cell bookkeeping, health flag).

Host-side mirror of PySDM/attributes/impl/{attribute,base_attribute,derived_attribute,
extensive_attribute}.py, PySDM/attributes/physics/{multiplicity,water_mass,volume,radius,
terminal_velocity}.py, PySDM/attributes/numerics/cell_id.py and
PySDM/impl/particle_attributes.py:13-125 / particle_attributes_factory.py:20-121.
"""
import numpy as np


# ---- attribute objects ---------------------------------------------------------------------
class Attribute:
    def __init__(self, builder, name, dtype=float, n_vector_components=0):
        self.particulator = builder.particulator
        self.timestamp = 0
        self.data = None
        self.dtype = dtype
        self.n_vector_components = n_vector_components
        self.name = name
        self.formulae = self.particulator.formulae

    def allocate(self, idx):
        n_sd = self.particulator.n_sd
        shape = (self.n_vector_components, n_sd) if self.n_vector_components >= 1 else (n_sd,)
        self.data = self.particulator.IndexedStorage.empty(idx, shape, dtype=self.dtype)

    def set_data(self, data):
        self.data = data

    def get(self):
        self.update()
        return self.data

    def update(self):
        pass

    def mark_updated(self):
        self.timestamp += 1

    def __str__(self):
        return self.name


class BaseAttribute(Attribute):
    def init(self, data):
        self.data.upload(data)
        self.mark_updated()


class ExtensiveAttribute(BaseAttribute):
    pass


class CellAttribute(BaseAttribute):
    pass


class DerivedAttribute(Attribute):
    def __init__(self, builder, name, dependencies):
        assert len(dependencies) > 0
        super().__init__(builder, name)
        self.dependencies = dependencies

    def update(self):
        stamp = 0
        for dependency in self.dependencies:
            dependency.update()
            stamp += dependency.timestamp
        if self.timestamp < stamp:
            self.timestamp = stamp
            self.recalculate()

    def recalculate(self):
        raise NotImplementedError()

    def mark_updated(self):
        raise AssertionError()


class Multiplicity(BaseAttribute):
    TYPE = np.int64
    MAX_VALUE = np.iinfo(np.int64).max

    def __init__(self, builder):
        super().__init__(builder, name="multiplicity", dtype=Multiplicity.TYPE)


class CellId(CellAttribute):
    def __init__(self, builder):
        super().__init__(builder, name="cell id", dtype=np.int64)


class CellOrigin(CellAttribute):
    def __init__(self, builder):
        super().__init__(
            builder, name="cell origin", dtype=np.int64,
            n_vector_components=builder.particulator.mesh.dim,
        )


class PositionInCell(CellAttribute):  # PySDM/attributes/numerics/position_in_cell.py
    def __init__(self, builder):
        super().__init__(
            builder, name="position in cell", dtype=float,
            n_vector_components=builder.particulator.mesh.dim,
        )


class SignedWaterMass(ExtensiveAttribute):
    def __init__(self, builder):
        super().__init__(builder, name="signed water mass")


class WaterMass(DerivedAttribute):
    """a view on the signed water mass (liquid-only particles)"""

    def __init__(self, builder):
        self.signed_water_mass = builder.get_attribute("signed water mass")
        super().__init__(builder, name="water mass", dependencies=(self.signed_water_mass,))

    def mark_updated(self):
        self.signed_water_mass.mark_updated()

    def allocate(self, idx):
        pass

    def recalculate(self):
        pass

    def get(self):
        return self.signed_water_mass.data


class Volume(DerivedAttribute):
    def __init__(self, builder):
        self.water_mass = builder.get_attribute("water mass")
        super().__init__(builder, name="volume", dependencies=(self.water_mass,))

    def recalculate(self):
        self.particulator.backend.volume_of_water_mass(self.data, self.water_mass.get())


class Radius(DerivedAttribute):
    def __init__(self, builder):
        self.volume = builder.get_attribute("volume")
        super().__init__(builder, name="radius", dependencies=(self.volume,))

    def recalculate(self):
        self.data.product(self.volume.get(), 1 / self.formulae.constants.PI_4_3)
        self.data **= 1 / 3


class Area(DerivedAttribute):  # cf. PySDM/attributes/physics/area.py
    def __init__(self, builder):
        self.volume = builder.get_attribute("volume")
        super().__init__(builder, name="area", dependencies=(self.volume,))

    def recalculate(self):
        self.data.product(self.volume.get(), 1 / self.formulae.constants.PI_4_3)
        self.data **= 2 / 3
        self.data *= self.formulae.constants.PI_4_3 * 3


class TerminalVelocity(DerivedAttribute):
    def __init__(self, builder, name="terminal velocity"):
        self.radius = builder.get_attribute("radius")
        super().__init__(builder, name=name, dependencies=(self.radius,))
        self.approximation = builder.formulae.terminal_velocity_class(builder.particulator)

    def recalculate(self):
        self.approximation(self.data, self.radius.get())


ATTRIBUTE_CLASSES = {
    "multiplicity": Multiplicity,
    "cell id": CellId,
    "cell origin": CellOrigin,
    "position in cell": PositionInCell,
    "signed water mass": SignedWaterMass,
    "water mass": WaterMass,
    "volume": Volume,
    "radius": Radius,
    "area": Area,
    "terminal velocity": TerminalVelocity,
    # no RelaxedVelocity dynamic on this path: the fall velocity IS the terminal velocity
    "relative fall velocity": lambda builder: TerminalVelocity(builder, "relative fall velocity"),
}


def get_attribute_class(name):
    return ATTRIBUTE_CLASSES[name]


# ---- the manager ---------------------------------------------------------------------------
class ParticleAttributes:  # pylint: disable=too-many-instance-attributes
    def __init__(self, *, particulator, idx, extensive_attribute_storage, extensive_keys,
                 cell_start, attributes):
        self.__valid_n_sd = particulator.n_sd
        self.__healthy_memory = particulator.Storage.from_ndarray(np.full((1,), 1))
        self.__idx = idx
        self.__extensive_attribute_storage = extensive_attribute_storage
        self.__extensive_keys = extensive_keys
        self.cell_idx = particulator.Index.identity_index(len(cell_start) - 1)
        self.__cell_start = particulator.Storage.from_ndarray(cell_start)
        self.__cell_caretaker = particulator.backend.make_cell_caretaker(
            self.__idx.shape, self.__idx.dtype, len(self.__cell_start),
            scheme=particulator.sorting_scheme,
        )
        self.__sorted = False
        self.__attributes = attributes

    @property
    def healthy(self) -> bool:
        return bool(self.__healthy_memory[0])

    @healthy.setter
    def healthy(self, value: bool):
        self.__healthy_memory[:] = value

    @property
    def cell_start(self):
        if not self.__sorted:
            self.__sort_by_cell_id()
        return self.__cell_start

    @property
    def super_droplet_count(self):
        assert self.healthy
        return len(self.__idx)

    def mark_updated(self, key):
        self.__attributes[key].mark_updated()

    def sanitize(self):
        if not self.healthy:
            self.__idx.length = self.__valid_n_sd
            self.__idx.remove_zero_n_or_flagged(self["multiplicity"])
            self.__valid_n_sd = self.__idx.length
            self.healthy = True
            self.__sorted = False

    def cut_working_length(self, length):
        assert length <= len(self.__idx)
        self.__idx.length = length

    def get_working_length(self):
        return len(self.__idx)

    def reset_working_length(self):
        self.__idx.length = self.__valid_n_sd

    def reset_cell_idx(self):
        self.cell_idx.reset_index()
        self.__sort_by_cell_id()

    def keys(self):
        return self.__attributes.keys()

    def __getitem__(self, item):
        return self.__attributes[item].get()

    def __contains__(self, key):
        return key in self.__attributes

    def has_attribute(self, attr):
        return attr in self.__attributes

    def get_attribute_object(self, name):
        return self.__attributes[name]

    def permutation(self, u01, local):
        if local:
            self.__idx.shuffle(u01, parts=self.cell_start)
        else:
            self.__idx.shuffle(u01)
            self.__sorted = False

    def __sort_by_cell_id(self):
        self.__cell_caretaker(self["cell id"], self.cell_idx, self.__cell_start, self.__idx)
        self.__sorted = True

    def get_extensive_attribute_storage(self):
        return self.__extensive_attribute_storage

    def get_extensive_attribute_keys(self):
        return self.__extensive_keys.keys()

    def reset_idx(self):
        self.__valid_n_sd = self.__idx.shape[0]
        self.__idx.reset_index()
        self.healthy = False

    # ---- hooks for the fused per-time-step path (state handed over wholesale) ---------------
    def _fused_view(self):
        return {
            "idx": self.__idx,
            "cell_start": self.__cell_start,
            "healthy": self.__healthy_memory,
            "sorted": self.__sorted,
            "valid_n_sd": self.__valid_n_sd,
            "caretaker": self.__cell_caretaker,
        }

    def _fused_commit(self, *, valid_n_sd, sorted_flag):
        self.__valid_n_sd = int(valid_n_sd)
        self.__idx.length = self.__idx.INT(int(valid_n_sd))
        self.__sorted = bool(sorted_flag)


def make_particle_attributes(particulator, req_attr, attributes):
    """allocates the SoA columns (all extensive attributes share one (A, n_sd) block) and wires
    them to one shared permutation index"""
    idx = particulator.Index.identity_index(particulator.n_sd)
    extensive = [name for name, a in req_attr.items() if isinstance(a, ExtensiveAttribute)]
    block = particulator.IndexedStorage.empty(idx, (len(extensive), particulator.n_sd), float)
    for attr in req_attr.values():
        if isinstance(attr, DerivedAttribute):
            if attr.name in attributes:
                raise ValueError(
                    f"attribute '{attr.name}' is a dummy/derived one, but values were provided"
                )
            attr.allocate(idx)
    extensive_keys = {}
    for row, name in enumerate(extensive):
        extensive_keys[name] = row
        req_attr[name].set_data(block[row, :])
        if name not in attributes:
            raise ValueError(
                f"attribute '{name}' requested by one of the components"
                f" but no initial values given"
            )
        req_attr[name].init(attributes[name])
    for name in ("multiplicity", "cell id", "cell origin", "position in cell"):
        if name not in req_attr:
            continue
        attr = req_attr[name]
        attr.allocate(idx)
        attr.init(attributes[name])
        attr.data = particulator.IndexedStorage.indexed(idx, attr.data)
    cell_start = np.empty(particulator.mesh.n_cell + 1, dtype=np.int64)
    return ParticleAttributes(
        particulator=particulator, idx=idx, extensive_attribute_storage=block,
        extensive_keys=extensive_keys, cell_start=cell_start, attributes=req_attr,
    )
