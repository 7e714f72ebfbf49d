"""`Particulator` (the object dynamics talk to) and `Builder` (wiring), collision-path subset.

Mirrors PySDM/particulator.py:20-96,157-213,298-399 and PySDM/builder.py:27-164: same attribute and
method names, so the `Collision` dynamic and the parity tests read like the reference's.
"""
import time
import warnings

import numpy as np

from .attributes import get_attribute_class, make_particle_attributes
from .backends.impl_common import (
    BackendMethods,
    make_Index,
    make_IndexedStorage,
    make_PairIndicator,
    make_PairwiseStorage,
)
from .formulae import Formulae
from .initialisation import discretise_multiplicities


class WallTimer:  # cf. PySDM/impl/wall_timer.py:9-22
    def __init__(self):
        self.time = None
        self.t0 = None

    def __enter__(self):
        self.t0 = time.perf_counter()

    def __exit__(self, *_):
        self.time = time.perf_counter() - self.t0


class Particulator:  # pylint: disable=too-many-instance-attributes
    def __init__(self, n_sd, backend):
        assert isinstance(backend, BackendMethods)
        self.__n_sd = n_sd
        self.backend = backend
        self.formulae = backend.formulae
        self.environment = None
        self.attributes = None
        self.dynamics = {}
        self.products = {}
        self.observers = []
        self.n_steps = 0
        self.sorting_scheme = "default"
        self.Index = make_Index(backend)  # pylint: disable=invalid-name
        self.PairIndicator = make_PairIndicator(backend)  # pylint: disable=invalid-name
        self.PairwiseStorage = make_PairwiseStorage(backend)  # pylint: disable=invalid-name
        self.IndexedStorage = make_IndexedStorage(backend)  # pylint: disable=invalid-name
        self.timers = {}
        self.null = self.Storage.empty(0, dtype=float)

    def run(self, steps):
        # a collision dynamic on its own with nobody observing the intermediate states: hand the
        # whole loop to the backend (same sequence of time steps, no Python in between)
        if steps > 1 and len(self.dynamics) == 1 and not self.observers:
            (key, dynamic), = self.dynamics.items()
            if getattr(dynamic, "run_steps", None) is not None and dynamic.run_steps(steps):
                self.n_steps += steps
                return
        for _ in range(steps):
            for key, dynamic in self.dynamics.items():
                with self.timers[key]:
                    dynamic()
            self.n_steps += 1
            for observer in reversed(self.observers):
                observer.notify()

    @property
    def Storage(self):  # pylint: disable=invalid-name
        return self.backend.Storage

    @property
    def Random(self):  # pylint: disable=invalid-name
        return self.backend.Random

    @property
    def n_sd(self) -> int:
        return self.__n_sd

    @property
    def dt(self):
        return None if self.environment is None else self.environment.dt

    @property
    def mesh(self):
        return None if self.environment is None else self.environment.mesh

    def normalize(self, prob, norm_factor):
        self.backend.normalize(
            prob=prob,
            cell_id=self.attributes["cell id"],
            cell_idx=self.attributes.cell_idx,
            cell_start=self.attributes.cell_start,
            norm_factor=norm_factor,
            timestep=self.dt,
            dv=self.mesh.dv,
        )

    def collision_coalescence_breakup(self, *, enable_breakup, gamma, rand, Ec, Eb, fragment_mass,
                                      coalescence_rate, breakup_rate, breakup_rate_deficit,
                                      is_first_in_pair, warn_overflows, max_multiplicity):
        view = self.attributes._fused_view()  # pylint: disable=protected-access
        common = {
            "multiplicity": self.attributes["multiplicity"],
            "idx": view["idx"],
            "attributes": self.attributes.get_extensive_attribute_storage(),
            "gamma": gamma,
            "healthy": view["healthy"],
            "cell_id": self.attributes["cell id"],
            "coalescence_rate": coalescence_rate,
            "is_first_in_pair": is_first_in_pair,
        }
        if enable_breakup:
            self.backend.collision_coalescence_breakup(
                **common, rand=rand, Ec=Ec, Eb=Eb, fragment_mass=fragment_mass,
                breakup_rate=breakup_rate, breakup_rate_deficit=breakup_rate_deficit,
                warn_overflows=warn_overflows, particle_mass=self.attributes["water mass"],
                max_multiplicity=max_multiplicity,
            )
        else:
            self.backend.collision_coalescence(**common)
        self.attributes.sanitize()
        self.mark_collision_outputs_updated()

    def mark_collision_outputs_updated(self):
        self.attributes.mark_updated("multiplicity")
        for key in self.attributes.get_extensive_attribute_keys():
            self.attributes.mark_updated(key)

    def recalculate_cell_id(self):
        if not self.attributes.has_attribute("cell origin"):
            return
        self.backend.cell_id(
            self.attributes["cell id"],
            self.attributes["cell origin"],
            self.backend.Storage.from_ndarray(np.asarray(self.environment.mesh.strides)),
        )
        self.attributes._ParticleAttributes__sorted = False  # pylint: disable=protected-access

    def sort_within_pair_by_attr(self, is_first_in_pair, attr_name):
        self.backend.sort_within_pair_by_attr(
            self.attributes._fused_view()["idx"],  # pylint: disable=protected-access
            is_first_in_pair,
            self.attributes[attr_name],
        )

    def moments(self, *, moment_0, moments, specs: dict, attr_name="signed water mass",
                attr_range=(-np.inf, np.inf), weighting_attribute="water mass",
                weighting_rank=0, skip_division_by_m0=False):
        if len(specs) == 0:
            raise ValueError("empty specs passed")
        attr_data, ranks = [], []
        for attr, attr_ranks in specs.items():
            for rank in attr_ranks:
                attr_data.append(self.attributes[attr])
                ranks.append(rank)
        assert len(set(attr_data)) <= 1
        self.backend.moments(
            moment_0=moment_0,
            moments=moments,
            multiplicity=self.attributes["multiplicity"],
            attr_data=attr_data[0],
            cell_id=self.attributes["cell id"],
            idx=self.attributes._fused_view()["idx"],  # pylint: disable=protected-access
            length=self.attributes.super_droplet_count,
            ranks=self.backend.Storage.from_ndarray(np.array(ranks, dtype=float)),
            min_x=attr_range[0],
            max_x=attr_range[1],
            x_attr=self.attributes[attr_name],
            weighting_attribute=self.attributes[weighting_attribute],
            weighting_rank=weighting_rank,
            skip_division_by_m0=skip_division_by_m0,
        )

    def spectrum_moments(self, *, moment_0, moments, attr, rank, attr_bins,
                         attr_name="water mass", weighting_attribute="water mass",
                         weighting_rank=0):
        self.backend.spectrum_moments(
            moment_0=moment_0,
            moments=moments,
            multiplicity=self.attributes["multiplicity"],
            attr_data=self.attributes[attr],
            cell_id=self.attributes["cell id"],
            idx=self.attributes._fused_view()["idx"],  # pylint: disable=protected-access
            length=self.attributes.super_droplet_count,
            rank=rank,
            x_bins=attr_bins,
            x_attr=self.attributes[attr_name],
            weighting_attribute=self.attributes[weighting_attribute],
            weighting_rank=weighting_rank,
        )

    # ---- displacement (particulator.py:401-440) ---------------------------------------------------
    def remove_precipitated(self, *, displacement, precipitation_counting_level_index) -> float:
        view = self.attributes._fused_view()  # pylint: disable=protected-access
        rainfall_mass = self.backend.flag_precipitated(
            cell_origin=self.attributes["cell origin"],
            position_in_cell=self.attributes["position in cell"],
            water_mass=self.attributes["water mass"],
            multiplicity=self.attributes["multiplicity"],
            idx=view["idx"],
            length=self.attributes.super_droplet_count,
            healthy=view["healthy"],
            precipitation_counting_level_index=precipitation_counting_level_index,
            displacement=displacement,
        )
        self.attributes.sanitize()
        return rainfall_mass

    def flag_out_of_column(self):
        view = self.attributes._fused_view()  # pylint: disable=protected-access
        self.backend.flag_out_of_column(
            cell_origin=self.attributes["cell origin"],
            position_in_cell=self.attributes["position in cell"],
            idx=view["idx"],
            length=self.attributes.super_droplet_count,
            healthy=view["healthy"],
            domain_top_level_index=self.mesh.grid[-1],
        )
        self.attributes.sanitize()

    def calculate_displacement(self, *, displacement, courant, cell_origin, position_in_cell,
                               n_substeps):
        for dim in range(len(self.environment.mesh.grid)):
            self.backend.calculate_displacement(
                dim=dim,
                displacement=displacement,
                courant=courant[dim],
                cell_origin=cell_origin,
                position_in_cell=position_in_cell,
                n_substeps=n_substeps,
            )

    def adaptive_sdm_end(self, dt_left):
        return self.backend.adaptive_sdm_end(dt_left, self.attributes.cell_start)


class Builder:
    def __init__(self, n_sd, backend, environment):
        assert not inspect_is_class(backend)
        self.formulae = backend.formulae
        self.particulator = Particulator(n_sd, backend)
        self.req_attr_names = ["multiplicity", "cell id"]
        self.req_attr = None
        self.particulator.environment = environment.instantiate(builder=self)

    def add_dynamic(self, dynamic):
        # all of Collision / Coalescence / Breakup register under one key (builder.py:54-58)
        key = getattr(dynamic, "DYNAMIC_KEY", dynamic.__class__.__name__)
        assert key not in self.particulator.dynamics
        self.particulator.dynamics[key] = dynamic

    def _resolve_attribute(self, name):
        if name not in self.req_attr:
            self.req_attr[name] = get_attribute_class(name)(self)

    def get_attribute(self, attribute_name):
        self.request_attribute(attribute_name)
        return self.req_attr[attribute_name]

    def request_attribute(self, attribute, variant=None):  # pylint: disable=unused-argument
        if self.req_attr_names is not None:
            self.req_attr_names.append(attribute)
        else:
            self._resolve_attribute(attribute)

    def build(self, attributes: dict, products: tuple = (),
              int_caster=discretise_multiplicities):
        assert self.particulator.environment is not None
        if products:
            raise NotImplementedError("products are outside the collision path")
        attributes = dict(attributes)
        if "n" in attributes and "multiplicity" not in attributes:
            attributes["multiplicity"] = attributes.pop("n")
            warnings.warn('renaming attributes["n"] to attributes["multiplicity"]',
                          DeprecationWarning)
        if "volume" in attributes and "water mass" not in attributes:
            attributes["water mass"] = (
                self.particulator.formulae.particle_shape_and_density.volume_to_mass(
                    attributes.pop("volume")
                )
            )
            self.request_attribute("volume")
        if "water mass" in attributes and "signed water mass" not in attributes:
            attributes["signed water mass"] = attributes.pop("water mass")
            self.request_attribute("water mass")

        names, self.req_attr_names, self.req_attr = self.req_attr_names, None, {}
        for name in names:
            self._resolve_attribute(name)
        for key, dynamic in self.particulator.dynamics.items():
            self.particulator.dynamics[key] = dynamic.instantiate(builder=self)
        for attribute in attributes:
            self.request_attribute(attribute)

        attributes["multiplicity"] = int_caster(attributes["multiplicity"])
        if self.particulator.mesh.dimension == 0:
            attributes["cell id"] = np.zeros_like(attributes["multiplicity"], dtype=np.int64)
        self.particulator.attributes = make_particle_attributes(
            self.particulator, self.req_attr, attributes
        )
        self.particulator.recalculate_cell_id()
        for key in self.particulator.dynamics:
            self.particulator.timers[key] = WallTimer()
        if (attributes["multiplicity"] == 0).any():
            self.particulator.attributes.healthy = False
            self.particulator.attributes.sanitize()
        return self.particulator


def inspect_is_class(obj):
    return isinstance(obj, type)


__all__ = ["Builder", "Particulator", "Formulae"]
