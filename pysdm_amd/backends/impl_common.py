"""Storage wrappers bound to a backend: `Index`, `IndexedStorage`, `PairwiseStorage`,
`PairIndicator` -- the types that cross the backend boundary on the collision path.

Host-side mirror of PySDM/backends/impl_common/{index.py:10-56, indexed_storage.py:8-55,
pairwise_storage.py:6-39, pair_indicator.py:6-19}: same class/method names and argument meaning,
so that code written against the reference's wrappers runs unchanged.
"""
import numpy as np

from .storage_base import StorageSignature


class BackendMethods:  # cf. PySDM/backends/impl_common/backend_methods.py:8-17
    def __init__(self):
        if not hasattr(self, "formulae"):
            self.formulae = None
        if not hasattr(self, "Storage"):
            self.Storage = None


class RandomCommon:  # cf. PySDM/backends/impl_common/random_common.py:6-10
    def __init__(self, size: int, seed: int):
        assert isinstance(size, int)
        assert isinstance(seed, int)
        self.size = size


def make_Index(backend):
    Storage = backend.Storage

    class Index(Storage):
        def __init__(self, data, length):
            assert isinstance(length, int)
            super().__init__(StorageSignature(data, length, Storage.INT))
            self.length = Storage.INT(length)

        def __len__(self):
            return self.length

        @staticmethod
        def identity_index(length):
            return Index.from_ndarray(np.arange(length, dtype=Storage.INT))

        def reset_index(self):
            backend.identity_index(self.data)

        @staticmethod
        def empty(*args, **kwargs):
            raise TypeError("'Index' class cannot be instantiated as empty.")

        @staticmethod
        def from_ndarray(array):
            signature = Storage._get_data_from_ndarray(array)
            return Index(signature.data, array.shape[0])

        def sort_by_key(self, keys):
            backend.sort_by_key(self, keys)

        def shuffle(self, temporary, parts=None):
            if parts is None:
                backend.shuffle_global(idx=self.data, length=self.length, u01=temporary.data)
            else:
                backend.shuffle_local(idx=self.data, u01=temporary.data, cell_start=parts.data)

        def remove_zero_n_or_flagged(self, indexed_storage):
            self.length = backend.remove_zero_n_or_flagged(
                indexed_storage.data, self.data, self.length
            )

    return Index


def make_IndexedStorage(backend):
    Storage = backend.Storage

    class IndexedStorage(Storage):
        def __init__(self, idx, signature):
            super().__init__(signature)
            assert idx is not None
            self.idx = idx

        def __len__(self):
            return len(self.idx)

        def __getitem__(self, item):
            result = Storage.__getitem__(self, item)
            if isinstance(result, Storage):
                return IndexedStorage.indexed(self.idx, result)
            return result

        @staticmethod
        def indexed(idx, storage):
            return IndexedStorage(
                idx, StorageSignature(storage.data, storage.shape, storage.dtype)
            )

        @staticmethod
        def empty(idx, shape, dtype):
            return IndexedStorage.indexed(idx, Storage.empty(shape, dtype))

        @staticmethod
        def from_ndarray(idx, array):
            return IndexedStorage.indexed(idx, Storage.from_ndarray(array))

        def to_ndarray(self, *, raw=False):
            result = Storage.to_ndarray(self)
            if raw:
                return result
            order = self.idx.to_ndarray()[: len(self)]
            if len(self.shape) == 1:
                return result[order]
            if len(self.shape) == 2:
                return result[:, order]
            raise NotImplementedError()

    return IndexedStorage


def make_PairwiseStorage(backend):
    Storage = backend.Storage

    class PairwiseStorage(Storage):
        @staticmethod
        def empty(shape, dtype):
            return PairwiseStorage(Storage._get_empty_data(shape, dtype))

        @staticmethod
        def from_ndarray(array):
            return PairwiseStorage(Storage._get_data_from_ndarray(array))

        def distance(self, other, is_first_in_pair):
            backend.distance_pair(self, other, is_first_in_pair, other.idx)

        def max(self, other, is_first_in_pair):
            backend.max_pair(self, other, is_first_in_pair, other.idx)

        def min(self, other, is_first_in_pair):
            backend.min_pair(self, other, is_first_in_pair, other.idx)

        def sort(self, other, is_first_in_pair):
            backend.sort_pair(self, other, is_first_in_pair, other.idx)

        def sum(self, other, is_first_in_pair):
            backend.sum_pair(self, other, is_first_in_pair, other.idx)

        def multiply(self, other, is_first_in_pair):
            backend.multiply_pair(self, other, is_first_in_pair, other.idx)

    return PairwiseStorage


def make_PairIndicator(backend):
    class PairIndicator:
        def __init__(self, length):
            self.indicator = backend.Storage.empty(length, dtype=bool)
            self.length = length

        def __len__(self):
            return self.length

        def update(self, cell_start, cell_idx, cell_id):
            backend.find_pairs(cell_start, self, cell_id, cell_idx, cell_id.idx)
            self.length = len(cell_id)

    return PairIndicator


def advection_scheme_id(formulae):
    """SDM scheme code of sdm_calculate_displacement for `formulae.particle_advection` (this
    package's or PySDM's own object: PySDM/physics/particle_advection/*.py)"""
    scheme = formulae.particle_advection
    code = getattr(scheme, "scheme_id", None)
    if code is None:
        # PySDM wraps the chosen class in a namespace that keeps its name (formulae.py:144-160)
        name = getattr(scheme, "__name__", type(scheme).__name__)
        code = {"ImplicitInSpace": 0, "ExplicitInSpace": 1}[name]
    return code
