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
