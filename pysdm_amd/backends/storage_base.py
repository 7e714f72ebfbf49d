"""Backend-neutral part of the `Storage` contract of the collision path.

Mirrors what the reference's callers rely on (PySDM/backends/impl_numba/storage.py:16-213 and
PySDM/backends/impl_common/storage_utils.py:10-79): `.data/.shape/.dtype`, `FLOAT/INT/BOOL`,
`empty` (NaN / -1 filled), `from_ndarray` (copies, casts by dtype prefix), slicing to views,
in-place arithmetic only.  Concrete backends provide the raw-array primitives (`_alloc`,
`_upload_raw`, `_download_raw`, `_ew`, `_reduce`).
"""
from collections import namedtuple

import numpy as np

StorageSignature = namedtuple("StorageSignature", ("data", "shape", "dtype"))

# element-wise op codes == SDM_EW_* of include/sdm_hip.h
EW_ADD, EW_SUB, EW_MUL, EW_DIV, EW_POW, EW_DIV_IF_NOT_ZERO = 0, 1, 2, 3, 4, 5
EW_FLOOR, EW_EXP, EW_ABS, EW_FILL, EW_ADD_MUL, EW_MOD = 6, 7, 8, 9, 10, 11


def _forbidden(hint):
    def raiser(self, other):
        raise TypeError(f"Use {hint}")

    return raiser


class StorageBase:
    FLOAT = np.float64
    INT = np.int64
    BOOL = np.bool_

    # out-of-place arithmetic is not part of the contract (storage_utils.py:28-44)
    __pow__ = _forbidden("**=")
    __mod__ = _forbidden("%=")
    __truediv__ = _forbidden("/=")
    __mul__ = _forbidden("*=")
    __sub__ = _forbidden("-=")
    __add__ = _forbidden("+=")

    def __init__(self, signature):
        self.data = signature.data
        shape = signature.shape
        self.shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(shape)
        self.dtype = signature.dtype
        self.backend = None

    def __len__(self):
        return self.shape[0]

    # ---- primitives supplied by the concrete class -------------------------------------------
    @classmethod
    def _alloc(cls, shape, dtype):
        raise NotImplementedError

    @classmethod
    def _upload_raw(cls, array):
        raise NotImplementedError

    @staticmethod
    def _download_raw(raw):
        raise NotImplementedError

    @staticmethod
    def _assign_raw(raw, key, value):
        raise NotImplementedError

    def _ew(self, op, a, b=None, scalar=0.0):
        """self.data = a (op) b-or-scalar, element-wise"""
        raise NotImplementedError

    def _reduce(self, kind):
        raise NotImplementedError

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def _resolve_dtype(cls, dtype):
        if dtype in (float, cls.FLOAT):
            return cls.FLOAT
        if dtype in (int, cls.INT):
            return cls.INT
        if dtype in (bool, cls.BOOL):
            return cls.BOOL
        raise NotImplementedError()

    @classmethod
    def _get_empty_data(cls, shape, dtype):
        dtype = cls._resolve_dtype(dtype)
        raw = cls._alloc(shape, dtype)
        sentinel = np.nan if dtype is cls.FLOAT else (True if dtype is cls.BOOL else -1)
        cls._assign_raw(raw, slice(None), sentinel)
        return StorageSignature(raw, shape, dtype)

    @classmethod
    def empty(cls, shape, dtype):
        return cls(cls._get_empty_data(shape, dtype))

    @classmethod
    def _get_data_from_ndarray(cls, array):
        kind = str(array.dtype)
        if kind.startswith("int"):
            dtype = cls.INT
        elif kind.startswith("float"):
            dtype = cls.FLOAT
        elif kind.startswith("bool"):
            dtype = cls.BOOL
        else:
            raise NotImplementedError()
        return StorageSignature(
            cls._upload_raw(np.ascontiguousarray(array.astype(dtype))), array.shape, dtype
        )

    @classmethod
    def from_ndarray(cls, array):
        return cls(cls._get_data_from_ndarray(array))

    # ---- host transfer ------------------------------------------------------------------------
    def to_ndarray(self):
        return self._download_raw(self.data)

    def upload(self, data):
        data = np.asarray(data)
        if not np.can_cast(data.dtype, self.dtype, casting="safe"):
            raise TypeError(f"cannot safely cast {data.dtype} to {self.dtype}")
        self._assign_raw(self.data, slice(None), self._upload_raw(data.astype(self.dtype)))

    def download(self, target, reshape=False):
        host = self.to_ndarray()
        np.copyto(target, host.reshape(target.shape) if reshape else host, casting="safe")

    # ---- indexing -----------------------------------------------------------------------------
    def __getitem__(self, item):
        dim = len(self.shape)
        cls = _plain_storage_class(type(self))
        if isinstance(item, slice):
            if (item.step or 1) != 1:
                raise NotImplementedError("step != 1")
            if dim not in (1, 2):
                raise NotImplementedError("Only 2 or less dimensions array is supported.")
            start = item.start or 0
            stop = item.stop or self.shape[0]
            if stop > self.shape[0]:
                raise IndexError(
                    f"requested a slice ({start}:{stop}) of Storage"
                    f" with first dim of length {self.shape[0]}"
                )
            shape = (stop - start,) + tuple(self.shape[1:])
            return cls(StorageSignature(self.data[item], shape, self.dtype))
        if isinstance(item, tuple) and dim == 2 and isinstance(item[1], slice):
            return cls(StorageSignature(self.data[item[0]], tuple(self.shape[1:]), self.dtype))
        return self._scalar(self.data[item])

    @staticmethod
    def _scalar(raw_element):
        return raw_element

    def __setitem__(self, key, value):
        self._assign_raw(self.data, key, value.data if isinstance(value, StorageBase) else value)
        return self

    # ---- in-place arithmetic (storage.py:63-109) -----------------------------------------------
    def __iadd__(self, other):
        if isinstance(other, StorageBase):
            self._ew(EW_ADD, self.data, other.data)
        elif (
            isinstance(other, tuple)
            and len(other) == 3
            and isinstance(other[0], float)
            and other[1] == "*"
            and isinstance(other[2], StorageBase)
        ):
            self._ew(EW_ADD_MUL, self.data, other[2].data, other[0])
        else:
            self._ew(EW_ADD, self.data, None, other)
        return self

    def __isub__(self, other):
        self._ew(EW_SUB, self.data, other.data)
        return self

    def __imul__(self, other):
        if isinstance(other, StorageBase):
            self._ew(EW_MUL, self.data, other.data)
        else:
            self._ew(EW_MUL, self.data, None, other)
        return self

    def __itruediv__(self, other):
        if isinstance(other, StorageBase):
            self._ew(EW_DIV, self.data, other.data)
        else:
            self._ew(EW_DIV, self.data, None, other)
        return self

    def __imod__(self, other):
        # row-wise modulo of a (n_dim, n_sd) storage by a per-row divisor (storage_impl.py:36-41)
        divisor = other.to_ndarray()
        for row in range(self.shape[0]):
            type(self)._row_mod(self, row, divisor[row])
        return self

    def _row_mod(self, row, divisor):
        view = self[row, :]
        view._ew(EW_MOD, view.data, None, divisor)

    def __ipow__(self, other):
        self._ew(EW_POW, self.data, None, other)
        return self

    def __bool__(self):
        if len(self) == 1:
            return bool(self.to_ndarray().ravel()[0] != 0)
        raise NotImplementedError("Logic value of array is ambiguous.")

    # ---- named ops ----------------------------------------------------------------------------
    def amin(self):
        return self._reduce(0)

    def amax(self):
        return self._reduce(1)

    def all(self):
        return bool(self.to_ndarray().all())

    def floor(self, other=None):
        self._ew(EW_FLOOR, self.data if other is None else other.data)
        return self

    def product(self, multiplicand, multiplier):
        if isinstance(multiplier, StorageBase):
            self._ew(EW_MUL, multiplicand.data, multiplier.data)
        else:
            self._ew(EW_MUL, multiplicand.data, None, multiplier)
        return self

    def ratio(self, dividend, divisor):
        self._ew(EW_DIV, dividend.data, divisor.data)
        return self

    def divide_if_not_zero(self, divisor):
        self._ew(EW_DIV_IF_NOT_ZERO, self.data, divisor.data)
        return self

    def sum(self, arg_a, arg_b):
        self._ew(EW_ADD, arg_a.data, arg_b.data)
        return self

    def ravel(self, other):
        host = other.to_ndarray() if isinstance(other, StorageBase) else np.asarray(other)
        self.upload(host.ravel().astype(self.dtype))

    def urand(self, generator):
        generator(self)

    def fill(self, other):
        if isinstance(other, StorageBase):
            self._assign_raw(self.data, slice(None), other.data)
        else:
            self._ew(EW_FILL, None, None, other)

    def exp(self):
        self._ew(EW_EXP, self.data)

    def abs(self):
        self._ew(EW_ABS, self.data)

    def detach(self):
        self.data = self._upload_raw(self.to_ndarray())


def _plain_storage_class(cls):
    """views produced by slicing are plain Storages of the backend (not Index / Pairwise...)"""
    for base in cls.__mro__:
        if base.__dict__.get("_IS_BACKEND_STORAGE", False):
            return base
    return cls
