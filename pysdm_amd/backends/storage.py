"""`Storage` of the PySDM-shaped backends: an engine array with the operator contract PySDM's
front-end relies on.

What callers may do with a Storage is fixed by the reference (PySDM/backends/impl_numba/
storage.py:16-213, impl_common/storage_utils.py:10-79): `.data / .shape / .dtype`, the dtype tags
`FLOAT / INT / BOOL`, `empty` (NaN / -1 / True filled), `from_ndarray` (copies; casts by dtype
family), slices as views, scalars as host values, arithmetic **in place only**, a few named
operations.  Everything element-wise is one `sdm_elementwise_*` launch of the engine's library
(SDM_EW_* codes); nothing is computed on the host.
"""
import ctypes

import numpy as np

from ..engine import BOOL, FLOAT, INT

# SDM_EW_* of include/sdm_hip.h
ADD, SUB, MUL, DIV, POW, DIV_IF_NOT_ZERO, FLOOR, EXP, ABS, FILL, ADD_MUL, MOD = range(12)

_FAMILY = {"i": INT, "u": INT, "f": FLOAT, "b": BOOL}
_SENTINEL = {FLOAT: np.nan, INT: -1, BOOL: True}


def _tag(dtype):
    if dtype in (float, FLOAT):
        return FLOAT
    if dtype in (int, INT):
        return INT
    if dtype in (bool, BOOL):
        return BOOL
    raise NotImplementedError(f"Storage dtype {dtype!r}")


def _in_place_only(symbol):
    def refuse(self, other):
        raise TypeError(f"Use {symbol}=")
    return refuse


class Storage:  # pylint: disable=too-many-public-methods
    """bound to an engine by `storage_class_for`"""

    FLOAT, INT, BOOL = FLOAT, INT, BOOL
    engine_getter = None

    __add__, __sub__, __mul__ = (_in_place_only(s) for s in "+-*")
    __truediv__, __mod__, __pow__ = (_in_place_only(s) for s in ("/", "%", "**"))

    def __init__(self, data, shape=None, dtype=None):
        if shape is None and isinstance(data, tuple):  # a (data, shape, dtype) triple
            data, shape, dtype = data
        self.data = data
        self.shape = (int(shape),) if isinstance(shape, (int, np.integer)) else tuple(shape)
        self.dtype = dtype
        self.backend = None

    # ---- engine -------------------------------------------------------------------------------------
    @classmethod
    def engine(cls):
        return cls.engine_getter()

    def _wrap(self, data, shape):
        plain = type(self).plain_class()
        return plain(data, shape, self.dtype)

    @classmethod
    def plain_class(cls):
        """views are plain Storages even when taken from a subclass (an index, a pair array...)"""
        for base in cls.__mro__:
            if base.__dict__.get("IS_PLAIN", False):
                return base
        return cls

    # ---- construction -------------------------------------------------------------------------------
    @classmethod
    def empty(cls, shape, dtype):
        tag = _tag(dtype)
        return cls(cls.engine().full(shape, tag, _SENTINEL[tag]), shape, tag)

    @classmethod
    def from_ndarray(cls, array):
        array = np.asarray(array)
        tag = _FAMILY.get(array.dtype.kind)
        if tag is None:
            raise NotImplementedError(f"Storage from {array.dtype}")
        return cls(cls.engine().upload(array.astype(tag)), array.shape, tag)

    # names PySDM's own wrapper factories use (impl_common/index.py:34, pairwise_storage.py:9-14)
    @classmethod
    def _get_empty_data(cls, shape, dtype):
        made = cls.plain_class().empty(shape, dtype)
        return made.data, made.shape, made.dtype

    @classmethod
    def _get_data_from_ndarray(cls, array):
        made = cls.plain_class().from_ndarray(array)
        return made.data, made.shape, made.dtype

    # ---- host transfer ------------------------------------------------------------------------------
    def to_ndarray(self):
        return self.engine().download(self.data)

    def upload(self, values):
        values = np.asarray(values)
        if not np.can_cast(values.dtype, self.dtype, casting="safe"):
            raise TypeError(f"cannot safely cast {values.dtype} to {self.dtype}")
        eng = self.engine()
        eng.assign(self.data, eng.upload(values.astype(self.dtype)))

    def download(self, target, reshape=False):
        host = self.to_ndarray()
        np.copyto(target, host.reshape(target.shape) if reshape else host, casting="safe")

    def detach(self):
        self.data = self.engine().upload(self.to_ndarray())

    # ---- indexing -----------------------------------------------------------------------------------
    def __len__(self):
        return self.shape[0]

    def __getitem__(self, item):
        rank = len(self.shape)
        if isinstance(item, slice):
            if item.step not in (None, 1):
                raise NotImplementedError("step != 1")
            if rank not in (1, 2):
                raise NotImplementedError("Only 2 or less dimensions array is supported.")
            first, last = item.start or 0, item.stop or self.shape[0]
            if last > self.shape[0]:
                raise IndexError(f"requested a slice ({first}:{last}) of Storage with first dim "
                                 f"of length {self.shape[0]}")
            return self._wrap(self.data[item], (last - first,) + self.shape[1:])
        if isinstance(item, tuple) and rank == 2 and isinstance(item[1], slice):
            return self._wrap(self.data[item[0]], self.shape[1:])
        element = self.data[item]
        if self.engine().size(element) != 1:
            # e.g. a column `storage[:, 0]` (tests/unit_tests/dynamics/displacement/
            # test_advection.py:84): the reference hands back `self.data[item]`, a host array there
            return self.engine().download(element)
        return element.item() if hasattr(element, "item") else element

    def __setitem__(self, key, value):
        self.data[key] = value.data if isinstance(value, Storage) else value
        return self

    def __bool__(self):
        if len(self) != 1:
            raise NotImplementedError("Logic value of array is ambiguous.")
        return bool(self.to_ndarray().ravel()[0] != 0)

    # ---- element-wise kernels -------------------------------------------------------------------------
    def _apply(self, code, left, right=None, scalar=0.0):
        """self.data = left (code) right-or-scalar"""
        eng = self.engine()
        out, n = self.data, eng.size(self.data)
        for operand in (left, right):  # the kernels index every operand by the output's length
            if operand is not None and eng.size(operand) != n:
                raise ValueError(f"operand of {eng.size(operand)} elements for an output of {n}")
        # an operand of the other numeric family (float += int storage: tests/unit_tests/backends/
        # storage/test_basic_ops.py:9-25; NumPy casts there) is converted first - the kernels are
        # typed.  The two mixed operations the displacement needs have kernels of their own (below)
        mixed = not ((code == FLOOR and self.dtype is INT) or
                     (code == SUB and self.dtype is FLOAT and right is not None and _is_int(right)))
        if mixed and self.dtype in (FLOAT, INT):
            want_int = self.dtype is INT
            left, right = (operand if operand is None or _is_int(operand) == want_int
                           else eng.upload(eng.download(operand).astype(self.dtype))
                           for operand in (left, right))
        if code == FLOOR and self.dtype is INT:
            eng.call("sdm_floor_to_i64", out, left, n)
        elif code == SUB and self.dtype is FLOAT and right is not None and _is_int(right):
            eng.call("sdm_subtract_i64", out, right, n)
        elif self.dtype is FLOAT:
            eng.call("sdm_elementwise_f64", code, out, left, right, float(scalar), n)
        elif self.dtype is INT:
            eng.call("sdm_elementwise_i64", code, out, left, right, int(scalar), n)
        else:
            raise NotImplementedError("arithmetic on bool storage")

    def _in_place(self, code, other):
        if isinstance(other, Storage):
            # operands of another length: NumPy's rule, as in the reference, whose in-place operators
            # are NumPy's (storage_impl.py:12-13,56-57) - a one-element array applies to every
            # element (SimpleGeometric multiplies its n_sd-long output by the single pair value in
            # tests/unit_tests/dynamics/collisions/test_kernels.py:33-57), anything else is an error.
            # (Checked here: the kernels index both operands by the output's length.)
            eng = self.engine()
            n_out, n_other = eng.size(self.data), eng.size(other.data)
            if n_other != n_out:
                if n_other != 1:
                    raise ValueError(f"operands could not be broadcast together with shapes "
                                     f"{self.shape} {other.shape}")
                value = other.to_ndarray().ravel()[0]
                self._apply(code, self.data, None, value)
                return self
            self._apply(code, self.data, other.data)
        else:
            self._apply(code, self.data, None, other)
        return self

    def __iadd__(self, other):
        # `x += (factor, "*", y)` is the reference's add-with-multiplier (storage.py:66-74)
        if (isinstance(other, tuple) and len(other) == 3 and isinstance(other[0], float)
                and other[1] == "*" and isinstance(other[2], Storage)):
            self._apply(ADD_MUL, self.data, other[2].data, other[0])
            return self
        return self._in_place(ADD, other)

    def __isub__(self, other):
        self._apply(SUB, self.data, other.data)
        return self

    def __imul__(self, other):
        return self._in_place(MUL, other)

    def __itruediv__(self, other):
        return self._in_place(DIV, other)

    def __ipow__(self, other):
        self._apply(POW, self.data, None, other)
        return self

    def __imod__(self, other):
        """row r of a (rows, n) storage modulo other[r] (storage_impl.py:36-41)"""
        divisors = other.to_ndarray()
        for row in range(self.shape[0]):
            view = self[row, :]
            view._apply(MOD, view.data, None, divisors[row])  # pylint: disable=protected-access
        return self

    # ---- named operations -----------------------------------------------------------------------------
    def _extreme(self, kind):
        if self.dtype is not FLOAT:
            host = self.to_ndarray()
            return host.min() if kind == 0 else host.max()
        eng = self.engine()
        return eng.scalar_out("sdm_reduce_f64", ctypes.c_double, kind, self.data,
                              eng.size(self.data))

    def amin(self):
        return self._extreme(0)

    def amax(self):
        return self._extreme(1)

    def all(self):
        return bool(self.to_ndarray().all())

    def floor(self, other=None):
        self._apply(FLOOR, self.data if other is None else other.data)
        return self

    def product(self, multiplicand, multiplier):
        if isinstance(multiplier, Storage):
            self._apply(MUL, multiplicand.data, multiplier.data)
        else:
            self._apply(MUL, multiplicand.data, None, multiplier)
        return self

    def ratio(self, dividend, divisor):
        self._apply(DIV, dividend.data, divisor.data)
        return self

    def sum(self, arg_a, arg_b):
        self._apply(ADD, arg_a.data, arg_b.data)
        return self

    def divide_if_not_zero(self, divisor):
        self._apply(DIV_IF_NOT_ZERO, self.data, divisor.data)
        return self

    def exp(self):
        self._apply(EXP, self.data)

    def abs(self):
        self._apply(ABS, self.data)

    def fill(self, other):
        if isinstance(other, Storage):
            self.engine().assign(self.data, other.data)
        elif isinstance(other, np.ndarray) and other.ndim:
            # e.g. `output.fill(np.exp(storage))` in the reference's Straub2010Ec / LowList1982Ec:
            # numpy evaluated a Storage on the host (through __array__), the result comes back
            self.upload(other.astype(self.dtype))
        else:
            self._apply(FILL, None, None, other)

    def __array__(self, dtype=None, copy=None):  # pylint: disable=unused-argument
        host = self.to_ndarray()
        return host if dtype is None else host.astype(dtype)

    def ravel(self, other):
        host = other.to_ndarray() if isinstance(other, Storage) else np.asarray(other)
        self.upload(host.ravel().astype(self.dtype))

    def urand(self, generator):
        generator(self)


def _is_int(array):
    return str(array.dtype).rsplit(".", maxsplit=1)[-1] == "int64"


def storage_class_for(engine_getter, name="Storage"):
    return type(name, (Storage,), {"engine_getter": staticmethod(engine_getter),
                                   "IS_PLAIN": True})
