"""Stand-in for `pint` (golden generation and the run of the reference's own unit tests only).

PySDM evaluates `(1.0 * si.<unit>).to_base_units().magnitude` once per unit name at import time
to build its FakeUnitRegistry of plain floats; its products parse a unit string and ask for the
magnitude in base units and for `dimensionality` (compared with that of the default unit).  This
shim supplies SI base-unit magnitudes and dimension exponents for the (prefix x unit) names PySDM
asks for, products and quotients of them, and a parser for expressions such as "m^-3" or
"kg / m**3".  Contains no reference code.
"""
import re

_PREFIX = {
    "nano": 1e-9, "micro": 1e-6, "milli": 1e-3, "centi": 1e-2, "deci": 1e-1, "": 1.0,
    "hecto": 1e2, "kilo": 1e3,
    "n": 1e-9, "u": 1e-6, "m": 1e-3, "c": 1e-2, "d": 1e-1, "h": 1e2, "k": 1e3,
}
# (magnitude in SI base units, exponents of (kg, m, s, K, mol))
_PRESSURE, _ENERGY = (1, -1, -2, 0, 0), (1, 2, -2, 0, 0)
_LENGTH, _MASS, _TIME, _NONE = (0, 1, 0, 0, 0), (1, 0, 0, 0, 0), (0, 0, 1, 0, 0), (0, 0, 0, 0, 0)
_UNIT = {
    "bar": (1e5, _PRESSURE), "metre": (1.0, _LENGTH), "meter": (1.0, _LENGTH),
    "gram": (1e-3, _MASS), "hertz": (1.0, (0, 0, -1, 0, 0)), "mole": (1.0, (0, 0, 0, 0, 1)),
    "joule": (1.0, _ENERGY), "kelvin": (1.0, (0, 0, 0, 1, 0)), "second": (1.0, _TIME),
    "minute": (60.0, _TIME), "pascal": (1.0, _PRESSURE), "litre": (1e-3, (0, 3, 0, 0, 0)),
    "liter": (1e-3, (0, 3, 0, 0, 0)), "hour": (3600.0, _TIME), "newton": (1.0, (1, 1, -2, 0, 0)),
    "watt": (1.0, (1, 2, -3, 0, 0)),
    "b": (1e-28, (0, 2, 0, 0, 0)), "m": (1.0, _LENGTH), "g": (1e-3, _MASS),
    "Hz": (1.0, (0, 0, -1, 0, 0)), "mol": (1.0, (0, 0, 0, 0, 1)), "J": (1.0, _ENERGY),
    "K": (1.0, (0, 0, 0, 1, 0)), "s": (1.0, _TIME), "min": (60.0, _TIME), "day": (86400.0, _TIME),
    "Pa": (1.0, _PRESSURE), "l": (1e-3, (0, 3, 0, 0, 0)), "h": (3600.0, _TIME),
    "N": (1.0, (1, 1, -2, 0, 0)), "W": (1.0, (1, 2, -3, 0, 0)),
    "dimensionless": (1.0, _NONE), "percent": (0.01, _NONE),
}


class Quantity:
    def __init__(self, magnitude, dimensionality=_NONE):
        self.magnitude = magnitude
        self.dimensionality = tuple(dimensionality)

    @staticmethod
    def _parts(other):
        if isinstance(other, Quantity):
            return other.magnitude, other.dimensionality
        return other, _NONE

    def __mul__(self, other):
        magnitude, dims = self._parts(other)
        return Quantity(self.magnitude * magnitude,
                        tuple(a + b for a, b in zip(self.dimensionality, dims)))

    __rmul__ = __mul__

    def __truediv__(self, other):
        magnitude, dims = self._parts(other)
        return Quantity(self.magnitude / magnitude,
                        tuple(a - b for a, b in zip(self.dimensionality, dims)))

    def __rtruediv__(self, other):
        return Quantity(other / self.magnitude, tuple(-a for a in self.dimensionality))

    def __pow__(self, power):
        return Quantity(self.magnitude ** power, tuple(a * power for a in self.dimensionality))

    def to_base_units(self):
        return self


class Unit(Quantity):
    pass


def _lookup(name):
    for cand in (name, name[:-1] if name.endswith("s") else None):
        if cand is None:
            continue
        if cand in _UNIT:
            return _UNIT[cand]
        for prefix in sorted(_PREFIX, key=len, reverse=True):
            if prefix and cand.startswith(prefix):
                rest = cand[len(prefix):]
                if rest in _UNIT:
                    return _PREFIX[prefix] * _UNIT[rest][0], _UNIT[rest][1]
    raise AttributeError(name)


class _Names(dict):
    def __missing__(self, name):
        return Unit(*_lookup(name))


class UnitRegistry:
    Quantity = Quantity
    Unit = Unit

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return Unit(*_lookup(name))

    @staticmethod
    def parse_expression(expression, *_a, **_k):
        text = str(expression).strip() or "dimensionless"
        if not re.fullmatch(r"[A-Za-z0-9_ .+\-*/^()]*", text):
            raise ValueError(f"unit expression not understood: {expression!r}")
        value = eval(text.replace("^", "**"), {"__builtins__": {}}, _Names())  # pylint: disable=eval-used
        return value if isinstance(value, Quantity) else Quantity(float(value))
