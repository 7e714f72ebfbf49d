"""Stand-in for `pint` (golden generation only).

PySDM evaluates `(1.0 * si.<unit>).to_base_units().magnitude` once per unit name at import time
to build its FakeUnitRegistry of plain floats.  This shim supplies those SI base-unit magnitudes
for the (prefix x unit) names PySDM asks for.  Contains no reference code.
"""

_PREFIX = {
    "nano": 1e-9, "micro": 1e-6, "milli": 1e-3, "centi": 1e-2, "deci": 1e-1, "": 1.0,
    "hecto": 1e2, "kilo": 1e3,
    "n": 1e-9, "u": 1e-6, "m": 1e-3, "c": 1e-2, "d": 1e-1, "h": 1e2, "k": 1e3,
}
# magnitudes in SI base units (kg, m, s, K, mol)
_UNIT = {
    "bar": 1e5, "metre": 1.0, "meter": 1.0, "gram": 1e-3, "hertz": 1.0, "mole": 1.0,
    "joule": 1.0, "kelvin": 1.0, "second": 1.0, "minute": 60.0, "pascal": 1.0,
    "litre": 1e-3, "liter": 1e-3, "hour": 3600.0, "newton": 1.0, "watt": 1.0,
    "b": 1e-28, "m": 1.0, "g": 1e-3, "Hz": 1.0, "mol": 1.0, "J": 1.0, "K": 1.0, "s": 1.0,
    "min": 60.0, "day": 86400.0, "Pa": 1.0, "l": 1e-3, "h": 3600.0, "N": 1.0, "W": 1.0,
    "kg": None,  # handled through prefix k + g
}


class Quantity:  # pylint: disable=too-few-public-methods
    def __init__(self, magnitude):
        self.magnitude = magnitude

    def __rmul__(self, other):
        return Quantity(other * self.magnitude)

    def __mul__(self, other):
        return Quantity(self.magnitude * other)

    def to_base_units(self):
        return self


class Unit(Quantity):  # pylint: disable=too-few-public-methods
    pass


def _lookup(name):
    for cand in (name, name[:-1] if name.endswith("s") else None):
        if cand is None:
            continue
        if cand in _UNIT and _UNIT[cand] is not None:
            return _UNIT[cand]
        for prefix in sorted(_PREFIX, key=len, reverse=True):
            if prefix and cand.startswith(prefix):
                rest = cand[len(prefix):]
                if rest in _UNIT and _UNIT[rest] is not None:
                    return _PREFIX[prefix] * _UNIT[rest]
    raise AttributeError(name)


class UnitRegistry:  # pylint: disable=too-few-public-methods
    Quantity = Quantity
    Unit = Unit

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return Unit(_lookup(name))

    def parse_expression(self, *_a, **_k):
        raise NotImplementedError("products need genuine pint")
