"""Stand-in for `chempy` (golden generation only); chemistry is off the collision path."""


class Substance:  # pylint: disable=too-few-public-methods
    def __init__(self, mass=1.0):
        self.mass = mass

    @staticmethod
    def from_formula(_formula):
        return Substance(1.0)
