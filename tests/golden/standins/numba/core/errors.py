class NumbaError(Exception):
    pass


class NumbaExperimentalFeatureWarning(Warning):
    pass


class UnsupportedParforsError(NumbaError):
    pass
