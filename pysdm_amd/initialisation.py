"""Initial conditions of the box configurations: exponential volume spectrum sampled at constant
multiplicity, and integer discretisation of multiplicities.

Closed-form restatement of what the reference does through scipy.stats
(PySDM/initialisation/spectra/exponential.py:11-13, impl/spectrum.py,
sampling/spectral_sampling.py:45-108, discretise_multiplicities.py:8-32); checked against the
reference's own samples stored in the trajectory goldens.
"""
import numpy as np

default_cdf_range = (0.00001, 0.99999)


class Exponential:
    """n(x) = norm_factor / scale * exp(-x / scale)"""

    def __init__(self, norm_factor, scale):
        self.norm_factor = norm_factor
        self.scale = scale

    def cumulative(self, arg):
        return self.norm_factor * -np.expm1(-np.asarray(arg) / self.scale)

    def percentiles(self, cdf_values):
        return -self.scale * np.log1p(-np.asarray(cdf_values))


class ConstantMultiplicity:
    def __init__(self, spectrum, size_range=None, error_threshold=None):
        self.spectrum = spectrum
        self.size_range = size_range or spectrum.percentiles(default_cdf_range)
        self.error_threshold = error_threshold or 0.01
        self.cdf_range = (
            spectrum.cumulative(self.size_range[0]),
            spectrum.cumulative(self.size_range[1]),
        )
        assert 0 < self.cdf_range[0] < self.cdf_range[1]

    def sample(self, n_sd, *, backend=None):  # pylint: disable=unused-argument
        cdf_arg = np.linspace(self.cdf_range[0], self.cdf_range[1], num=2 * n_sd + 1)
        cdf_arg /= self.spectrum.norm_factor
        grid = self.spectrum.percentiles(cdf_arg)
        assert np.isfinite(grid).all()
        x = grid[1:-1:2]
        cdf = self.spectrum.cumulative(grid[0::2])
        y_float = cdf[1:] - cdf[:-1]
        diff = abs(1 - np.sum(y_float) / self.spectrum.norm_factor)
        if diff > self.error_threshold:
            raise ValueError(
                f"{diff * 100:.3g}% error in total real-droplet number due to sampling "
                f"({len(x)} samples)"
            )
        return x, y_float


def discretise_multiplicities(values_arg):
    """NaNs are flagged with zero multiplicity; rounding must not lose >1% of the droplets"""
    values_arg = np.asarray(values_arg)
    values_int = np.where(np.isnan(values_arg), 0, values_arg).round().astype(np.int64)
    if np.issubdtype(values_arg.dtype, np.floating):
        if np.isnan(values_arg).all():
            return values_int
        if not np.logical_or(values_int > 0, np.isnan(values_arg)).all():
            raise ValueError(
                f"int-casting resulted in multiplicity of zero (min(y_float)={min(values_arg)})"
            )
        percent_diff = 100 * abs(1 - np.nansum(values_arg) / np.sum(values_int.astype(float)))
        if percent_diff > 1:
            raise ValueError(
                f"{percent_diff}% error in total real-droplet number"
                f" due to casting multiplicities to ints"
            )
    return values_int
