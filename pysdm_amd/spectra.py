"""Initial size spectra of the box set-ups and their deterministic sampling into super-droplets.

`Exponential(norm_factor, scale)`: n(x) = norm_factor / scale * exp(-x / scale), in closed form
(what the reference evaluates through scipy.stats.expon, PySDM/initialisation/spectra/
exponential.py:11-13).  Samplers return (x at the odd nodes of a 2 n_sd + 1 node grid, number of
real droplets between the even nodes), the construction of PySDM/initialisation/sampling/
spectral_sampling.py:45-108:
  * `sample_constant_multiplicity`: nodes equidistant in the cumulative distribution, so every
    super-droplet stands for (nearly) the same number of droplets - Shima et al. 2009;
  * `sample_logarithmic`: nodes equidistant in log10(x) - used for the rain spectrum.
The default range cuts the spectrum at the 1e-5 and 1 - 1e-5 quantiles.  The samples are inputs
that full-size parity digests depend on bit for bit (tests/golden/digest_*.npz pin them).
"""
import numpy as np
from scipy import special

CDF_RANGE = (0.00001, 0.99999)


class Exponential:
    def __init__(self, norm_factor, scale):
        self.norm_factor = norm_factor
        self.scale = scale

    def cumulative(self, x):
        # scipy.special (not numpy): scipy.stats.expon evaluates its cdf / ppf with these very
        # functions, and numpy's expm1 / log1p round differently in the last bit
        return self.norm_factor * -special.expm1(-(np.asarray(x) / self.scale))

    def percentiles(self, quantiles):
        return -special.log1p(-np.asarray(quantiles)) * self.scale


def _between_even_nodes(spectrum, grid, tolerance):
    x = grid[1:-1:2]
    cdf = spectrum.cumulative(grid[0::2])
    counts = cdf[1:] - cdf[:-1]
    lost = abs(1 - np.sum(counts) / spectrum.norm_factor)
    if lost > tolerance:
        raise ValueError(f"{lost * 100:.3g}% error in total real-droplet number due to sampling "
                         f"({len(x)} samples)")
    return x, counts


def sample_constant_multiplicity(spectrum, n_sd, size_range=None, tolerance=0.01):
    lo, hi = size_range or spectrum.percentiles(CDF_RANGE)
    cdf_lo, cdf_hi = spectrum.cumulative(lo), spectrum.cumulative(hi)
    if not 0 < cdf_lo < cdf_hi:
        raise ValueError("empty size range")
    quantiles = np.linspace(cdf_lo, cdf_hi, num=2 * n_sd + 1)
    quantiles /= spectrum.norm_factor
    grid = spectrum.percentiles(quantiles)
    if not np.isfinite(grid).all():
        raise ValueError("non-finite percentile")
    return _between_even_nodes(spectrum, grid, tolerance)


def sample_logarithmic(spectrum, n_sd, size_range=None, tolerance=0.01):
    lo, hi = size_range or spectrum.percentiles(CDF_RANGE)
    grid = np.logspace(np.log10(lo), np.log10(hi), num=2 * n_sd + 1)
    return _between_even_nodes(spectrum, grid, tolerance)
