"""Moment read-outs of a `Population` on the device: what the reference's
`Particulator.moments` / `spectrum_moments` (PySDM/particulator.py:315-396) hand to the backend
(moments_methods.py:14-182), so that verification needs no copy of whole columns to the host.
fp64 atomics: the order of the adds is free, results agree with the reference to ~1e-12."""
import numpy as np

from .engine import FLOAT


def _column(population, name, law):
    return population.column(name, law)


def moments(population, ranks, *, attr="volume", filter_attr="signed water mass",
            attr_range=(-np.inf, np.inf), weighting_attribute="water mass", weighting_rank=0,
            skip_division_by_m0=False, law=None):
    """(moment_0[n_cell], moments[len(ranks), n_cell]) of `attr` over the live super-droplets whose
    `filter_attr` lies in [attr_range[0], attr_range[1])"""
    eng, pop = population.engine, population
    ranks = np.asarray(ranks, dtype=float)
    if ranks.size == 0:
        raise ValueError("empty specs passed")
    moment_0 = eng.empty(pop.n_cell, FLOAT)
    out = eng.empty((len(ranks), pop.n_cell), FLOAT)
    x_attr = pop.mass if filter_attr in ("signed water mass", "water mass") else _column(
        pop, filter_attr, law)
    eng.call("sdm_moments", moment_0, out, pop.multiplicity, _column(pop, attr, law), pop.cell_id,
             pop.perm, pop.live, eng.upload(ranks), len(ranks), pop.n_cell, float(attr_range[0]),
             float(attr_range[1]), x_attr, _column(pop, weighting_attribute, law),
             float(weighting_rank), int(skip_division_by_m0))
    return eng.download(moment_0), eng.download(out)


def spectrum_moments(population, bin_edges, *, attr="volume", rank=1, bin_attr="water mass",
                     weighting_attribute="water mass", weighting_rank=0, law=None):
    """per (bin, cell): (moment_0, moment of `attr` of order `rank`); a super-droplet falls into
    the first bin with edges[k] <= bin_attr < edges[k + 1]"""
    eng, pop = population.engine, population
    edges = np.asarray(bin_edges, dtype=float)
    n_bins = len(edges) - 1
    moment_0 = eng.empty((n_bins, pop.n_cell), FLOAT)
    out = eng.empty((n_bins, pop.n_cell), FLOAT)
    eng.call("sdm_spectrum_moments", moment_0, out, pop.multiplicity, _column(pop, attr, law),
             pop.cell_id, pop.perm, pop.live, float(rank), eng.upload(edges), n_bins, pop.n_cell,
             _column(pop, bin_attr, law), _column(pop, weighting_attribute, law),
             float(weighting_rank))
    return eng.download(moment_0), eng.download(out)
