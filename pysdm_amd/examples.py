"""The box set-ups of BASELINE.json's configurations, built from this package's parts.

Parameters follow the reference's example settings (examples/PySDM_examples/):
Shima_et_al_2009/settings.py:14-33, Berry_1967/settings.py:14-47 with the breakup set-up of
deJong_Mackay_et_al_2023/settings_0D.py:21-52, the Straub case of
tests/smoke_tests/box/dejong_and_mackay_et_al_2023/test_fig_8.py:20-31, and a collisions-only
slice of Arabas_et_al_2015 (32 x 32 cells, 4096 super-droplets per cell, no advection).
"""
import numpy as np

from . import Builder, Formulae
from .dynamics.collisions import (
    Berry1967,
    Coalescence,
    Collision,
    ConstEb,
    Exponential,
    Geometric,
    Golovin,
    Straub2010Ec,
    Straub2010Nf,
)
from .environments import Box, Mesh
from .initialisation import ConstantMultiplicity
from .initialisation import Exponential as ExponentialSpectrum

TRIVIA = Formulae().trivia

CONFIGS = {
    # configs[0] / configs[1]: Shima 2009 box, Golovin kernel
    "shima": dict(n_sd=2**20, n_part=2**23, dv=1e6, radius=30.531e-6, adaptive=False,
                  make=lambda adaptive, fused: Coalescence(
                      collision_kernel=Golovin(b=1.5e3), adaptive=adaptive, fused=fused)),
    # configs[2]: Berry 1967 box, geometric kernel + breakup
    "berry_breakup": dict(
        n_sd=2**20, n_part=239e6, dv=10.0 * 2**20 / 2**13, radius=10e-6, adaptive=True,
        formulae_kwargs={"fragmentation_function": "Exponential"},
        make=lambda adaptive, fused: Collision(
            collision_kernel=Geometric(), coalescence_efficiency=Berry1967(),
            breakup_efficiency=ConstEb(1.0),
            fragmentation_function=Exponential(scale=TRIVIA.volume(radius=100e-6)),
            adaptive=adaptive, warn_overflows=False, fused=fused)),
    # configs[3]: 32 x 32 cells, 2^22 super-droplets, geometric kernel, adaptive, optimized_random
    "kinematic2d": dict(
        n_sd=2**22, n_part=239e6, dv=2197.0 * 1024, radius=15e-6, dt=5.0, grid=(32, 32),
        adaptive=True,
        make=lambda adaptive, fused: Coalescence(
            collision_kernel=Geometric(collection_efficiency=1), adaptive=adaptive,
            optimized_random=True, fused=fused)),
    # configs[4]: Straub 2010 breakup + geometric kernel
    "straub": dict(
        n_sd=2**22, n_part=100e6, dv=1.0 * 2**22 / 2**10, radius=30.531e-6, adaptive=True,
        formulae_kwargs={"fragmentation_function": "Straub2010Nf"},
        make=lambda adaptive, fused: Collision(
            collision_kernel=Geometric(), coalescence_efficiency=Straub2010Ec(),
            breakup_efficiency=ConstEb(1.0),
            fragmentation_function=Straub2010Nf(
                vmin=TRIVIA.volume(radius=30.531e-6) * 1e-3, nfmax=10),
            adaptive=adaptive, warn_overflows=False, fused=fused)),
}


def make_box(backend_class, name, *, n_sd=None, adaptive=None, fused=None, seed=44, dt=None,
             cell_block=None):
    """returns (particulator, dynamic) for configuration `name`; `cell_block` = (first, last)
    restricts a multi-cell configuration to a contiguous block of cells (one rank's shard)"""
    cfg = dict(CONFIGS[name])
    n_sd = n_sd or cfg["n_sd"]
    adaptive = cfg["adaptive"] if adaptive is None else adaptive
    formulae = Formulae(seed=seed, **cfg.get("formulae_kwargs", {}))
    # real-droplet concentration stays that of the configuration when n_sd is rescaled
    dv = cfg["dv"] * n_sd / cfg["n_sd"]
    spectrum = ExponentialSpectrum(norm_factor=cfg["n_part"] * dv,
                                   scale=TRIVIA.volume(radius=cfg["radius"]))
    volume, multiplicity = ConstantMultiplicity(spectrum).sample(n_sd)
    env = Box(dt=dt or cfg.get("dt", 1.0), dv=dv)
    attributes = {"volume": volume, "multiplicity": multiplicity}
    if "grid" in cfg:
        grid = cfg["grid"]
        n_cell = int(np.prod(grid))
        rng = np.random.default_rng(7)
        attributes["cell id"] = rng.integers(0, n_cell, size=n_sd).astype(np.int64)
        order = rng.permutation(n_sd)
        attributes["volume"], attributes["multiplicity"] = volume[order], multiplicity[order]
        if cell_block is not None:
            first, last = cell_block
            mine = np.flatnonzero((attributes["cell id"] >= first) & (attributes["cell id"] < last))
            attributes = {k: v[mine] for k, v in attributes.items()}
            attributes["cell id"] = attributes["cell id"] - first
            grid, n_cell_local = (last - first,), last - first
            n_sd = len(mine)
        else:
            n_cell_local = n_cell
        env.mesh = Mesh(grid, size=tuple(float(g) for g in grid))
        env.mesh.dv = dv / n_cell
        assert env.mesh.n_cell == n_cell_local
    builder = Builder(n_sd=n_sd, backend=backend_class(formulae), environment=env)
    dynamic = cfg["make"](adaptive, fused)
    builder.add_dynamic(dynamic)
    particulator = builder.build(attributes)
    return particulator, particulator.dynamics["Collision"]


def single_eddy_courant(grid, size, dt, w_max=0.6):
    """Courant numbers of the non-divergent single-eddy flow of the 2-D kinematic set-up
    (stream function psi = -w_max X/pi sin(pi z/Z) cos(2 pi x/X), as in
    examples/PySDM_examples/Szumowski_et_al_1998/.., Arabas_et_al_2015) on the Arakawa-C faces:
    differences of psi at the cell corners, so the discrete divergence vanishes"""
    (n_x, n_z), (len_x, len_z) = grid, size
    d_x, d_z = len_x / n_x, len_z / n_z
    x = np.linspace(0, len_x, n_x + 1).reshape(-1, 1)
    z = np.linspace(0, len_z, n_z + 1).reshape(1, -1)
    psi = -w_max * len_x / np.pi * np.sin(np.pi * z / len_z) * np.cos(2 * np.pi * x / len_x)
    courant_x = -(psi[:, 1:] - psi[:, :-1]) / d_z * dt / d_x  # (n_x + 1, n_z)
    courant_z = (psi[1:, :] - psi[:-1, :]) / d_x * dt / d_z   # (n_x, n_z + 1)
    return courant_x, courant_z


def make_kinematic_flow(backend_class, *, n_sd=2**22, grid=(32, 32), size=(1500.0, 1500.0), dt=5.0,
                        fused=None, seed=44, sedimentation=True):
    """configs[3] with the step that precedes collisions in the 2-D kinematic set-up:
    `Displacement` (single-eddy flow + sedimentation) followed by adaptive Geometric coalescence.
    Returns (particulator, displacement, collision)"""
    from .dynamics.displacement import Displacement  # pylint: disable=import-outside-toplevel

    cfg = CONFIGS["kinematic2d"]
    formulae = Formulae(seed=seed)
    n_cell = int(np.prod(grid))
    env = Box(dt=dt, dv=None)
    env.mesh = Mesh(grid, size)
    dv_total = env.mesh.dv * n_cell
    spectrum = ExponentialSpectrum(norm_factor=cfg["n_part"] * dv_total * n_sd / cfg["n_sd"],
                                   scale=TRIVIA.volume(radius=cfg["radius"]))
    volume, multiplicity = ConstantMultiplicity(spectrum).sample(n_sd)
    rng = np.random.default_rng(7)
    order = rng.permutation(n_sd)
    positions = rng.uniform(0, 1, (2, n_sd)) * np.asarray(grid).reshape(2, 1)
    cell_id, cell_origin, position_in_cell = env.mesh.cellular_attributes(positions)
    builder = Builder(n_sd=n_sd, backend=backend_class(formulae), environment=env)
    builder.add_dynamic(Displacement(enable_sedimentation=sedimentation))
    builder.add_dynamic(cfg["make"](True, fused))
    particulator = builder.build({
        "volume": volume[order], "multiplicity": multiplicity[order], "cell id": cell_id,
        "cell origin": cell_origin, "position in cell": position_in_cell,
    })
    displacement = particulator.dynamics["Displacement"]
    displacement.upload_courant_field(single_eddy_courant(grid, size, dt))
    return particulator, displacement, particulator.dynamics["Collision"]
