"""Shared harness: re-run a golden trajectory (tests/golden/traj_*.npz, produced by the reference)
with a given backend and compare state after the recorded steps."""
import glob
import os
import warnings

import numpy as np

from pysdm_amd import Builder, Formulae
from pysdm_amd.dynamics.collisions import (
    AlwaysN,
    Berry1967,
    Coalescence,
    Collision,
    ConstEb,
    ConstEc,
    Exponential,
    Geometric,
    Golovin,
    Straub2010Ec,
    Straub2010Nf,
)
from pysdm_amd.environments import Box, Mesh

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRIVIA = Formulae().trivia
X0 = TRIVIA.volume(radius=30.531e-6)


def golden_files(pattern):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, pattern)))


def _breakup_parts(name):
    """the plugged parts of tests/golden/gen_golden.py:gen_breakup, by case name"""
    exp = lambda: Exponential(scale=TRIVIA.volume(radius=100e-6))  # noqa: E731
    table = {
        "berry_exp": (Berry1967, exp, "Exponential", False),
        "berry_exp_dt10": (Berry1967, exp, "Exponential", False),
        "berry_exp_while": (Berry1967, exp, "Exponential", True),
        "const_alwaysn": (lambda: ConstEc(Ec=0.3), lambda: AlwaysN(n=4), "AlwaysN", False),
        "straub": (Straub2010Ec, lambda: Straub2010Nf(vmin=X0 * 1e-3, nfmax=10), "Straub2010Nf",
                   False),
        "straub_rain_hab0": (
            Straub2010Ec, lambda: Straub2010Nf(vmin=(0.01e-3) ** 3 * np.pi / 6, nfmax=10000),
            "Straub2010Nf", False),
        "straub_rain_hab1": (
            Straub2010Ec, lambda: Straub2010Nf(vmin=(0.01e-3) ** 3 * np.pi / 6, nfmax=10000),
            "Straub2010Nf", True),
    }
    if name.startswith("rain_") and name[5:] in ("gaussian", "feingold", "slams", "constmass",
                                                   "lowlist"):
        from pysdm_amd.dynamics import collisions as C  # pylint: disable=import-outside-toplevel

        vmin = (0.01e-3) ** 3 * np.pi / 6
        half = lambda: ConstEc(Ec=0.5)  # noqa: E731
        return {  # gen_golden.py:gen_breakup_more
            "gaussian": (half, lambda: C.Gaussian(mu=TRIVIA.volume(radius=0.4e-3),
                                                  sigma=TRIVIA.volume(radius=0.3e-3), vmin=vmin,
                                                  nfmax=100), "Gaussian", False),
            "feingold": (half, lambda: C.Feingold1988(scale=TRIVIA.volume(radius=0.5e-3),
                                                      vmin=vmin, nfmax=100), "Feingold1988",
                         False),
            "slams": (half, lambda: C.SLAMS(vmin=vmin, nfmax=100), "SLAMS", False),
            "constmass": (half, lambda: C.ConstantMass(
                c=float(1000.0 * TRIVIA.volume(radius=0.3e-3))), "ConstantMass", False),
            "lowlist": (C.LowList1982Ec, lambda: C.LowList1982Nf(vmin=vmin, nfmax=100),
                        "LowList1982Nf", False),
        }[name[5:]]
    return table[name]


def setup_from_golden(name, backend_class, fused=None):
    """returns (particulator, dynamic, golden npz, recorded steps)"""
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = gold["cfg"]
    n_sd, seed, adaptive, dt, dv = int(cfg[0]), int(cfg[1]), bool(cfg[2]), cfg[3], cfg[4]
    formulae_kwargs = {"seed": seed}
    env = Box(dt=dt, dv=dv)
    attributes = {"volume": gold["init/volume"], "multiplicity": gold["init/multiplicity"]}
    if name.startswith("traj_golovin"):
        kwargs = {}
        if "global" in name:
            kwargs["croupier"] = "global"
        if "optrand" in name:
            kwargs["optimized_random"] = True
        dynamic = Coalescence(collision_kernel=Golovin(b=cfg[5]), adaptive=adaptive, fused=fused,
                              **kwargs)
    elif name.startswith("traj_geometric"):
        dynamic = Coalescence(collision_kernel=Geometric(collection_efficiency=1),
                              adaptive=adaptive, fused=fused)
    elif name.startswith("traj_kernel"):
        from pysdm_amd.dynamics import collisions as C  # pylint: disable=import-outside-toplevel

        kernel = {"electric": C.Electric, "hydrodynamic": C.Hydrodynamic,
                  "simplegeometric": lambda: C.SimpleGeometric(C=5e7)}[name.split("_")[-1]]()
        dynamic = Coalescence(collision_kernel=kernel, adaptive=adaptive, fused=fused)
    elif name.startswith("traj_multicell"):
        grid = tuple(int(g) for g in gold["grid"])
        env.mesh = Mesh(grid, size=tuple(float(g) for g in grid))
        env.mesh.dv = dv
        attributes["cell id"] = gold["init/cell_id"]
        kern = Golovin(b=1.5e3) if "golovin" in name else Geometric(collection_efficiency=1)
        dynamic = Coalescence(collision_kernel=kern, adaptive=adaptive,
                              optimized_random=bool(cfg[6]), fused=fused)
    elif name.startswith("traj_breakup"):
        ec, frag, fname, hab = _breakup_parts(name[len("traj_breakup_"):])
        formulae_kwargs.update(fragmentation_function=fname, handle_all_breakups=hab)
        dynamic = Collision(collision_kernel=Geometric(), coalescence_efficiency=ec(),
                            breakup_efficiency=ConstEb(1.0), fragmentation_function=frag(),
                            adaptive=True, warn_overflows=False, fused=fused)
    else:
        raise ValueError(name)
    backend = backend_class(Formulae(**formulae_kwargs))
    builder = Builder(n_sd=n_sd, backend=backend, environment=env)
    builder.add_dynamic(dynamic)
    particulator = builder.build(attributes)
    dynamic = particulator.dynamics["Collision"]  # the built copy (builder.py:130-131)
    steps = sorted({int(k.split("/")[0][4:]) for k in gold.files if k.startswith("step")})
    return particulator, dynamic, gold, steps


def snapshot(particulator, dynamic):
    attrs = particulator.attributes
    idx = attrs._fused_view()["idx"]  # pylint: disable=protected-access
    snap = {
        "idx": idx.to_ndarray(),
        "length": np.asarray(len(idx)),
        "multiplicity": attrs["multiplicity"].to_ndarray(raw=True),
        "attributes": attrs.get_extensive_attribute_storage().to_ndarray(raw=True),
        "cell_start": attrs.cell_start.to_ndarray(),
        "collision_rate": dynamic.collision_rate.to_ndarray(),
        "collision_rate_deficit": dynamic.collision_rate_deficit.to_ndarray(),
        "coalescence_rate": dynamic.coalescence_rate.to_ndarray(),
        "stats_n_substep": dynamic.stats_n_substep.to_ndarray(),
        "stats_dt_min": dynamic.stats_dt_min.to_ndarray(),
    }
    if dynamic.enable_breakup:
        snap["breakup_rate"] = dynamic.breakup_rate.to_ndarray()
        snap["breakup_rate_deficit"] = dynamic.breakup_rate_deficit.to_ndarray()
    return snap


INT_KEYS = ("idx", "length", "multiplicity", "cell_start", "collision_rate",
            "collision_rate_deficit", "coalescence_rate", "stats_n_substep", "breakup_rate",
            "breakup_rate_deficit")


def compare(snap, gold, step, float_rtol=0.0, idx_tail=True):
    """ints bit-exact; floats within float_rtol (0 = bit-exact).  The raw slots of removed
    super-droplets are compared too.  `idx` beyond `length` is dead storage whose content depends
    on the caretaker's buffer-swap history (the reference leaves stale values there): it is
    compared only when `idx_tail` (the method-by-method route reproduces even that)."""
    length = int(snap["length"])
    for key, value in snap.items():
        ref = gold[f"step{step}/{key}"]
        if key == "idx" and not idx_tail:
            value, ref = value[:length], ref[:length]
        if key in INT_KEYS:
            np.testing.assert_array_equal(value, ref, err_msg=f"step {step}: {key}")
        elif float_rtol == 0.0:
            np.testing.assert_array_equal(value, ref, err_msg=f"step {step}: {key}")
        else:
            np.testing.assert_allclose(value, ref, rtol=float_rtol, atol=0,
                                       err_msg=f"step {step}: {key}")


def run_and_compare(name, backend_class, fused=None, float_rtol=0.0, max_step=None):
    particulator, dynamic, gold, steps = setup_from_golden(name, backend_class, fused=fused)
    for step in steps:
        if max_step is not None and step > max_step:
            break
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            particulator.run(step - particulator.n_steps)
        compare(snapshot(particulator, dynamic), gold, step, float_rtol=float_rtol,
                idx_tail=fused is False or not hasattr(particulator.backend, "collision_step"))
    return particulator, dynamic
