"""Shared harness: re-run a golden trajectory (tests/golden/traj_*.npz, produced by the reference)
on a given engine and route and compare the state after the recorded steps."""
import glob
import os
import warnings

import numpy as np

from pysdm_amd import recipe as R
from pysdm_amd.collisions import CollisionRunner
from pysdm_amd.physics import constants as const
from pysdm_amd.population import Population

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def volume_of_radius(radius):
    return const.PI_4_3 * np.power(radius, 3)


X0 = volume_of_radius(30.531e-6)
RAIN_VMIN = (0.01e-3) ** 3 * np.pi / 6


def golden_files(pattern):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, pattern)))


def breakup_parts(name):
    """(coalescence efficiency, fragmentation, handle_all_breakups) of the breakup goldens
    (tests/golden/gen_golden.py: gen_breakup, gen_breakup_more), by case name"""
    exp = R.Exponential(scale=volume_of_radius(100e-6))
    half = R.ConstEc(Ec=0.5)
    rain = R.Straub2010Nf(vmin=RAIN_VMIN, nfmax=10000)
    return {
        "berry_exp": (R.Berry1967(), exp, False),
        "berry_exp_dt10": (R.Berry1967(), exp, False),
        "berry_exp_while": (R.Berry1967(), exp, True),
        "const_alwaysn": (R.ConstEc(Ec=0.3), R.AlwaysN(n=4), False),
        "straub": (R.Straub2010Ec(), R.Straub2010Nf(vmin=X0 * 1e-3, nfmax=10), False),
        "straub_rain_hab0": (R.Straub2010Ec(), rain, False),
        "straub_rain_hab1": (R.Straub2010Ec(), rain, True),
        "rain_gaussian": (half, R.Gaussian(mu=volume_of_radius(0.4e-3),
                                           sigma=volume_of_radius(0.3e-3), vmin=RAIN_VMIN,
                                           nfmax=100), False),
        "rain_feingold": (half, R.Feingold1988(scale=volume_of_radius(0.5e-3), vmin=RAIN_VMIN,
                                               nfmax=100), False),
        "rain_slams": (half, R.SLAMS(vmin=RAIN_VMIN, nfmax=100), False),
        "rain_constmass": (half, R.ConstantMass(c=float(1000.0 * volume_of_radius(0.3e-3))),
                           False),
        "rain_lowlist": (R.LowList1982Ec(), R.LowList1982Nf(vmin=RAIN_VMIN, nfmax=100), False),
    }[name]


def setup_from_golden(name, engine, route="fused"):
    """returns (runner, golden npz, recorded steps)"""
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = gold["cfg"]
    n_sd, seed, adaptive, dt, dv = int(cfg[0]), int(cfg[1]), bool(cfg[2]), cfg[3], cfg[4]
    cell_id, grid = None, None
    options = {"seed": seed, "adaptive": adaptive}
    if name.startswith("traj_golovin"):
        if "global" in name:
            options["croupier"] = "global"
        if "optrand" in name:
            options["optimized_random"] = True
        setup = R.CollisionSetup.coalescence(R.Golovin(b=cfg[5]), **options)
    elif name.startswith("traj_geometric"):
        setup = R.CollisionSetup.coalescence(R.Geometric(collection_efficiency=1), **options)
    elif name.startswith("traj_kernel"):
        kernel = {"electric": R.Electric(), "hydrodynamic": R.Hydrodynamic(),
                  "simplegeometric": R.SimpleGeometric(C=5e7)}[name.split("_")[-1]]
        setup = R.CollisionSetup.coalescence(kernel, **options)
    elif name.startswith("traj_multicell"):
        grid = tuple(int(g) for g in gold["grid"])
        cell_id = gold["init/cell_id"]
        kernel = R.Golovin(b=1.5e3) if "golovin" in name else R.Geometric(collection_efficiency=1)
        if "global" in name:
            options["croupier"] = "global"
        setup = R.CollisionSetup.coalescence(kernel, optimized_random=bool(cfg[6]), **options)
    elif name.startswith("traj_breakup"):
        ec, frag, hab = breakup_parts(name[len("traj_breakup_"):])
        setup = R.CollisionSetup.collision(R.Geometric(), ec, R.ConstEb(1.0), frag, seed=seed,
                                           adaptive=True, warn_overflows=False,
                                           handle_all_breakups=hab)
    else:
        raise ValueError(name)
    population = Population(engine, multiplicity=gold["init/multiplicity"],
                            volume=gold["init/volume"], cell_id=cell_id, grid=grid)
    runner = CollisionRunner(population, setup, dt=dt, dv=dv, route=route)
    steps = sorted({int(k.split("/")[0][4:]) for k in gold.files if k.startswith("step")})
    return runner, gold, steps


INT_KEYS = ("idx", "length", "multiplicity", "cell_start", "collision_rate",
            "collision_rate_deficit", "coalescence_rate", "stats_n_substep", "breakup_rate",
            "breakup_rate_deficit")


def compare(snap, gold, step, float_rtol=0.0, idx_tail=True):
    """ints bit-exact; floats within float_rtol (0 = bit-exact).  The raw slots of removed
    super-droplets are compared too.  `idx` beyond `length` is dead storage whose content depends
    on the buffer-swap history of the counting sort (the reference leaves stale values there): it
    is compared only when `idx_tail` (the stage-by-stage routes reproduce even that)."""
    length = int(snap["length"])
    for key, value in snap.items():
        ref = gold[f"step{step}/{key}"]
        if key == "idx" and not idx_tail:
            value, ref = value[:length], ref[:length]
        if key in INT_KEYS or float_rtol == 0.0:
            np.testing.assert_array_equal(value, ref, err_msg=f"step {step}: {key}")
        else:
            np.testing.assert_allclose(value, ref, rtol=float_rtol, atol=0,
                                       err_msg=f"step {step}: {key}")


def run_and_compare(name, engine, route="fused", float_rtol=0.0, max_step=None):
    runner, gold, steps = setup_from_golden(name, engine, route=route)
    # the HIP fused step does not reproduce the dead tail of the permutation buffer
    idx_tail = not (engine.name == "hip" and route == "fused")
    for step in steps:
        if max_step is not None and step > max_step:
            break
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            runner.run(step - runner.steps_done)
        compare(runner.snapshot(), gold, step, float_rtol=float_rtol, idx_tail=idx_tail)
    return runner
