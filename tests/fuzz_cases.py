"""The randomised differential cases of tests/fuzz_parity.py and tests/fuzz_displacement.py as
functions: drawn from a seeded generator, run on the product and on the checker, compared to the
bit (positions and masses of the displacement to 1e-12).  The hand-run scripts loop over thousands
of them on an MI355X; `tests/test_hip_fuzz.py` runs a fixed-seed slice under `-m gpu` (round 3:
the fuzzers found three bugs that no fixed case had, and the driver never ran them)."""
import warnings

import numpy as np

from pysdm_amd import recipe as R
from pysdm_amd.cases import make_box
from pysdm_amd.collisions import CollisionRunner
from pysdm_amd.displacement import DisplacementRunner
from pysdm_amd.population import Population, locate


def draw_parity_case(rng):  # pylint: disable=too-many-branches
    """random small-to-medium collision set-up: grid or single cell, adaptive or not, thin
    multiplicities (deaths) or not, coalescence / breakup recipes, steps given as random chunks
    (so that the working copy, the launch-ahead and the step-by-step paths all get their turn)"""
    name = str(rng.choice(["shima", "kinematic2d", "berry_breakup", "straub", "straub_rain"]))
    grid = None
    if rng.random() < 0.6:
        grid = tuple(int(g) for g in rng.choice([2, 3, 4, 5, 8], size=2))
    cells = 1 if grid is None else grid[0] * grid[1]
    per_cell = int(rng.choice([3, 17, 64, 300, 1000, 3000]))
    n_sd = (max(2, min(per_cell * cells, 2**16)) if grid
            else int(rng.choice([2, 3, 257, 4096, 2**14])))
    adaptive = bool(rng.random() < 0.7)
    thin = 0.02 if (name == "shima" and rng.random() < 0.5) else None
    options = {}
    if not adaptive:
        options["substeps"] = int(rng.choice([1, 2, 3]))
    if rng.random() < 0.3 and name != "kinematic2d":  # (that configuration sets it itself)
        options["optimized_random"] = True
    # (the global croupier over several cells only without adaptive sub-stepping: with it the
    # reference duplicates ids once a working length is cut - tests/test_hip_parity.py says why -
    # and a serial and a parallel run then differ legitimately; "kinematic2d" brings its own grid)
    if (grid is None and name != "kinematic2d" or not adaptive) and rng.random() < 0.25:
        options["croupier"] = "global"
    if name in ("berry_breakup", "straub", "straub_rain"):
        if rng.random() < 0.3:
            options["handle_all_breakups"] = True
        if rng.random() < 0.2:
            options["max_multiplicity"] = int(rng.choice([10**7, 10**9, 10**12]))
    if adaptive and rng.random() < 0.3:
        options["dt_range"] = tuple(float(v) for v in rng.choice(
            [(0.1, 100.0), (0.5, 2.0), (1.0, 1.0), (0.01, 0.5)]))
    dt = float(rng.choice([1.0, 5.0, 50.0, 200.0])) if name in ("shima", "kinematic2d") else None
    # (the stage-by-stage route - one ABI symbol per backend method - on the smaller set-ups)
    route = "chain" if (n_sd <= 4096 and rng.random() < 0.35) else "fused"
    chunks = [int(c) for c in rng.choice([1, 2, 3, 5, 8], size=int(rng.integers(1, 4)))]
    seed = int(rng.integers(1, 1000))
    return {"name": name, "n_sd": n_sd, "grid": grid, "adaptive": adaptive, "thin": thin,
            "dt": dt, "options": options, "chunks": chunks, "seed": seed, "route": route}


def run_parity_case(product, checker, case):
    """-> "ok" or "refused" (a combination the set-up itself refuses); raises on a difference"""
    label = str(case)
    snaps = []
    try:
        for engine in (product, checker):
            runner = make_box(engine, case["name"], n_sd=case["n_sd"], adaptive=case["adaptive"],
                              dt=case["dt"], thin=case["thin"], grid=case["grid"],
                              seed=case["seed"], route=case["route"], **case["options"])
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for steps in case["chunks"]:
                    runner.run(steps)
            snaps.append((runner.snapshot(), runner.offset, runner.offset_breakup,
                          runner.sub_steps_done))
    except (ValueError, NotImplementedError) as refused:
        if "Radii can be interpolated" not in str(refused) and "dt_range" not in str(refused):
            raise
        return "refused"
    a, b = snaps
    assert a[1:] == b[1:], (label, a[1:], b[1:])
    length = int(b[0]["length"])
    for key, ref in b[0].items():
        value = a[0][key]
        if key == "idx":
            value, ref = value[:length], ref[:length]
        assert np.array_equal(value, ref, equal_nan=True), (label, key)
    return "ok"


def draw_displacement_case(rng):
    """random grid in 1-3 dimensions, random (moderate) Courant field, sedimentation on / off, both
    advection schemes, both routes, optionally adaptive coalescence after every displacement step"""
    n_dims = int(rng.integers(1, 4))
    grid = tuple(int(g) for g in rng.integers(2, 9, size=n_dims))
    n_sd = int(rng.choice([16, 300, 5000, 40000]))
    case = {"grid": grid, "n_sd": n_sd, "sedimentation": bool(rng.random() < 0.5),
            "scheme": str(rng.choice(["ImplicitInSpace", "ExplicitInSpace"])),
            "route": str(rng.choice(["fused", "chain"])), "collide": bool(rng.random() < 0.4),
            "adaptive": bool(rng.random() < 0.7), "steps": int(rng.integers(1, 5))}
    case["positions"] = rng.uniform(0, 1, (n_dims, n_sd)) * np.asarray(grid).reshape(-1, 1)
    case["volume"] = rng.exponential(4 / 3 * np.pi * (30e-6) ** 3, n_sd) + 1e-18
    case["multiplicity"] = rng.integers(1, 10**6, n_sd)
    case["courant"] = tuple(rng.uniform(-0.45, 0.45, tuple(g + (1 if a == d else 0)
                                                           for a, g in enumerate(grid)))
                            for d in range(n_dims))
    return case


def run_displacement_case(product, checker, case):
    grid = case["grid"]
    size = tuple(float(g) * 100.0 for g in grid)
    label = str({k: v for k, v in case.items()
                 if k not in ("positions", "volume", "multiplicity", "courant")})
    results = []
    for engine in (product, checker):
        cell_id, origin, within = locate(case["positions"], grid)
        pop = Population(engine, multiplicity=case["multiplicity"].copy(),
                         volume=case["volume"].copy(), cell_id=cell_id, grid=grid,
                         cell_origin=origin, position_in_cell=within)
        disp = DisplacementRunner(pop, dt=1.0, size=size,
                                  enable_sedimentation=case["sedimentation"],
                                  adaptive=case["adaptive"], scheme=case["scheme"],
                                  route=case["route"])
        disp.set_courant(case["courant"])
        coll = None
        if case["collide"]:
            dv = float(np.prod(np.asarray(size) / np.asarray(grid)))
            coll = CollisionRunner(pop, R.CollisionSetup.coalescence(R.Geometric(), adaptive=True,
                                                                     seed=44), dt=1.0, dv=dv,
                                   route=case["route"])
        rain = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(case["steps"]):
                rain.append(disp.run())
                if coll is not None:
                    coll.run(1)
        pop.compact()
        down = engine.download
        live = down(pop.perm)[: pop.live]
        results.append((pop.live, live, down(pop.cell_origin)[:, live], down(pop.cell_id)[live],
                        down(pop.multiplicity)[live], down(pop.position_in_cell)[:, live],
                        down(pop.mass)[live], np.asarray(rain)))
    a, b = results
    assert a[0] == b[0], (label, a[0], b[0])
    for k in (1, 2, 3, 4):
        assert np.array_equal(a[k], b[k]), (label, k)
    for k in (5, 6, 7):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-12, atol=1e-13, err_msg=label)
    return a[0]
