"""randomised differential run of the displacement step (+ collisions behind it), by hand on an
MI355X like tests/fuzz_parity.py:  python tests/fuzz_displacement.py [n_cases] [seed]
random grids in 1-3 dimensions, random (moderate) Courant fields, sedimentation on / off, both
advection schemes, both routes of this package (fused `sdm_displacement_step`, stage by stage),
optionally adaptive coalescence after every displacement step - HIP against the checker: cell
origins, cell ids, permutation, multiplicities exact; positions and masses to 1e-12."""
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, ".")
from oracle.engine import OracleEngine  # noqa: E402
from pysdm_amd import recipe as R  # noqa: E402
from pysdm_amd.collisions import CollisionRunner  # noqa: E402
from pysdm_amd.displacement import DisplacementRunner  # noqa: E402
from pysdm_amd.engine import HipEngine  # noqa: E402
from pysdm_amd.population import Population, locate  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
hip, oracle = HipEngine.get(), OracleEngine.get()
t_start = time.time()
for case in range(n_cases):
    n_dims = int(rng.integers(1, 4))
    grid = tuple(int(g) for g in rng.integers(2, 9, size=n_dims))
    size = tuple(float(g) * 100.0 for g in grid)
    n_sd = int(rng.choice([16, 300, 5000, 40000]))
    sedimentation = bool(rng.random() < 0.5)
    scheme = str(rng.choice(["ImplicitInSpace", "ExplicitInSpace"]))
    route = str(rng.choice(["fused", "chain"]))
    collide = bool(rng.random() < 0.4)
    adaptive = bool(rng.random() < 0.7)
    steps = int(rng.integers(1, 5))
    positions = rng.uniform(0, 1, (n_dims, n_sd)) * np.asarray(grid).reshape(-1, 1)
    volume = rng.exponential(4 / 3 * np.pi * (30e-6) ** 3, n_sd) + 1e-18
    multiplicity = rng.integers(1, 10**6, n_sd)
    courant = tuple(rng.uniform(-0.45, 0.45, tuple(g + (1 if a == d else 0)
                                                   for a, g in enumerate(grid)))
                    for d in range(n_dims))
    label = (f"case {case}: grid={grid} n_sd={n_sd} sedimentation={sedimentation} {scheme} "
             f"route={route} collide={collide} adaptive={adaptive} steps={steps}")
    results = []
    for engine in (hip, oracle):
        cell_id, origin, within = locate(positions, grid)
        pop = Population(engine, multiplicity=multiplicity.copy(), volume=volume.copy(),
                         cell_id=cell_id, grid=grid, cell_origin=origin, position_in_cell=within)
        disp = DisplacementRunner(pop, dt=1.0, size=size, enable_sedimentation=sedimentation,
                                  adaptive=adaptive, scheme=scheme, route=route)
        disp.set_courant(courant)
        coll = None
        if collide:
            dv = float(np.prod(np.asarray(size) / np.asarray(grid)))
            coll = CollisionRunner(pop, R.CollisionSetup.coalescence(R.Geometric(), adaptive=True,
                                                                     seed=44), dt=1.0, dv=dv,
                                   route=route)
        rain = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(steps):
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
    print(label, "-> ok, left", a[0], flush=True)
print("all cases equal the checker;", round(time.time() - t_start, 1), "s")
