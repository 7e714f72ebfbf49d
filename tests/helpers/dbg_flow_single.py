"""debug aid: the one-process side of tests/helpers/dbg_sharded_flow_case.py on both engines, control
words after every collision step.  python tests/helpers/dbg_flow_single.py [timing]"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from oracle.engine import OracleEngine  # noqa: E402
from pysdm_amd import recipe as R  # noqa: E402
from pysdm_amd.collisions import CollisionRunner  # noqa: E402
from pysdm_amd.displacement import DisplacementRunner  # noqa: E402
from pysdm_amd.engine import HipEngine  # noqa: E402
from pysdm_amd.population import Population  # noqa: E402
from tests.displacement_cases import locate  # noqa: E402
from tests.helpers.dbg_sharded_flow_case import CASE as c  # noqa: E402

grid, n_sd, seed = c["grid"], c["n_sd"], c["seed"]
rng = np.random.default_rng(seed)
dims = len(grid)
size = tuple(100.0 * g for g in grid)
positions = rng.uniform(0, 1, (dims, n_sd)) * np.asarray(grid).reshape(dims, 1)
volume = rng.uniform(1e-13, 1e-10, n_sd)
multiplicity = rng.integers(1, 4, n_sd).astype(np.int64)
field = tuple(rng.uniform(-c["courant"], c["courant"],
                          tuple(g + (1 if axis == d else 0) for axis, g in enumerate(grid)))
              for d in range(dims))


def build(engine):
    cell_id, cell_origin, position_in_cell = locate(positions, grid)
    population = Population(engine, multiplicity=multiplicity, volume=volume, cell_id=cell_id,
                            grid=grid, cell_origin=cell_origin, position_in_cell=position_in_cell)
    displacement = DisplacementRunner(
        population, dt=10.0, size=size, enable_sedimentation=c["sedimentation"],
        adaptive=c["adaptive_displacement"], precipitation_counting_level_index=0,
        scheme="ExplicitInSpace" if c["explicit"] else "ImplicitInSpace")
    dv = float(np.prod(np.asarray(size) / np.asarray(grid))) * 1e-6
    runner = CollisionRunner(population, R.CollisionSetup.coalescence(
        R.Geometric(), adaptive=True, seed=seed % 1000), dt=10.0, dv=dv)
    displacement.set_courant(field)
    return population, displacement, runner


hip, oracle = HipEngine.get(), OracleEngine.get()
if len(sys.argv) > 1 and sys.argv[1] == "timing":
    hip.call("sdm_ctx_set_timing", 1)
sides = [(e, *build(e)) for e in (hip, oracle)]
for step in range(1, c["steps"] + 1):
    line = []
    for engine, pop, disp, coll in sides:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            disp.run()
            coll.run(1)
        words = engine.download(pop.ctl)
        stats = np.zeros(8, dtype=np.int64)
        engine.call("sdm_ctx_read_stats", stats, 1)
        perm = engine.download(pop.perm)[: int(words[0])]
        line.append(f"{engine.name}: ctl {words[:4]} flagged at {np.nonzero(perm >= n_sd)[0]} "
                    f"substeps {stats[4]} taken back {stats[5]} n_substep max "
                    f"{engine.download(pop.stats_n_substep).max() if hasattr(pop, 'stats_n_substep') else '-'}")
    print(f"step {step}\n  " + "\n  ".join(line), flush=True)
