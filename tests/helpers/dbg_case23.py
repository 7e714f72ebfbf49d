import sys, warnings
import numpy as np
sys.path.insert(0, ".")
from oracle.engine import OracleEngine
from pysdm_amd.engine import HipEngine
from tests.trajectory import setup_from_golden
name = sys.argv[1] if len(sys.argv) > 1 else "traj_multicell_geometric_4x4_global"
route = sys.argv[2] if len(sys.argv) > 2 else "chain"
runs = []
for engine in (HipEngine.get(), OracleEngine.get()):
    runner, gold, steps = setup_from_golden(name, engine, route=route)
    runs.append(runner)
for step in range(1, 11):
    snaps = []
    for r in runs:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r.run(1)
        pop = r.population
        snaps.append((r.engine.download(pop.perm), r.engine.download(pop.perm_spare), pop.live, pop.working, r.sub_steps_done,
                      r.engine.download(pop.multiplicity)))
    a, b = snaps
    d = np.nonzero(a[0] != b[0])[0]
    d2 = np.nonzero(a[1] != b[1])[0]
    print("step", step, "live", a[2], b[2], "substeps", a[4], b[4], "perm diffs", len(d), d[:12], "spare diffs", len(d2), d2[:12],
          "mult equal", np.array_equal(a[5], b[5]), flush=True)
    if len(d):
        print("  hip", a[0][d[:12]], "oracle", b[0][d[:12]])
