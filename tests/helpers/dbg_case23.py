import sys, warnings
import numpy as np
sys.path.insert(0, ".")
from oracle.engine import OracleEngine
from pysdm_amd import recipe as R, cases
from pysdm_amd.collisions import CollisionRunner
from pysdm_amd.population import Population
from pysdm_amd.engine import HipEngine
hip, oracle = HipEngine.get(), OracleEngine.get()
volume, multiplicity, _, dv, _ = cases.initial_state("kinematic2d", 4096)
def run(engine, setup, steps):
    pop = Population(engine, multiplicity=multiplicity.copy(), volume=volume.copy())
    r = CollisionRunner(pop, setup, dt=5.0, dv=dv)
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(steps):
            r.run(1); out.append((r.snapshot(), r.offset, r.sub_steps_done))
    return out
for croupier in ("global", "local"):
    for adaptive in (True, False):
        for opt in (False, True):
            for kernel in (R.Geometric(collection_efficiency=1), R.Golovin(b=1.5e3)):
                setup = R.CollisionSetup.coalescence(kernel, adaptive=adaptive, croupier=croupier, optimized_random=opt, seed=406)
                a = run(hip, setup, 3); b = run(oracle, setup, 3)
                res = []
                for (sa, oa, na), (sb, ob, nb) in zip(a, b):
                    L = int(sb["length"])
                    res.append((bool(np.array_equal(sa["idx"][:L], sb["idx"][:L])), bool(np.array_equal(sa["multiplicity"], sb["multiplicity"])), oa == ob, na, nb, int((sb["collision_rate"]).sum())))
                print(croupier, "adaptive", adaptive, "opt", opt, type(kernel).__name__, res, flush=True)
