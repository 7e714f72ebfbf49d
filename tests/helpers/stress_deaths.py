"""stress for the death path of multi-cell adaptive runs (compaction inside the sub-step kernel,
re-sort, launch-ahead, working copy): many seeds, several steps per call, HIP against the checker"""
import sys, warnings
import numpy as np
sys.path.insert(0, ".")
from oracle.engine import OracleEngine
from pysdm_amd.engine import HipEngine
from pysdm_amd.cases import make_box
hip, oracle = HipEngine.get(), OracleEngine.get()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for seed in range(n):
    grid = [(8, 5), (3, 8), (5, 4), (4, 4)][seed % 4]
    n_sd = [40000, 65536, 20000, 8192][seed % 4]
    chunks = [[5], [2, 2, 1], [5, 8, 2], [3, 4]][seed % 4]
    snaps = []
    for e in (hip, oracle):
        r = make_box(e, "shima", n_sd=n_sd, adaptive=True, dt=200.0, thin=0.02, grid=grid, seed=1000 + seed,
                     optimized_random=bool(seed % 2))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for c in chunks:
                r.run(c)
        snaps.append((r.snapshot(), r.sub_steps_done, r.offset))
    (a, na, oa), (b, nb, ob) = snaps
    L = int(b["length"])
    ok = na == nb and oa == ob and all(np.array_equal(a[k][:L] if k == "idx" else a[k], b[k][:L] if k == "idx" else b[k], equal_nan=True) for k in b)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, grid, n_sd, chunks, na, nb, flush=True)
print("done", n, "cases,", bad, "mismatches")
