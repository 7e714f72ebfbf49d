import sys, warnings
import numpy as np
sys.path.insert(0, ".")
from oracle.engine import OracleEngine
from pysdm_amd.engine import HipEngine
from pysdm_amd.cases import make_box
cases = {"a": dict(n_sd=65536, grid=(3, 8), dt=5.0, seed=205, opts={}),
         "b": dict(n_sd=40000, grid=(8, 5), dt=200.0, seed=535, opts={"optimized_random": True})}
c = cases[sys.argv[1]]
runs = [make_box(e, "shima", n_sd=c["n_sd"], adaptive=True, dt=c["dt"], thin=0.02, grid=c["grid"], seed=c["seed"], **c["opts"])
        for e in (HipEngine.get(), OracleEngine.get())]
chunks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1] * 6
for step, chunk in enumerate(chunks, 1):
    snaps = []
    for r in runs:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r.run(chunk)
        s = r.snapshot()
        snaps.append((s, r.sub_steps_done, r.offset))
    (a, na, oa), (b, nb, ob) = snaps
    L = int(b["length"])
    print("step", step, "length", int(a["length"]), L, "substeps", na, nb, "offsets", oa == ob,
          {k: bool(np.array_equal(a[k][:L] if k == "idx" else a[k], b[k][:L] if k == "idx" else b[k], equal_nan=True)) for k in b}, flush=True)
    sizes = np.diff(b["cell_start"]); print("   max cell", sizes.max(), "cells", len(sizes))
