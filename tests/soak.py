"""soak (not collected by pytest; run by hand on an MI355X: python tests/soak.py): long runs at
full size, HIP fused route against the oracle - the whole 3600-step Shima-2009 experiment at
n_sd = 2^20, 400 adaptive steps, 1000 steps of the Berry breakup box (into its late phase of long
`break_up` loops), 400 of the Straub boxes at 2^18 (the rain spectrum: a third of the collisions
break up) and 60 at 2^21, 40 steps of 32 x 32 cells at 2^20, 600 steps of 16 x 16 cells at 2^18 (deaths in most
steps of the second half).  Everything must agree TO THE BIT, floats included:
both sides take pow / exp / log / erf / ... from csrc/sdm_math.h (round 2 diverged at step 147 of
the rain case - device libm against glibc - profiles/r02_first_divergence_straub_rain.txt)."""
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, ".")
from oracle.engine import OracleEngine  # noqa: E402
from pysdm_amd.cases import make_box  # noqa: E402
from pysdm_amd.engine import HipEngine  # noqa: E402

cases = [("shima", 2**20, False, 3600), ("shima", 2**20, True, 400),
         ("berry_breakup", 2**20, True, 1000), ("straub", 2**18, True, 400),
         ("straub_rain", 2**18, True, 400), ("kinematic2d", 2**20, True, 40),
         # between 2^20 and 2^22: successor words on event tiles of 16384, and deaths in most
         # sub-steps (the compaction from the list of the dead)
         ("straub", 2**21, True, 60),
         # 600 steps of a 16 x 16 grid: from step ~80 on super-droplets die in most time steps
         # (compaction, the closed-form re-sort or the counting sort, the working copy throughout)
         ("kinematic2d", 2**18, True, 600, (16, 16))]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, n_sd, adaptive, steps, *more in cases:
    snaps = []
    for engine in (HipEngine.get(), OracleEngine.get(threads=16)):
        t0 = time.time()
        runner = make_box(engine, name, n_sd=n_sd, adaptive=adaptive,
                          grid=more[0] if more else None)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for chunk in (1, 7, steps - 8):
                runner.run(chunk)
        snaps.append(runner.snapshot())
        print(name, engine.name, round(time.time() - t0, 1), "s", flush=True)
    a, b = snaps
    length = int(a["length"])
    worst = 0.0
    for key, value in a.items():
        ref = b[key]
        if key == "idx":
            value, ref = value[:length], ref[:length]
        if value.dtype.kind == "f":
            live = np.isfinite(ref) & (ref != 0)
            err = np.max(np.abs(value[live] - ref[live]) / np.abs(ref[live])) if live.any() else 0.0
            worst = max(worst, err)
            assert np.array_equal(value, ref, equal_nan=True), (name, key, err)
        else:
            assert np.array_equal(value, ref), (name, key)
    print("OK", name, "length", length, "of", n_sd, "substeps", b["stats_n_substep"][:3],
          "max rel err", worst, flush=True)
