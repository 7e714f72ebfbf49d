"""multiplicity spread late in the Berry breakup box (the gamma-long serial loop of `break_up`):
PYTHONPATH=. python profiles/tools/berry_late.py"""
import time

import numpy as np

from pysdm_amd.cases import make_box
from pysdm_amd.engine import HipEngine

engine = HipEngine.get()
runner = make_box(engine, "berry_breakup")
for target in (500, 800, 1000):
    t0 = time.time()
    runner.run(target - runner.steps_done)
    engine.synchronize()
    n = engine.download(runner.population.multiplicity)[runner.population.live_ids()]
    print(target, "steps; wall", round(time.time() - t0, 2), "n_sd", len(n), "mult min/median/max",
          n.min(), np.median(n), n.max(), "ratio", n.max() / n.min(), "substeps",
          engine.download(runner.stats_n_substep))
