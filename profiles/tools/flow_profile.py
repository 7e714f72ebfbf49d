"""kinematic flow (displacement + collisions, 2^22 super-droplets, 32 x 32 cells): time per step
and, under rocprofv3, the kernel breakdown:  PYTHONPATH=. python profiles/tools/flow_profile.py"""
import time
import warnings

import torch

from pysdm_amd.cases import make_kinematic_flow
from pysdm_amd.engine import HipEngine

displacement, collisions = make_kinematic_flow(HipEngine.get())


def run(steps):
    for _ in range(steps):
        displacement.run()
        collisions.run(1)


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 30
    run(steps)
    torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step; super-droplets left",
      collisions.population.live, "substeps/cell",
      collisions.engine.download(collisions.stats_n_substep)[:4])
