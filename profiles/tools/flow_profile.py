"""kinematic flow (displacement + collisions, 2^22 super-droplets, 32 x 32 cells): time per step
and, under rocprofv3, the kernel breakdown:  PYTHONPATH=. python profiles/tools/flow_profile.py
[n_sd nx ny]   (e.g. 720000 75 75: the grid of the reference's 2-D example, 128 per cell)"""
import sys
import time
import warnings

import torch

from pysdm_amd.cases import make_kinematic_flow
from pysdm_amd.engine import HipEngine

options = {}
if len(sys.argv) > 3:
    options = dict(n_sd=int(sys.argv[1]), grid=(int(sys.argv[2]), int(sys.argv[3])))
displacement, collisions = make_kinematic_flow(HipEngine.get(), **options)


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
