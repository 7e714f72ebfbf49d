"""kinematic flow (displacement + collisions, 2^22 super-droplets, 32 x 32 cells): time per step
and, under rocprofv3, the kernel breakdown:  PYTHONPATH=. python profiles/tools/flow_profile.py"""
import time
import warnings

import torch

from pysdm_amd.backends import HIP
from pysdm_amd.examples import make_kinematic_flow

particulator, displacement, collision = make_kinematic_flow(HIP)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    particulator.run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 30
    particulator.run(steps)
    _ = particulator.attributes.super_droplet_count
    torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step; super-droplets left",
      particulator.attributes.super_droplet_count, "substeps/cell",
      collision.stats_n_substep.to_ndarray()[:4])
