"""time per step of the method-by-method route (what PySDM's own dynamics drive):
PYTHONPATH=. python profiles/tools/method_route.py [workload] [n_sd] [steps]"""
import sys
import time

import torch

from pysdm_amd.backends import HIP
from pysdm_amd.examples import make_box

name = sys.argv[1] if len(sys.argv) > 1 else "shima"
n_sd = int(sys.argv[2]) if len(sys.argv) > 2 else 2**20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for fused in (False, None):
    particulator, _ = make_box(HIP, name, n_sd=n_sd, fused=fused)
    particulator.run(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    particulator.run(steps)
    _ = particulator.attributes.super_droplet_count
    torch.cuda.synchronize()
    print(name, n_sd, "fused" if fused is None else "methods",
          f"{(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
