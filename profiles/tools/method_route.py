"""time per step of the stage-by-stage ("chain") route - what a front-end drives method by method -
against the fused one:  PYTHONPATH=. python profiles/tools/method_route.py [workload] [n_sd] [steps]"""
import sys
import time

import torch

from pysdm_amd.cases import make_box
from pysdm_amd.engine import HipEngine

name = sys.argv[1] if len(sys.argv) > 1 else "shima"
n_sd = int(sys.argv[2]) if len(sys.argv) > 2 else 2**20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for route in ("chain", "fused"):
    runner = make_box(HipEngine.get(), name, n_sd=n_sd, route=route)
    runner.run(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.run(steps)
    torch.cuda.synchronize()
    print(name, n_sd, route, f"{(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
