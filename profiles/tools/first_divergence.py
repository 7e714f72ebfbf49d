"""first time step at which the HIP fused route and the oracle differ on a breakup box, and how:
PYTHONPATH=. python profiles/tools/first_divergence.py [workload] [n_sd] [steps]"""
import sys
import warnings

import numpy as np

from oracle.engine import OracleEngine
from pysdm_amd.cases import make_box
from pysdm_amd.engine import HipEngine

name = sys.argv[1] if len(sys.argv) > 1 else "straub_rain"
n_sd = int(sys.argv[2]) if len(sys.argv) > 2 else 2**18
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 150
runners = [make_box(e, name, n_sd=n_sd, adaptive=True) for e in (HipEngine.get(), OracleEngine.get())]
for step in range(1, steps + 1):
    snaps = []
    for runner in runners:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            runner.run(1)
        snaps.append(runner.snapshot())
    a, b = snaps
    n_a, n_b = a["multiplicity"], b["multiplicity"]
    m_a, m_b = a["attributes"][0], b["attributes"][0]
    if not np.array_equal(n_a, n_b) or not np.array_equal(a["idx"][: int(a["length"])],
                                                          b["idx"][: int(b["length"])]):
        bad = np.flatnonzero(n_a != n_b)
        print(f"step {step}: {len(bad)} multiplicities differ; first: id {bad[:5]} hip {n_a[bad[:5]]} "
              f"oracle {n_b[bad[:5]]}; masses there hip {m_a[bad[:5]]} oracle {m_b[bad[:5]]}")
        rel = np.abs(m_a - m_b) / np.maximum(np.abs(m_b), 1e-300)
        print("max relative mass difference over all droplets:", rel.max(),
              "sub-steps", a["stats_n_substep"], b["stats_n_substep"])
        break
    rel = np.abs(m_a - m_b) / np.maximum(np.abs(m_b), 1e-300)
    if step % 10 == 0:
        print(f"step {step}: identical integers, max relative mass difference {rel.max():.3e}", flush=True)
else:
    print("no divergence in", steps, "steps")
