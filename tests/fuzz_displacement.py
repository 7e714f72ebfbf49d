"""randomised differential run of the displacement step (+ collisions behind it), by hand on an
MI355X like tests/fuzz_parity.py:  python tests/fuzz_displacement.py [n_cases] [seed]
the cases of tests/fuzz_cases.py:draw_displacement_case - HIP against the checker: cell origins,
cell ids, permutation, multiplicities exact; positions and masses to 1e-12.  A fixed-seed slice
runs under `-m gpu` (tests/test_hip_fuzz.py)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.engine import OracleEngine  # noqa: E402
from pysdm_amd.engine import HipEngine  # noqa: E402
from tests import fuzz_cases  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
hip, oracle = HipEngine.get(), OracleEngine.get()
t_start = time.time()
for number in range(n_cases):
    case = fuzz_cases.draw_displacement_case(rng)
    left = fuzz_cases.run_displacement_case(hip, oracle, case)
    print(f"case {number}: grid={case['grid']} n_sd={case['n_sd']} route={case['route']} "
          f"collide={case['collide']} steps={case['steps']} -> ok, left {left}", flush=True)
print("all cases equal the checker;", round(time.time() - t_start, 1), "s")
