"""randomised differential run (not collected by pytest; by hand on an MI355X:
    python tests/fuzz_parity.py [n_cases] [seed] [first] [last]):
the cases of tests/fuzz_cases.py:draw_parity_case - HIP (fused or stage-by-stage route) against
the checker, everything to the bit.  A fixed-seed slice runs under `-m gpu`
(tests/test_hip_fuzz.py)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.engine import OracleEngine  # noqa: E402
from pysdm_amd.engine import HipEngine  # noqa: E402
from tests import fuzz_cases  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
# optional: execute only cases [first, last) of the sequence (the others are drawn and skipped) -
# to tell a failure that depends on what ran before in the process from one that does not
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
last = int(sys.argv[4]) if len(sys.argv) > 4 else n_cases
hip, oracle = HipEngine.get(), OracleEngine.get()
t_start = time.time()
for number in range(n_cases):
    case = fuzz_cases.draw_parity_case(rng)
    if not first <= number < last:
        continue
    print(f"case {number}: {case} ->", fuzz_cases.run_parity_case(hip, oracle, case), flush=True)
print("all cases equal the checker;", round(time.time() - t_start, 1), "s")
