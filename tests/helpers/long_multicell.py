"""progress of a long 32 x 32 run (deaths set in after ~50 steps): python tests/helpers/long_multicell.py [chunk] [n_chunks]"""
import sys, time
sys.path.insert(0, ".")
from pysdm_amd.cases import make_box
from pysdm_amd.engine import HipEngine
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
e = HipEngine.get()
r = make_box(e, "kinematic2d")
for k in range(n):
    t0 = time.time()
    r.run(chunk)
    e.synchronize()
    print("steps", r.steps_done, "sub-steps", r.sub_steps_done, "live", r.population.live,
          "s", round(time.time() - t0, 3), flush=True)
