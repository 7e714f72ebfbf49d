"""reproducer: a sorted multi-cell state handed over with another state's cell_start
(usage: python tests/helpers/stale_cell_start.py; must end in RuntimeError, code 4)"""
import faulthandler
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.dump_traceback_later(15, exit=False)
if os.environ.get("SDM_ALARM"):  # under a debugger: `handle SIGALRM stop nopass`, then `bt`
    import signal
    signal.alarm(int(os.environ["SDM_ALARM"]))

from pysdm_amd.cases import make_box  # noqa: E402
from pysdm_amd.engine import HipEngine  # noqa: E402

e = HipEngine.get()
runner = make_box(e, "kinematic2d", n_sd=2**14, grid=(4, 4))
runner.run(2)
pop = runner.population
starts = e.download(pop.cell_start)
starts[5:] -= 1
pop.cell_start.copy_(e.upload(starts))
pop.touch_state()
pop.ordered = True
print("running on the stale cell_start", flush=True)
try:
    runner.run(3)
    print("no error", flush=True)
except RuntimeError as err:
    print("refused:", err, flush=True)
