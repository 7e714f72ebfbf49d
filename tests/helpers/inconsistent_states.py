"""which inconsistent hand-overs end an adaptive multi-cell step in SDM_E_STATE (the sub-step bound)
rather than in a result: tries several cell_starts that have the right totals but do not belong to
the permutation (usage: python tests/helpers/inconsistent_states.py; every case must return)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pysdm_amd.cases import make_box  # noqa: E402
from pysdm_amd.engine import HipEngine  # noqa: E402

e = HipEngine.get()


def corrupt(name, edit, **box):
    runner = make_box(e, "kinematic2d", n_sd=2**14, grid=(4, 4), **box)
    runner.run(2)
    pop = runner.population
    starts = e.download(pop.cell_start).copy()
    edit(starts)
    assert starts[0] == 0 and starts[-1] == pop.live and (np.diff(starts) >= 0).all()
    pop.cell_start.copy_(e.upload(starts))
    pop.touch_state()
    pop.ordered = True
    try:
        runner.run(1)
        print(name, "-> returned, sub-steps", runner.sub_steps_done, flush=True)
    except RuntimeError as err:
        print(name, "-> error:", str(err)[:160], flush=True)


def merge(s):
    s[5] = s[4]


def shift(s):
    s[1:-1] += 300


def shift_back(s):
    s[1:-1] -= 300


def squeeze(s):
    s[1:-1] = s[-1]


def front(s):
    s[1:-1] = 0


for label, edit in (("merge 4 into 5", merge), ("all boundaries + 300", shift),
                    ("all boundaries - 300", shift_back), ("all in cell 0", squeeze),
                    ("all in the last cell", front)):
    corrupt(label, edit)
    corrupt(label + " (dt_range 0.01..1)", edit, dt_range=(0.01, 1.0))
