import numpy as np, time, torch
from pysdm_amd.backends import HIP
from pysdm_amd.examples import make_box
p, d = make_box(HIP, "berry_breakup")
for target in (500, 800, 1000):
    t=time.time(); p.run(target - p.n_steps); p.backend.synchronize() if hasattr(p.backend,"synchronize") else None
    n = p.attributes["multiplicity"].to_ndarray(); n = n[n>0]
    print(target, "steps; wall", round(time.time()-t,2), "n_sd", len(n), "mult min/median/max", n.min(), np.median(n), n.max(), "ratio", n.max()/n.min(), "substeps", d.stats_n_substep.to_ndarray())
