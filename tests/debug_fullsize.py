import sys, numpy as np, warnings
sys.path.insert(0, '.')
from tests.test_hip_full_size import CONFIGS, box, run
from tests.trajectory import snapshot
from pysdm_amd.backends import HIP
from oracle.backend import OracleBackend
name = sys.argv[1]
cfg = dict(CONFIGS[name]); make = cfg.pop("make")
if len(sys.argv) > 2:
    cfg["n_sd"] = int(sys.argv[2])
res = {}
for label, bc, fused in (("oracle", OracleBackend, False), ("methods", HIP, False), ("fused", HIP, None)):
    p, d = box(bc, dynamic=make(True, fused), **cfg)
    s0 = snapshot(p, d)
    run(p, 1)
    res[label] = (s0, snapshot(p, d))
for label in ("methods", "fused"):
    for when in (0, 1):
        a, b = res[label][when], res["oracle"][when]
        L = int(a["length"])
        bad = [k for k in a if not np.array_equal(a[k][:L] if k == "idx" else a[k], b[k][:L] if k == "idx" else b[k], equal_nan=True)]
        print(label, "after" if when else "before", "diff:", bad, "nsub", a["stats_n_substep"][:4], b["stats_n_substep"][:4])
