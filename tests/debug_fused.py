import sys, numpy as np
sys.path.insert(0, '.')
from tests.trajectory import setup_from_golden, snapshot
from pysdm_amd.backends import HIP
name = sys.argv[1]; nsteps = int(sys.argv[2])
pa, da, _, _ = setup_from_golden(name, HIP, fused=False)
pb, db, _, _ = setup_from_golden(name, HIP, fused=None)
for step in range(1, nsteps + 1):
    pa.run(1); pb.run(1)
    sa, sb = snapshot(pa, da), snapshot(pb, db)
    bad = [k for k in sa if not np.array_equal(sa[k], sb[k], equal_nan=True)]
    print(step, 'len', int(sa['length']), int(sb['length']), 'diff:', bad)
    if bad:
        for k in bad:
            w = np.flatnonzero(np.asarray(sa[k]).ravel() != np.asarray(sb[k]).ravel())
            print(' ', k, len(w), w[:10], np.asarray(sa[k]).ravel()[w[:10]], np.asarray(sb[k]).ravel()[w[:10]])
        break
