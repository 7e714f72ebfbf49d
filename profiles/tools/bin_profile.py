"""phase times inside k_bin_sort / k_bin_build2 (workgroup 7) from a -DBIN_PROFILE build:
SDM_HIP_LIB=build_variants/libsdm_binprof.so PYTHONPATH=. python profiles/tools/bin_profile.py"""
import ctypes

import numpy as np

from pysdm_amd import abi
from pysdm_amd.cases import make_box
from pysdm_amd.engine import HipEngine

engine = HipEngine.get()
runner = make_box(engine, "shima", adaptive=False, read_back=False)
runner.run(20)
engine.synchronize()
out = (ctypes.c_longlong * 32)()
rows = []
for _ in range(10):
    runner.run(1)
    engine.synchronize()
    assert abi.hip_library().cdll.sdm_debug_bin_profile(out) == 0
    rows.append(np.array([out[k] for k in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 12)], dtype=np.int64))
ticks = np.median(np.diff(np.array(rows), axis=1), axis=0) * 0.01  # 100 MHz -> us
names = ["sort: table to LDS", "sort: tile jump", "sort: targets (thread jump + 4 draws)",
         "sort: histogram", "sort: scan", "sort: placement", "sort: write back",
         "(between the kernels)", "build: run lengths + scan", "build: slot init",
         "build: events -> slots", "build: S words by place (successor words only)",
         "build: first / tsucc (records) out"]
for name, value in zip(names, ticks):
    print(f"{name:40s} {value:7.2f} us")
