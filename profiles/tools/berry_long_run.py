"""the Berry breakup box (BASELINE.json configs[2]) far into its run, where single pairs ask for
1e4..1e6 successive breakups (`break_up`, collisions_methods.py:62-132): candidate pairs/s and
per-phase times in windows of 250 steps up to step 1500.
    PYTHONPATH=. python profiles/tools/berry_long_run.py > profiles/r03_berry_breakup_1500steps.json"""
import ctypes
import json
import time

from pysdm_amd.cases import make_box
from pysdm_amd.engine import HipEngine

import sys

TIMING = "--phases" in sys.argv  # per-phase HIP events (slower: no launch-ahead) or plain wall clock
engine = HipEngine.get()
runner = make_box(engine, "berry_breakup")
runner.run(1)
name = engine.library.cdll.sdm_phase_name
name.restype = ctypes.c_char_p
windows = []
total_pairs, total_time = 0, 0.0
for target in range(250, 1501, 250):
    engine.call("sdm_ctx_set_timing", 1 if TIMING else 0)
    engine.synchronize()
    pairs0, sub0, t0 = runner.pairs_done, runner.sub_steps_done, time.perf_counter()
    steps = target - runner.steps_done
    runner.run(steps)
    engine.synchronize()
    elapsed = time.perf_counter() - t0
    ms = (ctypes.c_double * 12)()
    count = (ctypes.c_int64 * 12)()
    engine.call("sdm_ctx_read_timing", ms, count)
    engine.call("sdm_ctx_set_timing", 0)
    phases = {name(i).decode(): round(ms[i] / steps, 5) for i in range(12) if count[i] > 0}
    pairs = runner.pairs_done - pairs0
    total_pairs += pairs
    total_time += elapsed
    windows.append({"steps": f"{target - steps}..{target}", "pairs_per_s": pairs / elapsed,
                    "ms_per_step": elapsed / steps * 1e3,
                    "substeps_per_step": (runner.sub_steps_done - sub0) / steps,
                    "phase_ms_per_step (timing mode: no launch-ahead)": phases})
print(json.dumps({"workload": "berry_breakup, n_sd=2^20, adaptive, steps 1..1500"
                              + (" (timing mode on)" if TIMING else ""),
                  "library": __import__("os").environ.get("SDM_HIP_LIB", "pysdm_amd/libsdm_hip.so"),
                  "sustained_pairs_per_s": total_pairs / total_time, "windows": windows}, indent=1))
