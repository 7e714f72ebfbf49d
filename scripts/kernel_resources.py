#!/usr/bin/env python3
"""per-kernel VGPR / SGPR / scratch / LDS / occupancy of one translation unit (device pass only):
    scripts/kernel_resources.py pysdm_amd/csrc/fused.hip [extra hipcc flags]"""
import re
import subprocess
import sys

FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "--offload-device-only", "-c", "-o", "/dev/null",
         "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *sys.argv[1:]], capture_output=True,
                     text=True, check=False).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (.*)", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    g = lambda k: r.get(k, "?")  # noqa: E731
    print(f"{r['name'][:90]:90s} vgpr {g('VGPRs'):>4s} agpr {g('AGPRs'):>3s} sgpr {g('SGPRs'):>4s} "
          f"scratch {g('ScratchSize [bytes/lane]'):>5s} occ {g('Occupancy [waves/SIMD]'):>2s} "
          f"lds {g('LDS Size [bytes/block]'):>6s}")
