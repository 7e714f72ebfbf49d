"""kernel sequence (name, start offset, duration in us) of the last N dispatches of a
rocprofv3 --kernel-trace run: python step_sequence.py <dir> [N]"""
import csv
import glob
import sys

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(path, encoding="utf-8")), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 30):]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(a - t0) / 1e3:9.1f} us  +{(b - a) / 1e3:6.1f}  gap {(a - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:60]}")
    prev_end = b
