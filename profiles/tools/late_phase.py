"""average kernel duration per window of dispatches from a rocprofv3 --kernel-trace run:
python late_phase.py <dir> [n_windows] - shows how per-kernel time drifts over a long run"""
import collections
import csv
import glob
import sys

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
n_win = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = list(csv.DictReader(open(path, encoding="utf-8")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
acc = collections.defaultdict(lambda: [[0, 0] for _ in range(n_win)])
for r in rows:
    w = min(n_win - 1, (int(r["Start_Timestamp"]) - t0) * n_win // (t1 - t0 + 1))
    cell = acc[r["Kernel_Name"][:40]][w]
    cell[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cell[1] += 1
for name, wins in sorted(acc.items(), key=lambda kv: -sum(c[0] for c in kv[1]))[:10]:
    print(f"{name:40s}", " ".join(f"{c[0] / max(c[1], 1) / 1e3:7.1f}us x{c[1]:<5d}" for c in wins))
