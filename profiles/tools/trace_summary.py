"""per-kernel statistics and a stretch of the timeline out of a rocprofv3 run kept as a rocpd
database (`rocprofv3 --kernel-trace -d DIR -o NAME -- ...` writes DIR/NAME_results.db):
    python profiles/tools/trace_summary.py DB [first-kernel-substring] [n_timeline]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select k.start, k.end, s.kernel_name, k.grid_size_x, k.workgroup_size_x "
    "from rocpd_kernel_dispatch k join rocpd_info_kernel_symbol s on k.kernel_id = s.id "
    "order by k.start").fetchall()
start_at = sys.argv[2] if len(sys.argv) > 2 else None
n_line = int(sys.argv[3]) if len(sys.argv) > 3 else 24


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^_Z\d+", "", name)
    return name[:56]


if start_at:
    first = [i for i, r in enumerate(rows) if start_at in r[2]]
    rows = rows[first[0]:] if first else rows
stats = collections.defaultdict(lambda: [0, 0.0])
for a, b, name, _, _ in rows:
    stats[short(name)][0] += 1
    stats[short(name)][1] += (b - a) / 1e3
busy = sum(v[1] for v in stats.values())
print(f"{len(rows)} launches, kernels busy {busy:.0f} us")
for name, (count, total) in sorted(stats.items(), key=lambda x: -x[1][1])[:22]:
    print(f"{name:58s} {count:6d} {total:10.1f} us {total / count:8.2f} us each")
middle = len(rows) // 2
t0 = rows[middle][0]
previous_end = t0
for a, b, name, grid, wg in rows[middle:middle + n_line]:
    print(f"{(a - t0) / 1e3:9.1f} +{(a - previous_end) / 1e3:5.1f} gap {(b - a) / 1e3:8.1f} us  "
          f"{short(name):44s} {grid // max(wg, 1)} x {wg}")
    previous_end = b
