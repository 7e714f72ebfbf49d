"""launch sequence of one stretch of a rocprofv3 kernel trace kept as a rocpd database, copies and
fills included, and their counts per time step:
    python profiles/tools/trace_sequence.py DB MARKER_KERNEL [n_steps_to_print]
MARKER_KERNEL: a kernel launched exactly once per time step (e.g. k_shard_begin)"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select k.start, k.end, s.kernel_name from rocpd_kernel_dispatch k "
    "join rocpd_info_kernel_symbol s on k.kernel_id = s.id order by k.start").fetchall()
marker = sys.argv[2]
n_print = int(sys.argv[3]) if len(sys.argv) > 3 else 1


def short(name):
    name = re.sub(r"\(.*", "", name)
    return re.sub(r"^_Z\d+", "", name)[:48]


marks = [i for i, r in enumerate(rows) if marker in r[2]]
print(f"{len(rows)} launches, {len(marks)} x {marker}")
per_step = collections.Counter()
steps = 0
# (the timed repetition's steps: after the first three - scratch allocation, warm-up - and before
# the per-kernel timing pass and the read-out at the end of the bench)
first, last = 4, max(5, len(marks) - 4)
for a, b in zip(marks[first:last], marks[first + 1:last + 1]):
    steps += 1
    for r in rows[a:b]:
        per_step[short(r[2])] += 1
print(f"per time step (mean over {steps} steps):")
for name, count in per_step.most_common():
    print(f"  {name:50s} {count / steps:7.2f}")
copies = sum(v for k, v in per_step.items() if "copyBuffer" in k or "fillBuffer" in k) / max(steps, 1)
print(f"copy / fill launches per step: {copies:.1f}")
a, b = marks[first + 2], marks[first + 2 + n_print]
t0 = rows[a][0]
for start, end, name in rows[a:b]:
    print(f"{(start - t0) / 1e3:9.1f} {(end - start) / 1e3:7.1f} us  {short(name)}")
