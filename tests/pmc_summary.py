"""per-kernel mean of one PMC counter from a `rocprofv3 --kernel-trace --pmc X` run (helper for
profiles/traffic.json): python tests/pmc_summary.py <dir> <COUNTER>"""
import collections
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
total, calls = collections.Counter(), collections.Counter()
with open(path, encoding="utf-8") as handle:
    for row in csv.DictReader(handle):
        if row["Counter_Name"] != sys.argv[2]:
            continue
        name = row["Kernel_Name"].split("(")[0]
        total[name] += float(row["Counter_Value"])
        calls[name] += 1
for name, value in total.most_common(12):
    print(f"{name[:60]:60s} launches={calls[name]:5d} mean_{sys.argv[2]}={value / calls[name]:14.1f}")
