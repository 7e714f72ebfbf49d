"""prints the top kernels of a rocprofv3 --kernel-trace --stats run (helper for profiles/)"""
import csv
import glob
import sys

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for row in list(csv.DictReader(open(path, encoding="utf-8")))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{row['Name'][:64]:64s} calls={row['Calls']:>6s} avg_us={float(row['AverageNs']) / 1e3:>8.1f} "
          f"pct={row['Percentage']}")
