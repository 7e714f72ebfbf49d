"""profiles/r04_pmc_<workload>_<COUNTER>.txt (tests/pmc_summary.py lines) -> the "r04" entries of
profiles/traffic.json that bench.py's roofline object reads:
    python profiles/tools/traffic_r04.py WORKLOAD PHASE KERNEL_SUBSTRING COALESCED_READ_KB
PHASE: the bench's name of the kernel's phase ("pair_update", "pair_prob").  COALESCED_READ_KB:
what the kernel reads in whole 128-B requests, which gfx950 books at half in FETCH_SIZE
(MI355X_MICROARCH.md; calibrated on k_nm_init) - added once more; random 64-B sector misses are
booked at face value (profiles/tools/calib_gather.cpp).  Also writes "r04_calibration": the L2
requests and misses per read of k_calib_random, the in-run calibration kernel of bench.py."""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean(workload, counter, kernel):
    path = os.path.join(HERE, f"r04_pmc_{workload}_{counter}.txt")
    for line in open(path, encoding="utf-8"):
        if kernel in line:
            return (float(re.search(r"mean_\w+=\s*([\d.]+)", line).group(1)),
                    int(re.search(r"launches=\s*(\d+)", line).group(1)))
    raise KeyError(f"{kernel} not in {path}")


workload, phase, kernel, coalesced_kb = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
fetch_kb, launches = mean(workload, "FETCH_SIZE", kernel)
write_kb, _ = mean(workload, "WRITE_SIZE", kernel)
entry = {"kernel": kernel, "launches_profiled": launches, "fetch_kb": fetch_kb,
         "write_kb": write_kb, "coalesced_read_kb_booked_at_half": coalesced_kb,
         "bytes": (fetch_kb + coalesced_kb + write_kb) * 1024,
         "tcc_req": mean(workload, "TCC_REQ_sum", kernel)[0],
         "tcc_hit": mean(workload, "TCC_HIT_sum", kernel)[0],
         "tcc_miss": mean(workload, "TCC_MISS_sum", kernel)[0]}
path = os.path.join(HERE, "traffic.json")
table = json.load(open(path, encoding="utf-8"))
table.setdefault("r04", {}).setdefault(workload, {})[phase] = entry
if workload == "shima":
    reads = 4 * 2**20  # bench.py: random_sector_ceiling issues 4 n_sd reads per launch
    table["r04_calibration"] = {
        "kernel": "k_calib_random (4 n_sd independent random 16-B reads over the pair kernel's "
                  "table footprint; the bench times it live, the counters are from the same "
                  "offline passes)",
        "reads_per_launch": reads,
        "tcc_req_per_read": mean(workload, "TCC_REQ_sum", "k_calib_random")[0] / reads,
        "tcc_hit_per_read": mean(workload, "TCC_HIT_sum", "k_calib_random")[0] / reads,
        "tcc_miss_per_read": mean(workload, "TCC_MISS_sum", "k_calib_random")[0] / reads,
        "fetch_bytes_per_read": mean(workload, "FETCH_SIZE", "k_calib_random")[0] * 1024 / reads}
json.dump(table, open(path, "w", encoding="utf-8"), indent=1)
print(json.dumps(entry, indent=1))
