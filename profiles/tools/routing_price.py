"""pricing a re-routing of the shuffle's walks (XCD-local first hops): the rate
of random 16-byte READS for tables that do / do not fit an XCD's L2, against the rate of scattered
8-byte WRITES (what delivering 2^20 walk results to their pair slots would cost).
    PYTHONPATH=. python profiles/tools/routing_price.py > profiles/r03_xcd_routing_pricing.json"""
import ctypes
import json

from pysdm_amd.engine import HipEngine

engine = HipEngine.get()
n = 2**20
out = {"reads_per_s": {}, "writes_per_s": {}}
for label, records in (("2 MiB table (one XCD's share of the 16-MiB records: L2-resident)", 2**17),
                       ("16 MiB table (the records)", 2**20), ("32 MiB (records + mirror)", 2**21)):
    ms, checksum = ctypes.c_double(), ctypes.c_uint64()
    engine.call("sdm_calib_random_sectors", records, 4 * n, 10, ms, checksum)
    out["reads_per_s"][label] = 4 * n / (ms.value * 1e-3)
for label, words in (("8 MiB of int64 (one result per position)", 2**20),):
    ms = ctypes.c_double()
    engine.call("sdm_calib_random_writes", words, n, 10, ms)
    out["writes_per_s"][label] = n / (ms.value * 1e-3)
    out["us_for_2^20_scattered_writes"] = ms.value * 1e3
r_far = out["reads_per_s"]["16 MiB table (the records)"]
r_near = out["reads_per_s"]["2 MiB table (one XCD's share of the 16-MiB records: L2-resident)"]
out["us_saved_on_2^20_first_hops"] = (n / r_far - n / r_near) * 1e6
print(json.dumps(out, indent=1))
