"""HBM traffic per launch of the dominant conv kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter
per pass, no trace domains) of `bench.py --serial-branches`: usage
    pmc_dominant.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> <out.json>
Corrections as calibrated in profiles/r01_calibration_copy.txt (and /opt/skills/guides/MI355X_MICROARCH.md): both counters are
in KiB; FETCH_SIZE reports half of the bytes of coalesced streaming reads on gfx950 -> x2; WRITE_SIZE is exact."""
import csv
import json
import sys


def per_launch(path, counter, sub):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and sub in r["Kernel_Name"].replace(" ", ""):
            vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return sum(vals.values()) / max(len(vals), 1), len(vals)


fetch_kb, n = per_launch(sys.argv[1], "FETCH_SIZE", sys.argv[3])
write_kb, n2 = per_launch(sys.argv[2], "WRITE_SIZE", sys.argv[3])
out = {"kernel": sys.argv[3], "launches": n, "fetch_bytes_per_launch": fetch_kb * 1024 * 2, "write_bytes_per_launch": write_kb * 1024,
       "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --serial-branches; FETCH_SIZE KiB x1024 x2, "
                 "WRITE_SIZE KiB x1024 (profiles/r01_calibration_copy.txt)"}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
