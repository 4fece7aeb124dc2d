"""Per-kernel PMC summary of one bench step from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ/GRBM set):
    pmc_round.py <out.json> <kernel substring> <fetch.csv> <write.csv> [<sq.csv>]
Writes {hbm_bytes_per_launch, fetch/write, mfma_busy_frac, effective_clock_ghz, ...} for the kernels whose name contains
the substring (spaces ignored) and prints a table of all ttsamd kernels.
Corrections (profiles/r01_calibration_copy.txt, /opt/skills/guides/MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE are KiB;
FETCH_SIZE reports half of the bytes of coalesced streaming reads on gfx950 (x2); WRITE_SIZE is exact.
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over all SIMDs (256 CUs x 4); GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        name = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"])).replace(" ", "")
        agg[name][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {c: (sum(d.values()) / len(d), len(d)) for c, d in v.items()} for k, v in agg.items()}


out, sub = sys.argv[1], sys.argv[2].replace(" ", "")
tabs = [load(p) for p in sys.argv[3:]]
merged = collections.defaultdict(dict)
for t in tabs:
    for k, v in t.items():
        merged[k].update(v)
rows = []
for k, v in merged.items():
    if "ttsamd" not in k:
        continue
    n = max(c[1] for c in v.values())
    g = lambda c: v[c][0] if c in v else None  # noqa: E731
    row = {"kernel": k.replace("ttsamd::", ""), "launches": n}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        row["fetch_bytes_per_launch"] = g("FETCH_SIZE") * 1024 * 2
        row["write_bytes_per_launch"] = g("WRITE_SIZE") * 1024
        row["hbm_bytes_per_launch"] = row["fetch_bytes_per_launch"] + row["write_bytes_per_launch"]
    if g("GRBM_GUI_ACTIVE") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        cyc = g("GRBM_GUI_ACTIVE") / 8.0
        row["kernel_cycles"] = cyc
        row["mfma_busy_frac"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / cyc
        for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY",
                  "SQ_LDS_BANK_CONFLICT"):
            if g(c) is not None:
                row[c] = g(c)
    rows.append(row)
rows.sort(key=lambda r: -(r.get("kernel_cycles", 0) * r["launches"] or r.get("hbm_bytes_per_launch", 0) * r["launches"]))
print("%-58s %5s %12s %12s %9s %12s" % ("kernel", "n", "fetch MB", "write MB", "mfma busy", "cycles"))
for r in rows[:28]:
    print("%-58s %5d %12s %12s %9s %12s" % (r["kernel"][:58], r["launches"],
          "%.1f" % (r["fetch_bytes_per_launch"] / 1e6) if "fetch_bytes_per_launch" in r else "-",
          "%.1f" % (r["write_bytes_per_launch"] / 1e6) if "write_bytes_per_launch" in r else "-",
          "%.3f" % r["mfma_busy_frac"] if "mfma_busy_frac" in r else "-",
          "%.4g" % r["kernel_cycles"] if "kernel_cycles" in r else "-"))
sel = [r for r in rows if sub in r["kernel"].replace(" ", "")]
if sel:
    n = sum(r["launches"] for r in sel)
    avg = lambda key: sum(r[key] * r["launches"] for r in sel if key in r) / max(sum(r["launches"] for r in sel if key in r), 1)  # noqa: E731
    res = {"kernel": sub, "launches": n, "source": "rocprofv3 --pmc passes (one counter set per pass) over bench.py "
           "--serial-branches --lanes 1; FETCH_SIZE KiB x1024 x2, WRITE_SIZE KiB x1024; mfma_busy_frac = "
           "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)"}
    for key in ("fetch_bytes_per_launch", "write_bytes_per_launch", "hbm_bytes_per_launch", "mfma_busy_frac", "kernel_cycles"):
        if any(key in r for r in sel):
            res[key] = avg(key)
    import bench          # code_stamp: sha256 of the kernel sources these counters were recorded with (bench.py checks it)

    res["code_stamp"] = bench.code_stamp()
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))
