"""Full per-kernel PMC table of one headline step (every conv / pair instantiation, no row cut) from the separate
rocprofv3 --pmc passes of scripts/gpu_round6.sh over `bench.py --serial-branches --lanes 1 --steps 2`:
    python scripts/pmc_table_full.py <pass1.csv> <pass2.csv> ...
Columns: launches, HBM traffic (FETCH_SIZE KiB x 1024 x 2 — the gfx950 correction of MI355X_MICROARCH.md — + WRITE_SIZE KiB x 1024),
kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs), matrix-pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles), instruction mix per MFMA
(SQ_INSTS_VALU excluding MFMA, SQ_INSTS_LDS, SQ_INSTS_VMEM_RD), the issue-slot model 32 / (36 + 4 (valu + lds + vmem)) — an MFMA holds the matrix pipe 32 cycles and its own issue takes a
quad-cycle like every other instruction's — next to the measured busy fraction, LDS bank-conflict cycles / LDS-array cycles, share of wave cycles parked (SQ_WAIT_ANY) / issue-stalled
(SQ_WAIT_INST_ANY) / issuing (SQ_ACTIVE_INST_ANY)."""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"])).replace(" ", "").replace("ttsamd::", "")
        agg[name][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
rows = []
for k, v in agg.items():
    a = {c: sum(d.values()) / len(d) for c, d in v.items()}
    n = max(len(d) for d in v.values())
    cyc = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    rows.append((cyc * n, k, n, a, cyc))
rows.sort(reverse=True)
print("%-52s %5s %9s %9s %10s %6s %7s %6s %6s %6s %7s %6s %6s %6s" % ("kernel", "n", "fetch MB", "write MB", "cycles", "busy", "valu/mf", "lds/mf", "vm/mf", "model", "bankcf", "park", "stall", "issue"))
for _, k, n, a, cyc in rows:
    mf = a.get("SQ_INSTS_MFMA", 0.0)
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    f = lambda x, fmt="%.3f": (fmt % x) if x is not None else "-"  # noqa: E731
    valu = (a["SQ_INSTS_VALU"] - mf) / mf if (mf and "SQ_INSTS_VALU" in a) else None
    lds = a["SQ_INSTS_LDS"] / mf if (mf and "SQ_INSTS_LDS" in a) else None
    vm = a["SQ_INSTS_VMEM_RD"] / mf if (mf and "SQ_INSTS_VMEM_RD" in a) else None
    model = 32.0 / (36.0 + 4.0 * (valu + lds + vm)) if None not in (valu, lds, vm) else None
    print("%-52s %5d %9s %9s %10s %6s %7s %6s %6s %6s %7s %6s %6s %6s" % (
        k[:52], n, f(a["FETCH_SIZE"] * 2048 / 1e6, "%.1f") if "FETCH_SIZE" in a else "-", f(a["WRITE_SIZE"] * 1024 / 1e6, "%.1f") if "WRITE_SIZE" in a else "-",
        f(cyc, "%.4g") if cyc else "-", f(a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc) if (cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in a and mf) else "-",
        f(valu, "%.2f"), f(lds, "%.2f"), f(vm, "%.2f"), f(model),
        f(a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"]) if a.get("SQ_LDS_IDX_ACTIVE") else "-",
        f(a["SQ_WAIT_ANY"] / wc, "%.2f") if (wc and "SQ_WAIT_ANY" in a) else "-", f(a["SQ_WAIT_INST_ANY"] / wc, "%.2f") if (wc and "SQ_WAIT_INST_ANY" in a) else "-",
        f(a["SQ_ACTIVE_INST_ANY"] / wc, "%.2f") if (wc and "SQ_ACTIVE_INST_ANY" in a) else "-"))
