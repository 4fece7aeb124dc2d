"""Condense rocprofv3 CSV output (kernel stats / PMC counter collection) into the small text summaries that are
committed under profiles/.   usage: prof_summary.py stats <trace_kernel_stats.csv> | pmc <pmc_counter_collection.csv>"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"void ttsamd::conv1d_(mfma|x3)_kernel<(.*?)>.*", r"conv1d_\1_kernel<\2>", name)
    name = re.sub(r"void ttsamd::resblock_pair_x3_kernel<(.*?)>.*", r"resblock_pair_x3_kernel<\1>", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)[:70]


def stats(path, top=45):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("# rocprofv3 --kernel-trace --stats : %d kernels, %.3f ms total GPU time" % (len(rows), tot / 1e6))
    print("# NOTE avg_us is the MEAN over every launch of the process, including each kernel's first launches (code-object upload:"
          " one launch of tens of ms for the 137 KB-LDS fused kernel, 2x slower launches during the first two steps); the "
          "steady-state per-launch figure is the MEDIAN in the per-shape table next to this file (trace_shapes.py)")
    print("%-72s %6s %11s %11s %11s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "pct"))
    for r in rows[:top]:
        print("%-72s %6s %11.3f %11.1f %11.1f %7.2f" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                      float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                                                      float(r["Percentage"])))


def pmc(path, top=30):
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    names = sorted({c for k in agg for c in agg[k]})
    print("# rocprofv3 --pmc %s : per-kernel counter sums (dispatch count in brackets)" % " ".join(names))
    order = sorted(agg, key=lambda k: -max(agg[k].values()))
    for k in order[:top]:
        print("%-72s " % k + "  ".join("%s=%.6g[%d]" % (c, agg[k][c], cnt[k][c]) for c in names))


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
