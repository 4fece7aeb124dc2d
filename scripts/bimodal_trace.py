"""Per-sentence kernel time vs gap time of a configs[0] kernel trace: python scripts/bimodal_trace.py <kernel_trace.csv> [tag]
A sentence starts at embed_kernel.  Over the last 150 sentences: median period, median sum of kernel durations, median idle time,
and the 14 kernel names with the largest total time (median duration each) — to diff a fast process against a slow one."""
import csv
import re
import statistics as S
import sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
tag = sys.argv[2] if len(sys.argv) > 2 else "-"
starts = [i for i, r in enumerate(rows) if "embed_kernel" in r["Kernel_Name"]]
starts = starts[-151:]


def nm(r):
    return re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"])).replace("ttsamd::", "")[:60]


per, ksum, idle, nk = [], [], [], []
byname = defaultdict(list)
for a, b in zip(starts[:-1], starts[1:]):
    req = rows[a:b]
    t0, t1 = int(req[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    busy_end, gap = t0, 0
    for r in req:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > busy_end:
            gap += s - busy_end
        busy_end = max(busy_end, e)
        byname[nm(r)].append((e - s) / 1e3)
    gap += max(0, t1 - busy_end)
    per.append((t1 - t0) / 1e3)
    ksum.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in req) / 1e3)
    idle.append(gap / 1e3)
    nk.append(len(req))
print("TRACE %-18s sentences %d kernels/sentence %d | period p50 %.1f us | kernel-time sum p50 %.1f us | idle p50 %.1f us"
      % (tag, len(per), S.median(nk), S.median(per), S.median(ksum), S.median(idle)))
top = sorted(byname.items(), key=lambda kv: -sum(kv[1]))[:14]
for k, v in top:
    print("   %-60s n/sent %5.1f  median %7.2f us  total/sent %7.1f us" % (k, len(v) / len(per), S.median(v), sum(v) / len(per)))
