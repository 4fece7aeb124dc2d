"""Where a request's wall time goes that is NOT kernel time: python scripts/req_gaps.py <kernel_trace.csv> [marker]
For every request of the trace (a request starts at a kernel whose name contains `marker`, default embed_kernel): period to the
next request's start, time with no kernel running inside the period, and the largest gaps with the kernels either side."""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "embed_kernel"
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]


def nm(r):
    return re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"])).replace("ttsamd::", "")[:44]


for a, b in zip(starts[:-1], starts[1:]):
    req = rows[a:b]
    t0, t1 = int(req[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    busy_end, gaps = t0, []
    for k, r in enumerate(req):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > busy_end:
            gaps.append(((s - busy_end) / 1e3, nm(req[k - 1]) if k else "-", nm(r), (s - t0) / 1e3))
        busy_end = max(busy_end, e)
    if t1 > busy_end:
        gaps.append(((t1 - busy_end) / 1e3, nm(req[-1]), "NEXT REQUEST", (busy_end - t0) / 1e3))
    idle = sum(g[0] for g in gaps)
    print("request of %d kernels: period %.1f us, idle %.1f us (%d gaps)" % (len(req), (t1 - t0) / 1e3, idle, len(gaps)))
    for g in sorted(gaps, reverse=True)[:6]:
        print("      gap %6.1f us at +%7.1f  after %-44s before %s" % (g[0], g[3], g[1], g[2]))
