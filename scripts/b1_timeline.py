"""Timeline of the LAST request in a kernel trace of scripts/b1_trace_target.py (N identical eager requests).
usage: b1_timeline.py <kernel_trace.csv> N [min_us]   -> every kernel of the request in start order: offset, duration, stream,
idle gap before it (no kernel of the request running on any stream), plus a per-phase summary."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "embed" in r["Kernel_Name"]]  # a request starts with the embedding lookup
req = rows[first[-1]:] if first else rows[-(len(rows) // n):]
t0 = int(req[0]["Start_Timestamp"])
busy_end = t0
idle = 0.0
streams = {}
out = []
for r in req:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - busy_end) / 1e3
    idle += gap
    busy_end = max(busy_end, e)
    nm = re.sub(r"void ttsamd::conv1d_(mfma|x3)_kernel<(.*?)>.*", r"conv_\1<\2>", r["Kernel_Name"])
    nm = re.sub(r"void ttsamd::resblock_pair_x3_kernel<(.*?)>.*", r"resblock_x3<\1>", nm)
    nm = re.sub(r"^void ", "", re.sub(r"\(.*", "", nm)).replace("ttsamd::", "")[:40]
    q = streams.setdefault(r.get("Queue_Id", r.get("Stream_Id", "0")), len(streams))
    grid = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    out.append(((s - t0) / 1e3, (e - s) / 1e3, q, gap, nm, grid))
total = (busy_end - t0) / 1e3
print("request: %d kernels, %.1f us first start -> last end, %.1f us with no kernel running (%d%%), sum of durations %.1f us"
      % (len(req), total, idle, 100 * idle / total, sum(o[1] for o in out)))
minus = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
for off, d, q, gap, nm, grid in out:
    if d >= minus or gap >= minus:
        print("%8.1f +%7.1f us  q%d gap %6.1f  blocks %6d  %s" % (off, d, q, gap, grid, nm))
