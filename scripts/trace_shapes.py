"""Per-(kernel, grid) time breakdown from a rocprofv3 kernel trace CSV.
usage: trace_shapes.py <trace_kernel_trace.csv> <steps_total> [rows]
Durations are summarised by their MEDIAN (x launches per step): the first launch of a kernel instantiation in a process can
take tens of milliseconds (code-object upload), which a mean over ten launches would smear into the steady-state figure;
`max_us` shows it."""
import collections
import csv
import re
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
agg = collections.OrderedDict()
for r in rows:
    nm = re.sub(r"void ttsamd::conv1d_(mfma|x3|h2)_kernel<(.*?)>.*", r"conv_\1<\2>", r['Kernel_Name'])
    nm = re.sub(r"void ttsamd::resblock_pair_(x3|h2)_kernel<(.*?)>.*", r"resblock_\1<\2>", nm)
    nm = re.sub(r"^void ", "", re.sub(r"\(.*", "", nm)).replace("ttsamd::", "")[:42]
    key = (nm, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tab = [(k, len(d), statistics.median(d), max(d)) for k, d in agg.items()]
tot = sum(n * med for _, n, med, _ in tab)
print("total %.2f ms/step (sum over kernels of median duration x launches per step; %d steps in the trace)" % (tot / steps / 1e3, steps))
for k, n, med, mx in sorted(tab, key=lambda t: -t[1] * t[2])[:int(sys.argv[3]) if len(sys.argv) > 3 else 50]:
    print("%-42s grid=%5dx%3dx%3d n/step=%5.1f median_us=%9.1f max_us=%9.1f ms/step=%7.2f"
          % (k[0], k[1], k[2], k[3], n / steps, med, mx, n * med / steps / 1e3))
