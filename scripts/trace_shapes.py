"""Per-(kernel, grid) time breakdown from a rocprofv3 kernel trace CSV.  usage: trace_shapes.py <trace_kernel_trace.csv> <steps_total>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
agg = collections.OrderedDict()
for r in rows:
    nm = re.sub(r"void ttsamd::conv1d_(mfma|x3)_kernel<(.*?)>.*", r"conv_\1<\2>", r['Kernel_Name'])
    nm = re.sub(r"void ttsamd::resblock_pair_x3_kernel<(.*?)>.*", r"resblock_x3<\1>", nm)
    nm = re.sub(r"^void ", "", re.sub(r"\(.*", "", nm)).replace("ttsamd::", "")[:42]
    key = (nm, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(a[1] for a in agg.values())
print("total %.2f ms/step" % (tot / steps / 1e3))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 50]:
    print("%-42s grid=%5dx%3dx%3d n/step=%5.1f avg_us=%9.1f ms/step=%7.2f" % (k[0], k[1], k[2], k[3], a[0] / steps, a[1] / a[0], a[1] / steps / 1e3))
