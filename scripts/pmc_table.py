"""rocprofv3 counter_collection.csv -> per-kernel averages per dispatch:  python scripts/pmc_table.py <csv> [<csv> ...]"""
import collections
import csv
import re
import sys

agg = collections.OrderedDict()
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "ttsamd" not in name:
            continue
        name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name)).replace("ttsamd::", "")
        k = (name[:64], r["Grid_Size"])
        d = agg.setdefault(k, collections.defaultdict(lambda: [0.0, 0]))
        d[r["Counter_Name"]][0] += float(r["Counter_Value"])
        d[r["Counter_Name"]][1] += 1
for (name, grid), d in agg.items():
    print("%s grid=%s" % (name, grid))
    for c in sorted(d):
        print("    %-34s %14.5g  (avg of %d dispatches)" % (c, d[c][0] / d[c][1], d[c][1]))
