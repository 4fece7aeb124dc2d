#!/bin/bash
# kernel trace of the configs[0] sentence (Glow-TTS + HiFiGAN-v2, B=1): per-kernel totals of the last sentence
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/glow; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o g -- python $R/bench.py --workload glow_hifigan_v2 --steps 6 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1
T=$(find $OUT/tr -name '*kernel_trace.csv' | head -1)
python - "$T" <<'PY'
import csv, re, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "embed" in r["Kernel_Name"]]
req = rows[first[-1]:]
t0 = int(req[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in req)
agg = collections.OrderedDict()
for r in req:
    nm = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"])).replace("ttsamd::", "")[:60]
    a = agg.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("last sentence: %d kernels, %.1f us first start -> last end, sum of durations %.1f us" % (len(req), (t1 - t0) / 1e3, sum(v[1] for v in agg.values())))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print("  %-62s n=%3d total %7.1f us avg %6.1f" % (k, v[0], v[1], v[1] / v[0]))
PY
rm -rf $OUT/tr
