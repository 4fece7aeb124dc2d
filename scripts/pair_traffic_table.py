"""HBM traffic of every fused ResBlock-pair instantiation of the headline step against its ALGORITHMIC bytes ("2x fetch
on the dilation-5 pairs"):   python scripts/pair_traffic_table.py <fetch.csv> <write.csv>     (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
over `bench.py --serial-branches --lanes 1`).
Algorithmic bytes of a launch = x read once + y written once, PLUS the MRF accumulate operand read once where the launch has one: the last
iteration (dilation 5) of the 2nd and 3rd branch (kernel 7 and 11) adds the running sum z_sum of the branches before it
(hifigan_generator.py:255-261; tts_amd/hifigan.py: `accum = zsum if j > 0`) — that, not a re-read, is the second tensor the d = 5 pairs fetch.
Shapes of the headline step (B = 32): C = 128 -> T = 49 280, C = 64 -> 98 560, C = 32 -> 197 120: 807.4 MB per tensor pass each."""
import collections
import csv
import re
import sys

T_OF_C = {128: 49280, 64: 98560, 32: 197120}
B = 32


def load(path, ctr):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != ctr:
            continue
        m = re.search(r"resblock_pair_(x3|h2)_kernel<([^>]*)>", r["Kernel_Name"].replace(" ", ""))
        if m:
            agg[(m.group(1), m.group(2))][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (sum(d.values()) / len(d), len(d)) for k, d in agg.items()}


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("%-34s %4s %10s %10s %10s %8s %7s   %s" % ("kernel<K,D,C,WM,WN,NI>", "n", "fetch MB", "write MB", "algo MB", "hbm/algo", "accum", "note"))
worst = 0.0
for key in sorted(fetch, key=lambda k: [int(v) for v in k[1].split(",")][2::-1]):
    k, d, c = [int(v) for v in key[1].split(",")][:3]
    if c not in T_OF_C:
        continue
    tensor = 4.0 * B * c * T_OF_C[c] / 1e6
    has_accum = d == 5 and k in (7, 11)
    algo = tensor * (2 + has_accum)
    f, w = fetch[key][0] * 1024 * 2 / 1e6, write.get(key, (0, 0))[0] * 1024 / 1e6
    ratio = (f + w) / algo
    worst = max(worst, ratio)
    print("%-34s %4d %10.1f %10.1f %10.1f %8.3f %7s   %s" % ("resblock_pair_%s<%s>" % key, fetch[key][1], f, w, algo, ratio, "yes" if has_accum else "-",
                                                              "x + z_sum read, y written" if has_accum else "x read, y written"))
print("largest measured / algorithmic ratio: %.3f" % worst)
