#!/bin/bash
# round 6: the conv_h2 staging rework (even work split, two register sets for K < 7) against round 5 / pairs-only libraries
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_convs; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest_conv.txt
LIBS=${AB_LIBS:-"r5 o3 cur"}
for rep in 1 2; do for L in $LIBS; do
  [ "$L" = "cur" ] && lib=tts_amd/libtts_amd.so || lib=tts_amd/libtts_amd_$L.so
  TTSAMD_LIB_PATH=$lib timeout 300 python scripts/r6_pairs_ab.py ${AB_WHAT:-convs ups} 2>&1 | grep -v amdgpu.ids
done; done | tee $OUT/convs_ab.txt
sed -i 's#r6_pairs/pairs_ab.txt#r6_convs/convs_ab.txt#' /dev/null
python - <<'PY' | tee $OUT/convs_ab_summary.txt
import collections, re
rows = collections.OrderedDict()
for l in open("gpurun_out/r6_convs/convs_ab.txt"):
    m = re.match(r"(\S+)\s+(pair|conv|convT) (.*?)\s+([\d.]+) us.*?(\w{10})$", l.strip())
    if not m: continue
    rows.setdefault(m.group(2) + " " + m.group(3), collections.OrderedDict()).setdefault(m.group(1), []).append((float(m.group(4)), m.group(5)))
libs = []
for v in rows.values():
    for k in v:
        if k not in libs: libs.append(k)
print("%-36s" % "launch" + "".join("%22s" % k for k in libs) + "   digests equal")
tot = collections.Counter()
for name, v in rows.items():
    best = {k: min(t for t, _ in v[k]) for k in v}
    for k in best: tot[k] += best[k]
    dg = {d for k in v for _, d in v[k]}
    print("%-36s" % name + "".join("%22.1f" % best.get(k, float("nan")) for k in libs) + "   %s" % (len(dg) == 1))
print("%-36s" % "sum (us)" + "".join("%22.1f" % tot[k] for k in libs))
PY
[ -n "$AB_BENCH" ] && timeout 900 python scripts/bench_ab.py $AB_BENCH 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_ab.txt
