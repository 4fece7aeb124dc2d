#!/bin/bash
# round 6: same-box A/B of the fused-pair kernels at 2 / 3 / 4 waves per SIMD (libs: r5 = round-5 code, o2 / o3 = the new kernel
# capped at 2 / 3 waves per SIMD, cur = up to 4), digests for bitwise agreement, then the headline step on r5 vs cur.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_pairs; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_resblock_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest_resblock.txt
LIBS=${AB_LIBS:-"r5 o2 o3 cur"}
for rep in 1 2; do for L in $LIBS; do
  [ "$L" = "cur" ] && lib=tts_amd/libtts_amd.so || lib=tts_amd/libtts_amd_$L.so
  TTSAMD_LIB_PATH=$lib timeout 300 python scripts/r6_pairs_ab.py ${AB_WHAT:-pairs} 2>&1 | grep -v amdgpu.ids
done; done | tee $OUT/pairs_ab.txt
python - <<'PY' | tee $OUT/pairs_ab_summary.txt
import collections, re
rows = collections.OrderedDict()
for l in open("gpurun_out/r6_pairs/pairs_ab.txt"):
    m = re.match(r"(\S+)\s+(pair|conv|convT) (.*?)\s+([\d.]+) us.*?(\w{10})$", l.strip())
    if not m: continue
    rows.setdefault(m.group(2) + " " + m.group(3), collections.OrderedDict()).setdefault(m.group(1), []).append((float(m.group(4)), m.group(5)))
libs = []
for v in rows.values():
    for k in v:
        if k not in libs: libs.append(k)
print("%-28s" % "launch" + "".join("%22s" % k for k in libs) + "   digests equal")
tot = collections.Counter()
for name, v in rows.items():
    best = {k: min(t for t, _ in v[k]) for k in v}
    for k in best: tot[k] += best[k]
    dg = {d for k in v for _, d in v[k]}
    print("%-28s" % name + "".join("%22.1f" % best.get(k, float("nan")) for k in libs) + "   %s" % (len(dg) == 1))
print("%-28s" % "sum (us)" + "".join("%22.1f" % tot[k] for k in libs))
PY
[ -n "$AB_BENCH" ] && timeout 900 python scripts/bench_ab.py $AB_BENCH 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_ab.txt
