#!/bin/bash
# round 6: VALU diet of the three-product kernels (zero-C first MFMA instead of accumulator zero-init, row offsets in the loads' scalar
# operand, fma for the 2^-11 merge): d0 = before, cur = after; bitwise digests, then the headline step
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_diet; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_resblock_gpu.py tests/test_conv_gpu.py tests/test_vits_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest.txt
for rep in 1 2; do for L in d0 cur; do
  [ "$L" = "cur" ] && lib=tts_amd/libtts_amd.so || lib=tts_amd/libtts_amd_$L.so
  TTSAMD_LIB_PATH=$lib timeout 400 python scripts/r6_pairs_ab.py pairs convs ups 2>&1 | grep -v amdgpu.ids
done; done > $OUT/ab.txt
python - <<'PY' | tee $OUT/ab_summary.txt
import collections, re
rows = collections.OrderedDict()
for l in open("gpurun_out/r6_diet/ab.txt"):
    m = re.match(r"(\S+)\s+(pair|conv|convT) (.*?)\s+([\d.]+) us.*?(\w{10})$", l.strip())
    if not m: continue
    rows.setdefault(m.group(2) + " " + m.group(3), collections.OrderedDict()).setdefault(m.group(1), []).append((float(m.group(4)), m.group(5)))
libs = []
for v in rows.values():
    for k in v:
        if k not in libs: libs.append(k)
print("%-36s" % "launch" + "".join("%22s" % k for k in libs) + "   ratio  digests equal")
tot = collections.Counter()
for name, v in rows.items():
    best = {k: min(t for t, _ in v[k]) for k in v}
    for k in best: tot[k] += best[k]
    dg = {d for k in v for _, d in v[k]}
    print("%-36s" % name + "".join("%22.1f" % best.get(k, float("nan")) for k in libs) + "   %.3f  %s" % (best[libs[-1]] / best[libs[0]], len(dg) == 1))
print("%-36s" % "sum (us)" + "".join("%22.1f" % tot[k] for k in libs) + "   %.3f" % (tot[libs[-1]] / tot[libs[0]]))
PY
timeout 900 python scripts/bench_ab.py tts_amd/libtts_amd_d0.so tts_amd/libtts_amd.so 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_ab.txt
