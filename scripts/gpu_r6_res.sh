#!/bin/bash
# round 6: the residual of the fused pairs requested WITH the staging loads (e0 = before: requested before conv2 / in the output
# epilogue, x fetched twice from beyond the L2): timing + digests, tests, FETCH / WRITE per instantiation; then the MAS column step A/B.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_res; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_resblock_gpu.py tests/test_hifigan_gpu.py tests/test_mas_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest.txt
for rep in 1 2; do for L in e0 cur; do
  [ "$L" = "cur" ] && lib=tts_amd/libtts_amd.so || lib=tts_amd/libtts_amd_$L.so
  TTSAMD_LIB_PATH=$lib timeout 400 python scripts/r6_pairs_ab.py pairs 2>&1 | grep -v amdgpu.ids
done; done > $OUT/ab.txt
python - <<'PY' | tee $OUT/ab_summary.txt
import collections, re
rows = collections.OrderedDict()
for l in open("gpurun_out/r6_res/ab.txt"):
    m = re.match(r"(\S+)\s+(pair .*?|conv .*?|convT .*?)\s+([\d.]+) us.*?(\w+)\s*$", l.rstrip())
    if m: rows.setdefault(m.group(2).strip(), collections.defaultdict(list))[m.group(1)].append((float(m.group(3)), m.group(4)))
libs = ["libtts_amd_e0.so", "libtts_amd.so"]
print("%-28s %18s %18s %7s  %s" % ("launch", *libs, "ratio", "digests equal"))
tot = [0.0, 0.0]
for k, d in rows.items():
    t = [min(x[0] for x in d[l]) for l in libs]
    tot[0] += t[0]; tot[1] += t[1]
    print("%-28s %18.1f %18.1f %7.3f  %s" % (k, t[0], t[1], t[1] / t[0], len({x[1] for l in libs for x in d[l]}) == 1))
print("%-28s %18.1f %18.1f %7.3f" % ("sum (us)", tot[0], tot[1], tot[1] / tot[0]))
PY
timeout 900 python scripts/bench_ab.py tts_amd/libtts_amd_e0.so tts_amd/libtts_amd.so 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_ab.txt
cd /tmp && export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 600 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -o p -- python $R/scripts/r6_pairs_ab.py pairs > $OUT/pmc$i.log 2>&1; echo "pmc$i rc=$?"
  cp $(find $OUT/pmc$i -name '*counter_collection.csv' | head -1) $OUT/pmc$i.csv 2>/dev/null; rm -rf $OUT/pmc$i
done
(cd $R && python scripts/pair_traffic_table.py $OUT/pmc1.csv $OUT/pmc2.csv | tee $OUT/pair_traffic.txt)
rm -f $OUT/*.csv $OUT/pmc*.log
cd $R; bash scripts/gpu_r6_mas.sh
