#!/bin/bash
# round 6: the block -> XCD order on the latency-bound lines (XTTS first chunk, one Glow-TTS + HiFiGAN-v2 sentence, VITS B = 1),
# same box: xcdold = rounds 2-5 order, nowide = x-local only, default = x-local + weights-local for 16+ row blocks
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_xcd2; mkdir -p $OUT; cd $R
for rep in 1 2; do for v in xcdold nowide default; do
  L=tts_amd/libtts_amd_$v.so; [ $v = default ] && L=tts_amd/libtts_amd.so
  for w in xtts_stream glow_hifigan_v2 vits_b1; do
    echo -n "$v $w: "; TTSAMD_LIB_PATH=$R/$L timeout 300 python bench.py --workload $w --no-cpu-baseline --no-live-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d.get('ms_per_step'))"
  done
done; done | tee $OUT/ab.txt
