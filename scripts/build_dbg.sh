#!/bin/bash
# debug library with per-phase clock stamps in the split-bf16 conv kernel -> tts_amd/build_dbg/libtts_amd_dbg.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tts_amd/build_dbg
for f in tts_amd/csrc/*.hip; do
  o=tts_amd/build_dbg/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-pass-failed ${TTSAMD_DBG_FLAGS:--DTTSAMD_PHASE_CLOCKS} -c $f -o $o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tts_amd/build_dbg/libtts_amd_dbg.so tts_amd/build_dbg/*.o
ls -la tts_amd/build_dbg/libtts_amd_dbg.so
