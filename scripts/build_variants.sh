#!/bin/bash
# Build the experiment variants of the library HERE (hipcc cross-compiles; the .so files travel with the gpurun snapshot):
#   bash scripts/build_variants.sh && gpurun --timeout 1500 -- 'bash scripts/gpu_next_ab.sh'
set -e
cd "$(dirname "$0")/.."
python -m tts_amd.build > /dev/null
TTSAMD_BUILD_TAG=pairs   TTSAMD_EXTRA_FLAGS="-DTTSAMD_SPLIT_PAIRS=1" python -m tts_amd.build > /dev/null
TTSAMD_BUILD_TAG=noslp   TTSAMD_EXTRA_FLAGS="-fno-slp-vectorize" python -m tts_amd.build > /dev/null
TTSAMD_BUILD_TAG=pairsns TTSAMD_EXTRA_FLAGS="-DTTSAMD_SPLIT_PAIRS=1 -fno-slp-vectorize" python -m tts_amd.build > /dev/null
TTSAMD_BUILD_TAG=x3sall  TTSAMD_EXTRA_FLAGS="-DTTSAMD_X3S_ALL=1" python -m tts_amd.build > /dev/null
(cd scripts/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -Wno-unused-value att_v2.hip -o att_v2)
ls -la tts_amd/libtts_amd*.so scripts/ubench/att_v2
