for rep in 1 2; do for L in "" _a22e _a32e _a24e _a32; do echo "== lib$L"; TTSAMD_ATT_ONLY=v3 TTSAMD_LIB_PATH=tts_amd/libtts_amd$L.so python scripts/att_ab.py 2>&1 | grep "^v3"; done; done
