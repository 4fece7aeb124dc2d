"""Same-box A/B of library variants on the headline bench line (each variant twice, interleaved):
  TTSAMD_BUILD_TAG=<tag> TTSAMD_EXTRA_FLAGS="<flags>" python -m tts_amd.build      # -> tts_amd/libtts_amd_<tag>.so
  python scripts/bench_ab.py tts_amd/libtts_amd.so tts_amd/libtts_amd_<tag>.so ...
e.g. <flags> = -fno-slp-vectorize (scalar instead of packed f32 VALU ops in the staging code) or a
-DTTSAMD_X3_CFG128=... tile arrangement."""
import sys, time, os, subprocess, json
libs = sys.argv[1:]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ, TTSAMD_LIB_PATH=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print("%-28s %.2f ms/step  rtf_x=%.0f  dominant=%.1f TF  allconv=%.1f TF" % (os.path.basename(lib), d["ms_per_step"], d["rtf_x"], d["roofline"]["achieved"], d["roofline"]["all_conv_launches"]["tflops"]), flush=True)
