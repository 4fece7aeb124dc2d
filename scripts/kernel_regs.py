"""Register / scratch / LDS footprint of every kernel in one HIP source (cross-compiles for gfx950, no GPU needed):
python scripts/kernel_regs.py tts_amd/csrc/conv_k11.hip [substring] [extra hipcc flags] -> VGPRs, AGPRs, SGPRs, scratch bytes."""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-Wno-pass-failed", "--cuda-device-only", "-S", src, "-o", d + "/k.s"] + sys.argv[3:],
                          stderr=subprocess.DEVNULL)
    txt = open(d + "/k.s").read()
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in re.split(r"\n  - ", meta)[1:]:
    f = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]  # noqa: E731
    name = f("name")
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void ttsamd::", "")
    if flt and flt not in dem:
        continue
    print("%-58s vgpr %3s agpr %3s sgpr %3s scratch %5s static_lds %6s" % (
        dem[:58], f("vgpr_count"), f("agpr_count"), f("sgpr_count"), f("private_segment_fixed_size"), f("group_segment_fixed_size")))
