"""TEST INFRASTRUCTURE: compile the reference's Cython MAS core (core.pyx) into oracle/_ref/.

Sources are read where they lie under /root/reference; only build products (generated .c,
.so) are written, and only under oracle/_ref/ (git-ignored, travels with gpurun).
Flags mirror the reference's setup.py:74-79 (no -fopenmp => prange is serial).
"""
import os
import subprocess
import sys
import sysconfig

import numpy as np


def main(ref_root: str, out_dir: str) -> None:
    out_dir = os.path.abspath(out_dir)
    os.makedirs(out_dir, exist_ok=True)
    pyx = os.path.join(ref_root, "TTS/tts/utils/monotonic_align/core.pyx")
    c_out = os.path.join(out_dir, "ref_mas_core.c")
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    so_out = os.path.join(out_dir, "ref_mas_core" + ext)
    if os.path.exists(so_out) and os.path.getmtime(so_out) > os.path.getmtime(pyx):
        return
    # module name must match the init symbol -> cythonize under the name ref_mas_core
    subprocess.check_call([sys.executable, "-m", "cython", "-3", "--module-name", "ref_mas_core", pyx, "-o", c_out])
    inc = sysconfig.get_paths()["include"]
    subprocess.check_call(
        ["gcc", "-O2", "-fPIC", "-shared", "-w", "-I", inc, "-I", np.get_include(), c_out, "-o", so_out]
    )
    print("built", so_out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
