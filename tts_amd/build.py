"""Build libtts_amd.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

The shared library has no torch / Python dependency: it is the C-ABI drop-in boundary
(include/tts_amd.h).  `python -m tts_amd.build` or `__graft_entry__.build()` runs this.
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# TTSAMD_BUILD_TAG=<tag> + TTSAMD_EXTRA_FLAGS="-D..." build a variant library libtts_amd_<tag>.so next to the default one
# (A/B measurements: point TTSAMD_LIB_PATH at it); the default build takes neither.
TAG = os.environ.get("TTSAMD_BUILD_TAG", "")
OBJ = os.path.join(HERE, "build" + ("_" + TAG if TAG else ""))
LIB = os.path.join(HERE, "libtts_amd%s.so" % ("_" + TAG if TAG else ""))
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-result", "-Wno-pass-failed"] + os.environ.get("TTSAMD_EXTRA_FLAGS", "").split()


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.splitext(os.path.basename(s))[0] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (s, r.stdout, r.stderr))
        return r.stderr

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for warn in ex.map(cc, jobs):
                if verbose and warn:
                    print(warn)
    if jobs or force or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
