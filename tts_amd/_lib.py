"""ctypes binding of the C ABI in include/tts_amd.h (libtts_amd.so).

north_star asks for a cffi layer; `cffi` is not installed in this image, the ABI
is plain C so `ctypes` binds the identical symbols.  There is NO CPU/PyTorch fallback: if the
HIP library is missing or a call fails this raises.
"""
import ctypes
import os
import re
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TTSAMD_LIB_PATH") or os.path.join(_HERE, "libtts_amd.so")
HEADER_PATH = os.path.join(_HERE, "..", "include", "tts_amd.h")

_lib = None


class TtsAmdError(RuntimeError):
    pass


def declared_symbols():
    """Every function name declared in include/tts_amd.h (used by the CPU-side ABI test)."""
    src = open(HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ttsamd_\w+)\s*\(", src)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TtsAmdError(
                "libtts_amd.so is missing (%s): build it with `python -m tts_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH
            )
        # torch ships its own libamdhip64.so.7; load it FIRST so that libtts_amd.so (NEEDED
        # libamdhip64.so.7) binds to the same HIP runtime instance by SONAME — streams and device
        # pointers are shared with torch.  (Two HIP runtimes in one process see no device.)
        import torch  # noqa: F401

        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ttsamd_last_error.restype = ctypes.c_char_p
        _lib.ttsamd_arch.restype = ctypes.c_char_p
        _lib.ttsamd_maximum_path_workspace_bytes.restype = ctypes.c_size_t
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().ttsamd_last_error().decode("utf-8", "replace")
        raise TtsAmdError("%s failed (rc=%d): %s" % (what or "tts_amd call", rc, msg))


def P(t):
    """Device pointer of a torch tensor (or NULL for None)."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_OWNERS = weakref.WeakValueDictionary()      # HIP stream handle -> the OwnedStream that will destroy it


def stream_owner(handle):
    """The live OwnedStream behind a raw stream handle (None for torch's own streams): whoever keeps work or graphs bound
    to the stream holds this reference so the stream outlives them."""
    return _OWNERS.get(int(handle)) if handle else None


class OwnedStream:
    """A dedicated HIP stream created through the C ABI (ttsamd_stream_create) and wrapped for torch
    (`torch.cuda.ExternalStream`): unlike `torch.cuda.Stream()`, whose objects are handed out round-robin from a pool of
    32 per priority and therefore alias each other in a long-lived process, two OwnedStreams are never the same
    underlying stream.  `.stream` is the torch view; destroyed with the owner (queued work still completes)."""

    def __init__(self, device=None, priority=0):
        import torch

        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = ctypes.c_void_p(0)
        with torch.cuda.device(self.device):
            check(lib().ttsamd_stream_create(int(priority), ctypes.byref(h)), "stream_create")
        self.handle = h.value
        self.stream = torch.cuda.ExternalStream(self.handle, device=self.device)
        _OWNERS[int(self.handle)] = self

    def __del__(self):
        try:
            import sys

            if sys.is_finalizing():          # the HIP runtime may already be gone at interpreter exit: leave the stream to it
                return
            if getattr(self, "handle", None):
                _lib.ttsamd_stream_destroy(ctypes.c_void_p(self.handle))
                self.handle = None
        except Exception:
            pass


def require_gpu(t, name="tensor"):
    if not t.is_cuda:
        raise TtsAmdError(
            "%s must live on the GPU: tts_amd runs only hand-written HIP kernels (no CPU fallback)" % name
        )
