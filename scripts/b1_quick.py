"""VITS B=1 request latency, graphed path only (waveform left on the device; synchronised per request): python scripts/b1_quick.py [n]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from tts_amd import synthetic as W  # noqa: E402
from tts_amd.vits import Vits  # noqa: E402

dev = torch.device("cuda:0")
m = Vits({"model_args": {}})
m.load_state_dict(W.make_vits_state({}, seed=1))
m.to(dev)
x, xl, dur = bench.synthetic_batch(1, 128, 0, dev)
aux = {"x_lengths": xl, "durations": dur, "run_duration_predictor": True}
for _ in range(5):
    m.inference(x, aux)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lat = []
for _ in range(n):
    t0 = time.perf_counter()
    m.inference(x, aux)
    torch.cuda.synchronize()
    lat.append((time.perf_counter() - t0) * 1e3)
lat.sort()
print("VITS B=1 request: p50 %.3f ms  p10 %.3f  p90 %.3f" % (lat[n // 2], lat[n // 10], lat[(9 * n) // 10]))
import ctypes  # noqa: E402

from tts_amd import _lib  # noqa: E402

L = _lib.lib()
L.ttsamd_launch_count.restype = ctypes.c_uint64
m.inference(x, dict(aux, no_graph=True))
torch.cuda.synchronize()
n0 = int(L.ttsamd_launch_count())
m.inference(x, dict(aux, no_graph=True))
torch.cuda.synchronize()
print("kernel launches per request (library launches of one eager run of the same chain): %d" % (int(L.ttsamd_launch_count()) - n0))
