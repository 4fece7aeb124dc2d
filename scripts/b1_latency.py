"""Latency of one request at small batch (VitsArgs defaults, 128-char utterance = 257 ids, 770 frames):
python scripts/b1_latency.py [B] -> per call wall time and host-issue time, eager launches vs the two-graph path."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from tts_amd import synthetic as W  # noqa: E402
from tts_amd.vits import Vits  # noqa: E402

# time the host spends BLOCKED in the request's one device sync (y_lengths.max().item()): "inside inference()" minus this is
# what the host itself costs (issue of the launches / graph replays, output clones)
_item = torch.Tensor.item
_blocked = [0.0]


def _timed_item(self):
    t = time.perf_counter()
    v = _item(self)
    _blocked[0] += time.perf_counter() - t
    return v


torch.Tensor.item = _timed_item
dev = torch.device("cuda:0")
m = Vits({"model_args": {}})
m.load_state_dict(W.make_vits_state({}, seed=1))
m.to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x, xl, dur = bench.synthetic_batch(B, 128, 0, dev)
for mode, extra in (("eager tail (front end graphed)", {"no_graph_tail": True}), ("two graphs (front + tail)", {}),
                    ("all eager", {"no_graph": True})):
    aux = dict({"x_lengths": xl, "durations": dur, "run_duration_predictor": True, "ragged_exact": B > 1}, **extra)
    if "no_graph_tail" in extra:
        m.graph_tail_max_frames = 0
    else:
        m.graph_tail_max_frames = 1 << 20
    for _ in range(4):
        m.inference(x, aux)
    torch.cuda.synchronize()
    n, t_issue = 20, 0.0
    _blocked[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        t1 = time.perf_counter()
        o = m.inference(x, aux)
        t_issue += time.perf_counter() - t1
        o["model_outputs"].cpu()
    dt = (time.perf_counter() - t0) / n
    print("B=%d %-32s %.2f ms per request (waveform on the host), %.2f ms inside inference() of which %.2f blocked in the duration "
          "sync => host issue %.2f ms, rtf_x=%.0f"
          % (B, mode, dt * 1e3, t_issue / n * 1e3, _blocked[0] / n * 1e3, (t_issue - _blocked[0]) / n * 1e3,
             B * 197120 / 22050 / dt), flush=True)

# request THROUGHPUT with several requests in flight (tts_amd.parallel.Lanes; inside a lane the generator runs its MRF branches
# on the lane's own stream): ms per request over 40 back-to-back requests
from tts_amd import parallel  # noqa: E402

aux = {"x_lengths": xl, "durations": dur, "run_duration_predictor": True, "ragged_exact": B > 1}
m.graph_tail_max_frames = 1 << 20
for nl, pr in ((1, -1), (2, -1), (2, 0), (3, -1)):
    lanes = parallel.Lanes(nl, device=dev, priority=pr)
    for _ in range(6):
        lanes.run(m.inference, x, aux)
    lanes.sync()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        lanes.run(m.inference, x, aux)
    lanes.sync(timeout_s=60.0)
    dt = (time.perf_counter() - t0) / n
    print("B=%d %d lane(s), priority %2d: %.2f ms per request (%.0f requests/s, rtf_x=%.0f)"
          % (B, nl, pr, dt * 1e3, 1.0 / dt, B * 197120 / 22050 / dt), flush=True)
