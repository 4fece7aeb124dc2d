"""Request-lanes stress (the round-2 stall: tests/test_vits_gpu.py::test_vits_request_lanes_equal_single_stream hung once in
a full-suite run): N rounds of six requests issued round-robin on two lanes, results compared bitwise with the single-stream
run, the per-lane hipGraph captures dropped every few rounds so that capture happens again and again while the other lane
has work queued.  A watchdog dumps every Python stack and exits if a round takes longer than 60 s.
    python scripts/lanes_stress.py [rounds=200] [lanes=2] [recapture_every=5]"""
import faulthandler
import sys
import time

import torch

sys.path.insert(0, ".")
from tts_amd import parallel  # noqa: E402
from tts_amd import synthetic as W  # noqa: E402
from tts_amd.vits import Vits  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    nl = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    gpu = torch.device("cuda:0")
    args = dict(upsample_initial_channel_decoder=64)
    m = Vits({"model_args": args})
    m.load_state_dict(W.make_vits_state(args, seed=5))
    m.to(gpu)
    g = torch.Generator().manual_seed(3)
    reqs = []
    for i in range(6):
        B, T = (1 if i == 4 else 2 + i % 3), 30 + 7 * (i % 2)          # one B = 1 request: the tail graph too
        x = torch.randint(0, 100, (B, T), generator=g).to(gpu)
        dur = (2 + (torch.arange(T) % 3)).float().repeat(B, 1).to(gpu)
        aux = {"x_lengths": torch.full((B,), T, dtype=torch.int64, device=gpu), "durations": dur,
               "run_duration_predictor": True, "noise_dp": torch.randn(B, 2, T, generator=g).to(gpu),
               "noise_z": torch.randn(B, 192, int(dur[0].sum()), generator=g).to(gpu)}
        reqs.append((x, aux))
    want = [m.inference(x, dict(aux, no_graph=True))["model_outputs"].clone() for x, aux in reqs]
    torch.cuda.synchronize()
    lanes = parallel.Lanes(nl, device=gpu)
    t0 = time.time()
    for r in range(rounds):
        faulthandler.dump_traceback_later(60, exit=True)
        if every and r % every == 0:
            m._front.clear()
            m._tail.clear()
        outs = [lanes.run(m.inference, x, aux) for x, aux in reqs]
        lanes.sync()
        for k, (o, w) in enumerate(zip(outs, want)):
            assert torch.equal(o["model_outputs"], w), (r, k)
        faulthandler.cancel_dump_traceback_later()
    print("lanes stress OK: %d rounds x %d requests on %d lanes in %.1f s; front captures %d replays %d, tail captures %d replays %d"
          % (rounds, len(reqs), nl, time.time() - t0, m._front.stats["captures"], m._front.stats["replays"],
             m._tail.stats["captures"], m._tail.stats["replays"]), flush=True)


if __name__ == "__main__":
    main()
