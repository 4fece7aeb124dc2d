"""Does running the waveform decoder over small batch slabs (activations of a slab closer to the 256 MiB Infinity Cache)
beat one pass over the whole batch?  python scripts/slab_probe.py"""
import sys, time, torch
sys.path.insert(0, '.')
from tts_amd import synthetic as W
from tts_amd.vits import Vits
import bench
dev = torch.device("cuda:0")
m = Vits({"model_args": {}}); m.load_state_dict(W.make_vits_state({}, seed=1)); m.to(dev)
x, xl, dur = bench.synthetic_batch(32, 128, 0, dev)
o = m.inference(x, {"x_lengths": xl, "durations": dur, "run_duration_predictor": True})
z = (o["z"] * o["y_mask"]).contiguous()
dec = m.waveform_decoder
def run(slab, serial):
    dec.concurrent_branches = not serial
    outs = []
    for lo in range(0, 32, slab):
        outs.append(dec.forward(z[lo:lo + slab].contiguous()))
    return outs
for serial in (False, True):
    for slab in (32, 16, 8, 4, 2, 1):
        run(slab, serial); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): run(slab, serial)
        torch.cuda.synchronize()
        print("branches %s slab %2d: %7.2f ms per 32 utterances" % ("serial    " if serial else "concurrent", slab, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
