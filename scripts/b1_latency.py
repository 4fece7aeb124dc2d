import sys, time, torch
sys.path.insert(0, '.')
from tts_amd import synthetic as W
from tts_amd.vits import Vits
import bench
dev = torch.device("cuda:0")
m = Vits({"model_args": {}}); m.load_state_dict(W.make_vits_state({}, seed=1)); m.to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x, xl, dur = bench.synthetic_batch(B, 128, 0, dev)
aux = {"x_lengths": xl, "durations": dur, "run_duration_predictor": True}
for _ in range(3): m.inference(x, aux)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): o = m.inference(x, aux)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("B=%d: %.2f ms per call, rtf_x=%.0f" % (B, dt * 1e3, B * 197120 / 22050 / dt))
# host-only cost: time to ISSUE the decoder launches (no sync)
z = o["z"]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): w = m.waveform_decoder.forward(z)
t_issue = (time.perf_counter() - t0) / n
torch.cuda.synchronize(); t_total = (time.perf_counter() - t0) / n
print("decoder: host issue %.2f ms, total %.2f ms per call" % (t_issue * 1e3, t_total * 1e3))
