import sys, time, torch
sys.path.insert(0, '.')
from tts_amd import synthetic as W
from tts_amd.vits import Vits
import bench
dev = torch.device('cuda:0')
sd = W.make_vits_state({}, seed=1234)
m = Vits({"model_args": {}}); m.load_state_dict(sd); m.to(dev)
x, xl, dur = bench.synthetic_batch(32, 128, 0, dev)
aux = {"x_lengths": xl, "durations": dur, "run_duration_predictor": True}
out = m.inference(x, aux)
z = out["z"]
print("z stats", z.abs().max().item(), z.std().item(), torch.isfinite(z).all().item())
print("wav stats", out["model_outputs"].abs().max().item(), out["model_outputs"].std().item())
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
r = torch.randn_like(z)
print("decoder(randn)    ms", timeit(lambda: m.waveform_decoder.forward(r)))
print("decoder(z)        ms", timeit(lambda: m.waveform_decoder.forward(z)))
print("decoder(z*0.1)    ms", timeit(lambda: m.waveform_decoder.forward(z * 0.1)))
print("decoder(randn*std) ms", timeit(lambda: m.waveform_decoder.forward(r * z.std())))
print("decoder(zeros)    ms", timeit(lambda: m.waveform_decoder.forward(torch.zeros_like(z))))
print("full inference    ms", timeit(lambda: m.inference(x, aux)))
print("decoder(randn) again ms", timeit(lambda: m.waveform_decoder.forward(r)))
