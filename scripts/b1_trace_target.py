"""rocprofv3 target: N eager VITS requests at batch B (default 1), each followed by the waveform D2H.
rocprofv3 --kernel-trace -d out -o b1 --output-format csv -- python scripts/b1_trace_target.py [B] [N] [graph]
then: python scripts/b1_timeline.py out/b1_kernel_trace.csv N"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from tts_amd import synthetic as W  # noqa: E402
from tts_amd.vits import Vits  # noqa: E402

dev = torch.device("cuda:0")
m = Vits({"model_args": {}})
m.load_state_dict(W.make_vits_state({}, seed=1))
m.to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
x, xl, dur = bench.synthetic_batch(B, 128, 0, dev)
graph = len(sys.argv) > 3 and sys.argv[3] == "graph"          # default: eager launches; "graph": the two-graph replay path
aux = {"x_lengths": xl, "durations": dur, "run_duration_predictor": True, "ragged_exact": B > 1, "no_graph": not graph}
for _ in range(N):
    m.inference(x, aux)["model_outputs"].cpu()
torch.cuda.synchronize()
