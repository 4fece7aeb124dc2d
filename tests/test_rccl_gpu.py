"""The RCCL leg of the multi-GPU path on the ONE GPU a test box has (replicas + one weight broadcast):
`bench.py --gpus 1 --backend nccl --force-pg` initialises the process group (backend "nccl" = RCCL on ROCm) at world size 1,
sends the 116 MB VITS weight blob through `parallel.broadcast_state_dict` on the device, checks it came back bit-identical,
and runs the barrier / MAX / SUM reductions of the timing contract through the same communicator — so that the first 8-GPU
run does not die on an init keyword or an IPC setting."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--force-pg"] + extra,
                          capture_output=True, text=True, env=env, cwd=ROOT, timeout=timeout)


def test_rccl_process_group_and_weight_broadcast_on_one_gpu(gpu):
    p = _run(["--workload", "launch_check", "--steps", "2"])
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    cfg = line["config"]
    assert cfg["backend"] == "nccl" and cfg["process_group"] is True and cfg["weights_identical"] is True
    assert cfg["weight_broadcast_bytes"] > 100e6 and 0 < cfg["weight_broadcast_s"] < 60.0       # the real VITS blob, 116 MB
    assert line["n_gpus"] == 1 and cfg["units_per_step_all_ranks"] == 1000.0
    print("RCCL world-1: %.1f MB broadcast in %.3f s" % (cfg["weight_broadcast_bytes"] / 1e6, cfg["weight_broadcast_s"]))


def test_headline_workload_runs_through_the_process_group(gpu):
    """The headline workload itself under --force-pg (two short steps): weights arrive through the RCCL broadcast, the timed
    region is fenced by the group's barrier, value / time go through its all-reduces."""
    p = _run(["--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--no-live-pmc"])
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["value"] > 1e6 and line["config"]["weight_broadcast_bytes"] > 100e6
    assert line["roofline"]["frac"] > 0.1
