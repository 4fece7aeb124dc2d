"""Multi-process (gloo, world_size 2, CPU) test of the only multi-GPU mechanism the path needs: the one-shot
flat-blob weight broadcast + utterance sharding (tts_amd/parallel.py)."""
import json
import os
import socket
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_broadcast_and_shard_world2():
    port = _free_port()
    worker = os.path.join(ROOT, "tests", "_parallel_worker.py")
    env = dict(os.environ, PYTHONPATH=ROOT)
    ps = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, env=env, cwd=ROOT) for r in range(2)]
    res = {}
    for p in ps:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        line = [ln for ln in out.splitlines() if ln.startswith("RESULT ")][-1]
        r = json.loads(line[7:])
        res[r["rank"]] = r
    assert res[0]["ok"] and res[1]["ok"], "broadcast state_dict differs from the source"
    assert (res[0]["lo"], res[0]["hi"], res[1]["lo"], res[1]["hi"]) == (0, 17, 17, 33)


def test_flatten_roundtrip_and_length_sharding():
    sys.path.insert(0, ROOT)
    from tts_amd import parallel

    sd = {"a.weight": torch.randn(3, 4, 5), "b.bias": torch.randn(7), "n": torch.tensor([3], dtype=torch.int64)}
    blob, man = parallel.flatten_state_dict(sd)
    back = parallel.unflatten_state_dict(blob, man)
    assert all(torch.equal(back[k], sd[k]) and back[k].dtype == sd[k].dtype for k in sd)
    lens = [10, 50, 20, 40, 30, 60, 5]
    shards = parallel.shard_by_length(lens, 3)
    assert sorted(sum(shards, [])) == list(range(7)) and max(map(len, shards)) - min(map(len, shards)) <= 1
    tot = [sum(lens[i] for i in s) for s in shards]
    assert max(tot) - min(tot) <= max(lens)


def test_bench_self_launch_world2_gloo():
    """`python bench.py --gpus 2` outside torchrun re-launches itself as 2 ranks (torch.distributed.run, 127.0.0.1) and
    rank 0 prints ONE JSON line with n_gpus=2: the launcher, process group, flat-blob weight broadcast, barrier-fenced
    timed region, MAX-over-ranks time and SUM-over-ranks units — the skeleton every GPU workload runs on — on gloo/CPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "launch_check",
                        "--backend", "gloo", "--steps", "3"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    # output contract: bulky {"detail": ...} tables first, the headline line LAST and compact (the driver keeps ~10 KB of
    # tail: round 2's single 21 KB line lost its head and nothing parsed), carrying roofline.frac and cpu_baseline
    assert len(lines) >= 2 and all("detail" in json.loads(ln) for ln in lines[:-1]), p.stdout[-2000:]
    assert len(lines[-1]) < 4096 and p.stdout.rstrip().endswith(lines[-1])
    assert sum(len(ln) for ln in lines[:-1]) > 10000            # the filler table really is bulky
    r = json.loads(lines[-1])
    assert isinstance(r["roofline"]["frac"], float) and "traffic" in r["roofline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(r["cpu_baseline"])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["config"]["parallelism"] == "replicas x2"
    assert r["config"]["weights_identical"] and r["config"]["weight_broadcast_bytes"] > 1e6
    assert r["config"]["units_per_step_all_ranks"] == 2001.0                  # SUM over ranks
    assert r["ms_per_step"] >= 20.0 * 0.99                                     # MAX over ranks: rank 1 sleeps 20 ms per step
    # a rank count that does not match --gpus is refused, never silently measured as something else
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "launch_check"],
                       capture_output=True, text=True, env=dict(env, WORLD_SIZE="2", RANK="0"), cwd=ROOT, timeout=120)
    assert p.returncode != 0 and "--gpus 1" in (p.stderr + p.stdout)


def test_bench_headline_stays_compact():
    """bench.headline_json: whatever explanatory strings a workload attaches, the last line stays under the limit and no
    number is dropped."""
    sys.path.insert(0, ROOT)
    import bench

    line = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
            "config": {"workload": "w", "weights": "x" * 3000},
            "roofline": {"bound": "mfma", "achieved": 1.0, "peak": 2.0, "frac": 0.5, "traffic": None, "peak_note": "y" * 3000,
                         "measured": "z" * 500},
            "cpu_baseline": {"value": 1.0, "unit": "u", "cores": 8, "kind": "port", "sample": "s" * 3000}}
    out = bench.headline_json(line)
    assert len(out) <= bench.HEADLINE_MAX_BYTES
    r = json.loads(out)
    assert r["roofline"]["frac"] == 0.5 and r["cpu_baseline"]["value"] == 1.0 and r["value"] == 1.0
    # the stamp that ties a committed PMC file to the running kernel sources is stable and 16 hex digits
    assert bench.code_stamp() == bench.code_stamp() and len(bench.code_stamp()) == 16
