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


def test_bench_forced_process_group_world1_gloo():
    """`--force-pg`: the process group, the weight broadcast (with the bit-identity check) and the reductions also run with
    ONE rank — the switch tests/test_rccl_gpu.py uses to exercise RCCL on a single GPU, here on gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "launch_check", "--backend",
                        "gloo", "--force-pg", "--steps", "2"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["config"]["process_group"] is True and r["config"]["weights_identical"] and r["n_gpus"] == 1


def test_bench_compact_lines_fit_the_tail():
    """bench.compact_line: an extra-workload line keeps every number the contract names and fits its byte budget; the three
    closing lines of a default run (four extra workloads + the headline) stay under 7.2 KB together; the `observed` block (step
    percentiles, regime) survives compaction."""
    sys.path.insert(0, ROOT)
    import bench

    full = {"metric": "audio samples/sec (" + "x" * 200 + ")", "value": 123456.789, "unit": "samples/s", "n_gpus": 1, "steps": 50,
            "warmup": 5, "ms_per_step": 2.6745123, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (" + "d" * 90 + ")", "data": "synthetic", "rtf_x": 1423.85,
            "config": {"workload": "configs[0]: " + "w" * 300, "weights": "r" * 200, "frames": 318},
            "roofline": {"bound": "mfma", "kernel": "k" * 300, "achieved": 7.6, "peak": 416.667, "unit": "TFLOP/s", "frac": 0.0183,
                         "traffic": None, "peak_note": "n" * 500, "launches_per_request": 140},
            "cpu_baseline": {"value": 1.2e6, "unit": "samples/s", "cores": 16, "kind": "port", "host_cores": 256, "reps": 5,
                             "value_min": 1.1e6, "value_max": 1.3e6, "threads_sweep_samples_per_s": {"8": 1.0, "16": 2.0},
                             "sample": "s" * 400}}
    c = bench.compact_line(json.loads(json.dumps(full)))
    assert len(json.dumps(c)) <= bench.EXTRA_MAX_BYTES
    assert c["value"] == float("%.6g" % full["value"]) and c["ms_per_step"] == float("%.6g" % full["ms_per_step"])
    assert c["roofline"]["frac"] == 0.0183 and c["roofline"]["launches_per_request"] == 140
    assert c["cpu_baseline"]["value"] == 1.2e6 and c["cpu_baseline"]["reps"] == 5 and c["cpu_baseline"]["cores"] == 16
    head = bench.headline_json(bench.compact_line(json.loads(json.dumps(full)), limit=None))
    # four extra lines (configs[0], configs[2], the VITS B=1 request, the XTTS streaming vocoder half) + the headline: inside the
    # ~8 KB of stdout the driver's record keeps (BENCH_r04.json: stdout_tail of 8 081 bytes)
    assert len(head) <= bench.HEADLINE_MAX_BYTES and 4 * (bench.EXTRA_MAX_BYTES + 70) + bench.HEADLINE_MAX_BYTES <= 7200
    full["observed"] = {"step_ms_p50": 1.4712345, "step_ms_p90": 1.49, "step_ms_max": 1.71, "sentence_latency_ms_p50": 1.53,
                        "gpu_ms_per_sentence_p50": 1.44, "warmup_steps_run": 340, "mode": "kernel-chain-bound, fast"}
    c = bench.compact_line(json.loads(json.dumps(full)))
    assert len(json.dumps(c)) <= bench.EXTRA_MAX_BYTES and c["observed"]["step_ms_p50"] == 1.47123 and c["observed"]["mode"].endswith("fast")
