#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): audio samples/s + real-time factor of LJSpeech-shaped VITS
end-to-end inference (text ids -> waveform, 22.05 kHz) on MI355X.

    python bench.py --gpus N --steps K --warmup W
        N>1 without a torchrun environment: bench.py re-launches itself as N ranks under torch.distributed.run
        (one process per GPU, RCCL); under torchrun (WORLD_SIZE set) it is one of those ranks and --gpus must match.
    python bench.py --workload glow_hifigan_v2 | hifigan_v1 | mas | xtts_stream   (the other BASELINE configs)

A "step" = one full `Vits.inference` pass (text encoder, stochastic duration predictor, prior expansion incl.
both random draws, 4 coupling flows, HiFiGAN waveform decoder) over a batch of 32 synthetic utterances of
128 characters (257 token ids with blanks).  Token ids are resident in HBM when the
timed region starts; random-init weights of the exact `VitsArgs` default architecture (no network => no
released checkpoint).  Output length is pinned with the synthetic duration pattern 2+(t mod 3) (770 frames =
197 120 samples per utterance) so the workload is identical run to run — the duration predictor still runs
inside the timed region, nothing is skipped.

N GPUs = N independent replicas (one process per GPU), each with its own 32-utterance shard (weak scaling);
the only collective is the one-time weight broadcast from rank 0 (RCCL), outside the timed region (its time and
size are reported in `config`).

Output contract: the LAST line rank 0 prints is the headline JSON line of the selected workload — compact (< 4 KB:
metric/value/unit/... + "roofline" + "cpu_baseline", no tables).  Everything else comes BEFORE it, one JSON object per
line: {"detail": ...} lines carry the per-kernel tables of the roofline pass, and at N=1 the default invocation also
measures BASELINE configs[0] (Glow-TTS + HiFiGAN-v2, one 64-char sentence) and configs[2] (HiFiGAN-v1 vocoder only,
256 x 8192-frame mels), each in its own fresh process after the headline's timed region, and prints each as its OWN line
({"extra_workload": ...}; `--no-extras` skips them).  Every line carries its CPU baseline: the oracle on the host cores at the best of a
thread-count sweep (thread count and core count stated).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SAMPLE_RATE = 22050
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_bf16 dense peak (no sparsity)
PEAK_HBM_GBPS = 8000.0          # same guide: HBM3E spec peak (a float4 copy reaches 6.29 TB/s)
X3_PRODUCTS = 6                 # bf16 MFMA products issued per fp32 product by the split-bf16 kernels (conv_kernel_x3.h)
H2_PRODUCTS = 3                 # fp16 MFMA products per fp32 product of the two-part fp16 kernels (conv_kernel_h2.h); same dense peak
DTYPE = {"h2": "f32 (conv products of the large-grid launches: fp32 operands split 2-way into fp16 with exact power-of-two "
               "pre-scales, 3 fp16-MFMA products each in two fp32 accumulators; small-grid launches: the 3-way bf16 / 6-product split)",
         "x3": "f32 (conv products: fp32 operands split 3-way into bf16, 6 bf16-MFMA products each, fp32 accumulate)",
         "f32": "f32"}
PRODUCTS = {"h2": H2_PRODUCTS, "x3": X3_PRODUCTS}


def conv_peak(precision):
    """Peak the conv kernels are priced against, in ALGORITHMIC (fp32-equivalent) TFLOP/s: the split kernels issue three
    (two-part fp16) or six (three-part bf16) 16-bit MFMA products per fp32 product, so their ceiling is the 16-bit dense peak
    divided by that."""
    return PEAK_BF16_MFMA_TFLOPS / PRODUCTS[precision] if precision in PRODUCTS else PEAK_FP32_MFMA_TFLOPS


def conv_kernel_name(precision, tmpl):
    return {"h2": "ttsamd::conv1d_h2_kernel<%s>", "x3": "ttsamd::conv1d_x3_kernel<%s>"}.get(precision, "ttsamd::conv1d_mfma_kernel<%s>") % tmpl


# dominant kernel instantiation of the headline step per precision: template arguments and the substring its dispatches carry
DOMINANT_TMPL = {"h2": "11,1,1,4,4,1,0", "x3": "11,1,1,4,4,1,0", "f32": "11,1,2,2,2,2,0"}
DOMINANT_SUB = {"h2": "conv1d_h2_kernel<11,1,1,4,4,1,0>", "x3": "conv1d_x3_kernel<11,1,1,4,4,1,0,", "f32": "conv1d_mfma_kernel<11,1,2,2,2,2,0>"}


# ---------------------------------------------------------------------------------------------------------------------
# launcher + distributed context (replicas, one process per GPU, one weight broadcast, no data-path collective)
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """`python bench.py --gpus N` outside torchrun: become the launcher — N ranks of this same script under
    torch.distributed.run on 127.0.0.1 — and exit with its return code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(args.gpus, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env, cwd=ROOT)


class Ctx:
    """One rank of the job: device, process group, and the timing contract (barrier + synchronize on both sides of the
    timed region, MAX over ranks of the elapsed time, SUM over ranks of the units processed)."""

    def __init__(self, args):
        import torch.distributed as dist

        self.dist = dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, self.world))
        self.backend = args.backend or ("nccl" if torch.cuda.is_available() else "gloo")
        self.gpu = torch.cuda.is_available() and self.backend != "gloo"
        if self.gpu:
            n_dev = torch.cuda.device_count()
            if n_dev < self.world or self.local_rank >= n_dev:
                raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (self.world, n_dev))
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
        else:
            self.dev = torch.device("cpu")
        # --force-pg: the process group (RCCL communicator, barrier, all-reduce, weight broadcast) also at world size 1 —
        # the multi-GPU code path on the one GPU a test box has
        self.pg = self.world > 1 or bool(getattr(args, "force_pg", False))
        if self.pg:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            kw = {"device_id": self.dev} if self.backend == "nccl" else {}
            dist.init_process_group(self.backend, rank=self.rank, world_size=self.world, **kw)

    def fence(self):
        if self.pg:
            self.dist.barrier()
        if self.gpu:
            torch.cuda.synchronize()

    def _reduce(self, v, op):
        t = torch.tensor([float(v)], dtype=torch.float64, device=self.dev)
        if self.pg:
            self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max(self, v):
        return self._reduce(v, self.dist.ReduceOp.MAX)

    def sum(self, v):
        return self._reduce(v, self.dist.ReduceOp.SUM)

    def broadcast_weights(self, make_sd):
        """rank 0 builds the synthetic checkpoint, everyone else receives it as ONE flat fp32 blob (tts_amd.parallel).
        -> (state_dict, seconds, bytes); the broadcast is outside every timed region."""
        from tts_amd import parallel

        sd = make_sd() if self.rank == 0 else None
        self.fence()
        t0 = time.perf_counter()
        src = sd
        sd = parallel.broadcast_state_dict(sd, src=0, device=self.dev, force=self.pg)
        self.fence()
        dt = time.perf_counter() - t0
        if self.pg and self.rank == 0:      # the blob came back through the collective: every tensor must be bit-identical
            bad = [k for k in src if not torch.equal(src[k].cpu(), sd[k])]
            if bad:
                raise SystemExit("bench.py: weight broadcast changed %d tensors (first: %s)" % (len(bad), bad[0]))
        return sd, self.max(dt), 4 * sum(v.numel() for v in sd.values())

    def close(self):
        if self.pg:
            self.dist.barrier()
            self.dist.destroy_process_group()


DETAILS = []    # bulky tables of the roofline passes: printed as their own {"detail": ...} lines BEFORE the workload's line


def base_line(args, ctx, metric, value, unit, elapsed_max, workload, dtype, higher=True, **cfg):
    return {"metric": metric, "value": value, "unit": unit, "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": higher, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": dict({"workload": workload, "parallelism": "replicas x%d" % ctx.world}, **cfg)}


def code_stamp():
    """sha256 over the kernel sources (tts_amd/csrc/*, include/*.h) with comments and whitespace stripped, first 16 hex
    digits: what a PMC file must have been recorded with for its figures to describe the code that is running (the GPU box
    has no .git; an edit to a comment does not make a measurement stale)."""
    import re

    h = hashlib.sha256()
    for d in ("tts_amd/csrc", "include"):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            if f.endswith((".hip", ".h", ".inc")):
                src = open(os.path.join(ROOT, d, f), "r", encoding="utf-8", errors="replace").read()
                src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
                src = re.sub(r"//[^\n]*", "", src)
                h.update(f.encode())
                h.update("".join(src.split()).encode())
    return h.hexdigest()[:16]


def load_pmc(name, key="hbm_bytes_per_launch"):
    """A figure from the committed PMC passes (profiles/<name>, written by scripts/gpu_round3.sh + pmc_round.py) — only if
    the file was recorded with the kernel sources that are running now (`code_stamp`), else None: a stale counter file
    is not a measurement of this code."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None
    return d.get(key) if d.get("code_stamp") == code_stamp() else None


def pmc_state(name):
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return "no PMC file"
    return ("rocprofv3 --pmc passes of this code (stamp %s)" % d.get("code_stamp") if d.get("code_stamp") == code_stamp()
            else "null: committed PMC file is stale (recorded with stamp %s, running %s)" % (d.get("code_stamp"), code_stamp()))


def kernel_family(name):
    """Kernel name -> the family key the per-family traffic table uses: conv k/d/mode, fused pair k/d/c, conv_post, other."""
    import re

    name = name.replace(" ", "")
    m = re.search(r"conv1d_\w+_kernel<(\d+),(\d+),(?:[^>]*,)?(\d+)>", name)
    if m:
        return "conv k%s d%s mode%s" % m.groups()
    m = re.search(r"resblock_pair_\w+_kernel<(\d+),(\d+),(\d+),", name)
    if m:
        return "pair k%s d%s c%s" % m.groups()
    if "conv_post_kernel" in name:
        return "conv_post"
    m = re.search(r"(\w+)_kernel", name)
    return m.group(1) if m else name[:24]


def pmc_passes(target_argv, match, timeout_s=200):
    """Two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE — one counter set per pass, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes) over a child process `python target_argv...`; per counter the sum over the
    dispatches whose kernel name contains one of `match` (spaces stripped) and their count.  Corrections of the same guide are
    applied by the callers: both counters are KiB; FETCH_SIZE reports half the bytes of coalesced streaming reads on gfx950 (x2).
    -> ({counter: (sum, dispatches)}, None) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="ttsamd_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT)
            r = subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable] + list(target_argv),
                               capture_output=True, text=True, timeout=timeout_s, cwd=ROOT, env=env)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc=%d)" % (ctr, r.returncode)
            per, by_kernel = {}, {}
            for row in csv.DictReader(open(files[0])):
                name = row["Kernel_Name"].replace(" ", "")
                if row["Counter_Name"] == ctr and any(m in name for m in match):
                    per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
                    fam = kernel_family(name)
                    by_kernel[fam] = by_kernel.get(fam, 0.0) + float(row["Counter_Value"])
            if not per:
                return None, "no dispatch of %s in the counter file" % (match,)
            vals[ctr] = (sum(per.values()), len(per))
            vals[ctr + "_by_family"] = by_kernel
    except Exception as e:          # the measurement is an extra: never cost the bench line
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return vals, None


def pmc_traffic_live(precision, kernel_sub, timeout_s=150):
    """HBM bytes per launch of the dominant kernel instantiation measured IN THIS RUN (pmc_passes over
    scripts/pmc_dominant_target.py: the kernel at its two headline shapes).  -> (bytes per launch or None, description)."""
    vals, why = pmc_passes([os.path.join(ROOT, "scripts", "pmc_dominant_target.py"), precision], [kernel_sub], timeout_s)
    if vals is None:
        return None, why
    fetch = vals["FETCH_SIZE"][0] / vals["FETCH_SIZE"][1] * 1024 * 2
    write = vals["WRITE_SIZE"][0] / vals["WRITE_SIZE"][1] * 1024
    return fetch + write, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over %d launches of the kernel at "
                           "its two headline shapes; fetch %.4g B (KiB x1024 x2, gfx950 correction) + write %.4g B per launch"
                           % (vals["FETCH_SIZE"][1], fetch, write))


class NodeProbe:
    """sysfs sampler (20 Hz, no subprocess) of the GPU this process runs on — picked by PCI address: the box may hold eight — and of
    the node's OTHER GPUs' load, over a timed region: the evidence for (or against) the two explanations of a slow single-sentence
    loop — our clocks, or neighbours sharing the node."""

    def __init__(self):
        import glob
        import threading

        self._glob = glob
        cards = [c for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")) if os.path.exists(os.path.join(c, "pp_dpm_sclk"))]
        mine = []
        try:
            pr = torch.cuda.get_device_properties(0)
            want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            mine = [c for c in cards if want in os.path.realpath(c)]
        except Exception:
            pass
        if not mine and len(cards) == 1:
            mine = cards
        self.card = mine[0] if mine else None
        self.others = [c for c in cards if c != self.card]
        self.rows, self._stop = [], False
        self._t = threading.Thread(target=self._run, daemon=True)

    def _cur(self, path):
        import re

        try:
            for ln in open(path).read().splitlines():
                if "*" in ln:
                    m = re.search(r"(\d+)Mhz", ln)
                    if m:
                        return float(m.group(1))
        except Exception:
            pass
        return float("nan")

    def _num(self, pattern, scale=1.0):
        try:
            f = self._glob.glob(pattern)
            return float(open(f[0]).read()) * scale if f else float("nan")
        except Exception:
            return float("nan")

    def _run(self):
        while not self._stop:
            busy = [self._num(o + "/gpu_busy_percent") for o in self.others]
            self.rows.append((self._cur(self.card + "/pp_dpm_sclk"), self._num(self.card + "/hwmon/hwmon*/power1_average", 1e-6),
                              sum(1 for b in busy if b == b and b > 20)))
            time.sleep(0.05)

    def __enter__(self):
        if self.card:
            self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self.card:
            self._t.join(timeout=2)

    def summary(self):
        if not self.rows:
            return {"gpus_on_node": len(self.others) + (1 if self.card else 0)}
        n = len(self.rows)
        try:
            vbios = open(self.card + "/vbios_version").read().strip()
        except Exception:
            vbios = None
        # (VBIOS and host kernel: the two slow-regime boxes whose configuration was recorded differ from the thirteen fast ones in
        # this pairing — one runs VBIOS ...-030A, one host kernel 6.18.50 instead of ...-020F + 6.18.51 — at identical clocks: DESIGN.md §5)
        return {"our_sclk_mhz": sum(r[0] for r in self.rows) / n, "gpus_on_node": len(self.others) + 1,
                "other_gpus_busy": sum(r[2] for r in self.rows) / n, "vbios": vbios, "host_kernel": os.uname().release}


def timer_table(res):
    return {k: {"launches": r["launches"], "avg_us": r["ms"] * 1e3 / r["launches"],
                "tflops": r["flops"] / (r["ms"] * 1e-3) / 1e12, "algorithmic_bytes_per_launch": r["bytes"] / r["launches"],
                "algorithmic_gbps": r["bytes"] / (r["ms"] * 1e-3) / 1e9,
                "frac_of_8TBps": r["bytes"] / (r["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS}
            for k, r in sorted(res.items()) if r["launches"]}


# ---------------------------------------------------------------------------------------------------------------------
# configs[1]: VITS end to end (the headline line)
# ---------------------------------------------------------------------------------------------------------------------
def synthetic_batch(batch, n_chars, seed, device):
    """128-char utterances -> 2*128+1 ids with the blank id interleaved (add_blank=True, vits_config.py:146;
    tokenizer.py:126-134): blank (id 0 here) at even positions, uniform random character ids at odd ones."""
    g = torch.Generator().manual_seed(seed)
    T = 2 * n_chars + 1
    x = torch.zeros(batch, T, dtype=torch.int64)
    x[:, 1::2] = torch.randint(1, 100, (batch, n_chars), generator=g)
    dur = (2 + (torch.arange(T) % 3)).float().repeat(batch, 1)
    return x.to(device), torch.full((batch,), T, dtype=torch.int64, device=device), dur.to(device)


def cpu_thread_candidates():
    """Thread counts of the CPU-baseline sweep: {8, 16, 32, 64} capped at the host's core count (the oracle's convs stop
    scaling — and then slow down — well before 256 threads)."""
    cores = os.cpu_count() or 1
    c = sorted({min(t, cores) for t in (8, 16, 32, 64)})
    return c, cores


def cpu_sweep(run, warm=None, reps=3):
    """The CPU's best, not an over-subscribed number: `run()` (returns the units it processed) is timed once per thread
    count after one `warm()` (default: `run`) at that count; the fastest count is then timed `reps` (>= 3) more times and
    the MEDIAN of those repetitions is the reported rate (1 warm-up + >= 3 timed reps), min / max beside it.
    -> dict(rate, threads, host_cores, table {threads: units/s}, sec (median seconds per run), reps, rate_min, rate_max)."""
    cands, cores = cpu_thread_candidates()
    table = {}
    reps = max(int(reps), 3)
    with torch.no_grad():
        for t in cands:
            torch.set_num_threads(t)
            (warm or run)()
            t0 = time.perf_counter()
            units = run()
            table[t] = units / (time.perf_counter() - t0)
        best = max(table, key=table.get)
        torch.set_num_threads(best)
        run()                                              # warm again at the winning count (the sweep left another count's pool)
        secs, units = [], 0
        for _ in range(reps):
            t0 = time.perf_counter()
            units = run()
            secs.append(time.perf_counter() - t0)
    secs.sort()
    med = secs[len(secs) // 2] if len(secs) % 2 else 0.5 * (secs[len(secs) // 2 - 1] + secs[len(secs) // 2])
    return {"rate": units / med, "threads": best, "host_cores": cores,
            "table": {str(k): float("%.4g" % v) for k, v in table.items()}, "sec": med, "reps": reps,
            "rate_min": units / secs[-1], "rate_max": units / secs[0]}


def cpu_entry(sw, unit, sample):
    """The `cpu_baseline` object of a bench line from a cpu_sweep result: value = median of the timed repetitions."""
    return {"value": sw["rate"], "unit": unit, "cores": sw["threads"], "kind": "port", "host_cores": sw["host_cores"],
            "reps": sw["reps"], "value_min": sw["rate_min"], "value_max": sw["rate_max"],
            "threads_sweep_" + unit.replace("/", "_per_"): sw["table"], "sample": sample}


def cpu_baseline_vits(sd, n_chars):
    """The CPU oracle (oracle/tts_oracle.py: torch fp32 restatement of the reference's modules, pinned to them)
    on this box's host cores, reference call pattern: one utterance at a time (synthesizer.py:384)."""
    from oracle import tts_oracle as O

    x, xl, dur = synthetic_batch(1, n_chars, 0, "cpu")
    noise_dp = torch.randn(1, 2, x.shape[1])
    xs, xls, durs = synthetic_batch(1, 16, 0, "cpu")          # short utterance: warms the thread pool at each count

    def warm():
        O.vits_inference(sd, xs, xls, {}, durations=durs.view(1, 1, -1))

    def run():
        # the oracle skips the DP when durations are injected; run it separately so no work is skipped
        O.vits_inference(sd, x, xl, {}, noise_dp=noise_dp, stop_after="prior")
        out = O.vits_inference(sd, x, xl, {}, noise_dp=noise_dp, durations=dur.view(1, 1, -1))
        return out["model_outputs"].shape[-1]

    sw = cpu_sweep(run, warm, reps=3)
    out = cpu_entry(sw, "samples/s",
                    "1 utterance of %d chars (197120 samples) per run, B=1 as the reference's sentence loop; oracle (torch fp32 "
                    "CPU ops), median of %d timed reps at the best of the thread sweep: %d threads of %d cores, %.2f "
                    "s/utterance, rtf_x=%.1f" % (n_chars, sw["reps"], sw["threads"], sw["host_cores"], sw["sec"],
                                                 sw["rate"] / SAMPLE_RATE))
    # SURVEY §8d also asks for the batched `x_lengths` mode: 4 ragged utterances in one oracle call at the same thread count
    nb = 4
    xb, xlb, durb = synthetic_batch(nb, n_chars, 1, "cpu")
    xlb = torch.tensor([x.shape[1], x.shape[1] - 20, x.shape[1] - 40, x.shape[1] - 57])
    durb = durb * (torch.arange(x.shape[1])[None, :] < xlb[:, None]).float()
    nzb = torch.randn(nb, 2, x.shape[1])

    def run_b():
        O.vits_inference(sd, xb, xlb, {}, noise_dp=nzb, stop_after="prior")
        o = O.vits_inference(sd, xb, xlb, {}, noise_dp=nzb, durations=durb.view(nb, 1, -1))
        return int(o["y_mask"].sum().item()) * 256

    with torch.no_grad():
        torch.set_num_threads(sw["threads"])
        run_b()
        secs = []
        for _ in range(3):
            t0 = time.perf_counter()
            units = run_b()
            secs.append(time.perf_counter() - t0)
    secs.sort()
    out["batched_x_lengths_mode"] = {"value": units / secs[1], "unit": "samples/s", "batch": nb, "reps": 3,
                                     "value_min": units / secs[2], "value_max": units / secs[0], "threads": sw["threads"]}
    return out


def arithmetic_check(dev):
    """Measured in the run (a second of work, after the timed region): the error of each conv arithmetic against an fp64 conv on
    the adversarial operands of tests/test_conv_gpu.py (mantissas that maximise the parts a split drops; |x| in [2^-7, 2^4),
    |w| in [2^-12, 2^-4); C = 128, k = 11, K = 1408 products per output), as max |err| / sum|w x| over all outputs — the number
    behind the `dtype` string's "fp32-class"."""
    import numpy as np
    import torch.nn.functional as F

    from tts_amd import ops

    rng = np.random.default_rng(128 + 11)
    C, K, T = 128, 11, 260

    def vals(shape, e_lo, e_hi):
        n = int(np.prod(shape))
        mant = (rng.integers(0, 128, n) << 16) | (rng.choice([0x7F, 0x80, 0x7E, 0x81, 0x0F, 0x10], n) << 8) | rng.choice([0x7F, 0x80, 0xFF, 0x01], n)
        bits = (rng.integers(0, 2, n).astype(np.uint32) << 31) | (rng.integers(e_lo, e_hi + 1, n).astype(np.uint32) << 23) | mant.astype(np.uint32)
        return torch.from_numpy(bits.view(np.float32).copy()).reshape(shape)

    x, w = vals((1, C, T), 120, 130), vals((C, C, K), 115, 122)
    want = F.conv1d(x.double(), w.double(), None, padding=K // 2)
    scale = F.conv1d(x.abs().double(), w.abs().double(), None, padding=K // 2)
    out = {"operands": "maximal-residual mantissas, C=128 k=11 (1408 products per output), against fp64"}
    was_p, was_g = ops.conv_precision(), ops.set_conv_small_grid(0)       # the large-grid kernels are the ones the step runs
    try:
        pc = ops.PackedConv(w, None, dev)
        for prec, key in (("h2", "three_fp16_products"), ("x3", "six_bf16_products"), ("f32", "fp32_input_mfma")):
            ops.set_conv_precision(prec)
            y = torch.empty(1, C, T, device=dev)
            ops.conv1d(pc, x.to(dev), y)
            out[key] = float(((y.cpu().double() - want).abs() / scale).max())
    finally:
        ops.set_conv_precision(was_p)
        ops.set_conv_small_grid(was_g)
    out["torch_fp32_cpu_conv"] = float(((F.conv1d(x, w, None, padding=K // 2).double() - want).abs() / scale).max())
    return out


def wl_vits_e2e(args, ctx):
    from tts_amd import ops, parallel
    from tts_amd import synthetic as W
    from tts_amd.vits import Vits

    dev = ctx.dev
    sd, bcast_s, bcast_bytes = ctx.broadcast_weights(lambda: W.make_vits_state({}, seed=1234))
    model = Vits({"model_args": {}})
    model.load_state_dict(sd)
    model.to(dev)

    x, xl, dur = synthetic_batch(args.batch, args.chars, seed=ctx.rank, device=dev)
    aux = {"x_lengths": xl, "durations": dur, "run_duration_predictor": True}
    lanes = parallel.Lanes(args.lanes, device=dev, priority=args.lane_priority) if args.lanes > 1 else None

    def step():
        if lanes is not None:
            return lanes.run(model.inference, x, aux)
        return model.inference(x, aux)

    # prime every lane twice (a front-end shape is captured into a hipGraph on its 2nd occurrence; allocator pools): untimed
    for _ in range(2 * (args.lanes if lanes is not None else 1)):
        step()
    for _ in range(args.warmup):
        out = step()

    def select(pc, a):
        # every launch that dispatches to the 128x128-block k=11 d=1 NORMAL instantiation:
        # the ResBlock1 k=11 d=1 convs of the 256- and 128-channel MRF stages
        if pc.kernel == 11 and pc.dilation == 1 and a.mode == 0 and ((pc.c_out + 31) // 32) % 4 == 0:
            return "dominant"
        # every other waveform-decoder / flow conv; the tiny text-side launches are left untouched
        if 2.0 * pc.c_out * pc.c_in * pc.kernel * a.t_out * a.batch < 1e9:
            return None
        return "conv c%d k%d d%d %s" % (pc.c_in, pc.kernel, pc.dilation,
                                        {0: "normal", 2: "convT-polyphase"}.get(a.mode, "mode%d" % a.mode))

    if args.serial_branches:
        model.waveform_decoder.concurrent_branches = False
    ctx.fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    ctx.fence()
    elapsed = time.perf_counter() - t0

    # Roofline pass (rank 0 only, AFTER the timed region, same inputs): in the timed region the three MRF resblock
    # branches run on three HIP streams, so an event pair around one launch also counts the kernels co-running with it.
    # Here the branches are serialised on one stream and every conv launch of the decoder / flows is bracketed by HIP
    # events on the stream it is launched on.
    timer = ops.ConvTimer(select)
    roof_steps = 2
    lanes = None          # the roofline pass runs on the default stream, one request at a time
    if ctx.rank == 0:
        was = model.waveform_decoder.concurrent_branches
        model.waveform_decoder.concurrent_branches = False
        # the timed region ran behind the model-level C handle (tts_amd.Vits.use_native); the per-launch HIP events of this pass
        # are recorded by the Python host's launch wrappers, so it takes the Python-driven path — the same kernel-level calls
        # (tests/test_native_models_gpu.py: bitwise the same outputs)
        was_native, model.use_native = model.use_native, False
        for _ in range(2):        # the default stream's front-end graph is captured on its 2nd occurrence: before timing
            step()
        torch.cuda.synchronize()
        ops.set_conv_timer(timer)
        for _ in range(roof_steps):
            step()
        torch.cuda.synchronize()
        ops.set_conv_timer(None)
        model.waveform_decoder.concurrent_branches = was
        model.use_native = was_native

    samples_per_step = int(out["y_mask"].sum().item()) * 256        # valid output samples of this rank's shard
    elapsed_max, total_samples_per_step = ctx.max(elapsed), ctx.sum(samples_per_step)
    if ctx.rank != 0:
        return None
    res = timer.results()
    dom = res.get("dominant", dict(launches=0, flops=0.0, bytes=0.0, ms=1.0))
    allc = dict(launches=0, flops=0.0, bytes=0.0, ms=0.0)
    for r in res.values():
        for k in allc:
            allc[k] += r[k]
    ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["launches"] else 0.0
    value = total_samples_per_step * args.steps / elapsed_max
    line = base_line(
        args, ctx, "audio samples/sec (LJSpeech VITS -> HiFiGAN decoder, 22.05 kHz, end-to-end inference)", value,
        "samples/s", elapsed_max,
        "configs[1]: LJSpeech VITS end-to-end, batch=%d random %d-char utterances per GPU (257 ids, 770 frames, 197120 "
        "samples each), 22.05 kHz" % (args.batch, args.chars), DTYPE[args.precision],
        utterances_per_gpu=args.batch, chars=args.chars,
        weights="random-init VitsArgs defaults (29.1 M params)", weight_broadcast_s=bcast_s, weight_broadcast_bytes=bcast_bytes,
        weight_broadcast_backend=ctx.backend if ctx.world > 1 else "none (1 rank)",
        mrf_branch_streams=1 if (args.serial_branches or args.lanes > 1 or model.use_native) else 3, request_lanes=args.lanes,
        host="model-level C handle (ttsamd_vits_encode / _decode)" if model.use_native else "Python host over the kernel-level ABI",
        fused_resblocks=bool(getattr(model.waveform_decoder, "fuse_resblocks", False)))
    line["rtf_x"] = value / SAMPLE_RATE
    line["rtf_x_per_gpu"] = value / SAMPLE_RATE / ctx.world
    pmc = "pmc_dominant_%s.json" % args.precision
    live_traffic, live_note = (None, "skipped (--no-live-pmc)")
    if ctx.world == 1 and not args.no_live_pmc:
        live_traffic, live_note = pmc_traffic_live(args.precision, DOMINANT_SUB[args.precision])
    line["roofline"] = {
        "bound": "mfma",
        "kernel": conv_kernel_name(args.precision, DOMINANT_TMPL[args.precision])
                  + " (ResBlock1 k=11 d=1 convs, 256->256 and 128->128)",
        "achieved": ach, "peak": conv_peak(args.precision), "unit": "TFLOP/s",
        "frac": ach / conv_peak(args.precision),
        # SURVEY §8(d) asks for both of its own rooflines per kernel: algorithmic bytes against 8 TB/s and algorithmic FLOP
        # against the 157.3 TF fp32 vector/fp32-MFMA peak (the split-bf16 kernels run on the bf16 pipe, so this one can exceed 1)
        "frac_of_8TBps": (dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS) if dom["launches"] else 0.0,
        "frac_of_157TF": ach / PEAK_FP32_MFMA_TFLOPS,
        "traffic": live_traffic if live_traffic is not None else load_pmc(pmc),
        "traffic_source": live_note if live_traffic is not None else (pmc_state(pmc) + "; live pass: " + live_note),
        # from the same PMC passes: fraction of kernel cycles the matrix pipe is busy and the kernel's cycle count
        "pmc_mfma_busy_frac": load_pmc(pmc, "mfma_busy_frac"), "pmc_kernel_cycles": load_pmc(pmc, "kernel_cycles"),
        "peak_note": ("algorithmic fp32 FLOP (2*c_out*c_in*k*t_out*B) / HIP-event launch time; peak = 16-bit dense MFMA "
                      "2500 TF / %d products per fp32 product; on the pipe: %.0f of 2500 TF" % (PRODUCTS[args.precision], ach * PRODUCTS[args.precision]))
                     if args.precision in PRODUCTS else "fp32-input MFMA peak (= fp32 vector peak); exact fp32 arithmetic",
        "launches_timed": dom["launches"], "avg_launch_us": dom["ms"] * 1e3 / max(dom["launches"], 1),
        "algorithmic_bytes_per_launch": dom["bytes"] / max(dom["launches"], 1),
        "all_conv_launches": {"launches": allc["launches"],
                              "tflops": allc["flops"] / (allc["ms"] * 1e-3) / 1e12 if allc["ms"] else 0.0,
                              "frac": (allc["flops"] / (allc["ms"] * 1e-3) / 1e12 if allc["ms"] else 0.0) / conv_peak(args.precision),
                              "ms_per_step": allc["ms"] / roof_steps},
        "measured": "%d steps after the timed region, MRF branch streams serialised, HIP events on the launch stream" % roof_steps,
    }
    DETAILS.append({"detail": "configs[1] per-kernel table of the roofline pass (HIP events per launch class)",
                    "per_kernel": timer_table(res)})
    try:
        if not args.no_extras:      # (profiling invocations pass --no-extras: no stray launch of the dominant instantiation in their counters)
            line["arithmetic_max_err_over_sum_abs_wx"] = _round(arithmetic_check(dev))
    except Exception as e:          # an extra: never cost the headline
        line["arithmetic_max_err_over_sum_abs_wx"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if ctx.world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_vits(sd, args.chars)
    del model
    return line


# ---------------------------------------------------------------------------------------------------------------------
# configs[0]: Glow-TTS + HiFiGAN-v2, one 64-char sentence through the mel seam, CPU oracle beside it
# ---------------------------------------------------------------------------------------------------------------------
def wl_glow_hifigan_v2(args, ctx):
    """BASELINE configs[0] (SURVEY §8d "Config 1"): B=1, 64 token ids uniform in [0,130) seed 0 (Glow add_blank=False),
    GlowTTSConfig defaults (glow_tts_config.py:101-152, 12 flow blocks), durations pinned to 4+(t mod 3) -> 318 frames,
    AudioProcessor denormalize -> vocoder normalize seam (synthesizer.py:412-429; on the device here),
    HifiganGenerator v2 (C0=128) with inference_padding=5 -> 83 968 samples.  A step = one sentence, the reference's own
    call pattern (TTS/tts/models/glow_tts.py:341-374 -> TTS/utils/synthesizer.py:412-433)."""
    from tts_amd import synthetic as W
    from tts_amd.audio import AudioProcessor, mel_renorm_device
    from tts_amd.glow_tts import GlowTTS
    from tts_amd.hifigan import HifiganGenerator

    dev = ctx.dev
    hcfg = dict(W.HIFIGAN_V2)
    gsd, t1, b1 = ctx.broadcast_weights(lambda: W.make_glow_state({}, seed=4321))
    hsd, t2, b2 = ctx.broadcast_weights(lambda: W.make_hifigan_state(hcfg, 80, seed=1234))
    glow = GlowTTS({})
    glow.load_state_dict(gsd)
    glow.to(dev)
    voc = HifiganGenerator(80, 1, hcfg["resblock_type"], hcfg["resblock_dilation_sizes"], hcfg["resblock_kernel_sizes"],
                           hcfg["upsample_kernel_sizes"], hcfg["upsample_initial_channel"], hcfg["upsample_factors"],
                           inference_padding=hcfg["inference_padding"])
    voc.load_state_dict(hsd)
    voc.to(dev)
    ap_t, ap_v = AudioProcessor(), AudioProcessor()
    T = 64
    ids = torch.randint(0, 130, (1, T), generator=torch.Generator().manual_seed(ctx.rank))
    dur = (4 + (torch.arange(T) % 3)).float().view(1, T)
    x, xl, d = ids.to(dev), torch.tensor([T], device=dev), dur.to(dev)

    from tts_amd import _lib
    from tts_amd.synthesizer import SentencePipeline

    # the Synthesizer's request path (tts_amd/synthesizer.py: Synthesizer.tts_batch -> SentencePipeline): acoustic model ->
    # seam -> vocoder, token ids in, waveform on the device out
    pipe = SentencePipeline(glow, voc, ap_t, ap_v)
    aux = {"x_lengths": xl, "durations": d}

    def step_unfused():          # the three calls one after the other (round 3's measured path; `--unfused-sentence`)
        o = glow.inference(x, aux)
        mel = mel_renorm_device(o["model_outputs"].transpose(1, 2), ap_t, ap_v)
        return voc.inference(mel)

    def step():
        if args.unfused_sentence:
            return step_unfused()
        return pipe(x, aux)[0]

    # kernel launches of ONE request, counted by the library around an eager (graph-free) run of the same launch sequence
    def eager_step():
        if args.unfused_sentence:
            glow.use_graphs = voc.use_graphs = False
            try:
                return step_unfused()
            finally:
                glow.use_graphs = voc.use_graphs = True
        return pipe(x, aux, eager=True)[0]

    import ctypes

    _lib.lib().ttsamd_launch_count.restype = ctypes.c_uint64
    eager_step()
    torch.cuda.synchronize()
    n0 = int(_lib.lib().ttsamd_launch_count())
    eager_step()
    torch.cuda.synchronize()
    launches = int(_lib.lib().ttsamd_launch_count()) - n0

    # warm-up: W steps AND at least 0.5 s of sentences — on some boxes of the pool the first ~0.1 s of small-kernel load contains
    # one 20-100 ms stall (clock / power-state change; 8 of 8 processes on one box, none on another: a 50-step timed region that
    # starts right after five warm-up steps then reads 1.85 ms instead of 1.45)
    tw = time.perf_counter()
    nw = 0
    while nw < max(args.warmup, 4) or time.perf_counter() - tw < 0.5:
        wav = step()
        nw += 1
    ctx.fence()
    probe = NodeProbe()
    with probe:
        stamps = [time.perf_counter()]
        for _ in range(args.steps):
            wav = step()
            stamps.append(time.perf_counter())      # the host returns once per sentence (it waits for the sentence's extent)
        ctx.fence()
        elapsed = time.perf_counter() - stamps[0]
    deltas = sorted((b_ - a_) * 1e3 for a_, b_ in zip(stamps, stamps[1:]))
    lat = []
    for _ in range(min(args.steps, 20)):        # per-sentence latency incl. the D2H of the waveform (what a caller waits for)
        torch.cuda.synchronize()
        t1_ = time.perf_counter()
        step().cpu()
        lat.append((time.perf_counter() - t1_) * 1e3)
    # the same sentences with TWO in flight (tts_amd.parallel.Lanes: the next sentence's encoder runs under this one's vocoder) —
    # a throughput figure next to the sequential loop above, which stays the line's `value` (the reference's call pattern)
    lanes_ms = None
    try:
        from tts_amd import parallel

        lanes = parallel.Lanes(2, device=dev, priority=-1)
        for _ in range(8):
            lanes.run(step)
        lanes.sync()
        t_l = time.perf_counter()
        for _ in range(args.steps):
            lanes.run(step)
        lanes.sync(timeout_s=60.0)
        lanes_ms = (time.perf_counter() - t_l) / args.steps * 1e3
        lanes.close([glow, voc, pipe])
    except Exception as e:          # an extra: never cost the line
        lanes_ms = "%s: %s" % (type(e).__name__, e)
    # GPU time of a sentence (first kernel start to last kernel end on the request's stream), LAST: timing events put the queue
    # into profiling mode for the rest of the process (see _run).  step time ~ GPU time: the loop is bound by the sentence's
    # kernel chain; step time >> GPU time: by the host / dispatch path — the two readings of a "slow mode" on another box.
    gpu_ms = []
    for _ in range(min(args.steps, 30)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        gpu_ms.append(e0.elapsed_time(e1))
    gpu_ms.sort()
    samples = wav.shape[-1]
    elapsed_max, total = ctx.max(elapsed), ctx.sum(samples)
    if ctx.rank != 0:
        return None
    value = total * args.steps / elapsed_max
    line = base_line(args, ctx, "audio samples/sec (Glow-TTS -> mel seam -> HiFiGAN-v2, one 64-char sentence, B=1)", value,
                     "samples/s", elapsed_max,
                     "configs[0]: tts_models/en/ljspeech/glow-tts + vocoder_models/en/ljspeech/hifigan_v2 shape, single "
                     "64-char sentence (64 ids, 318 frames, %d samples), B=1 sentence loop" % samples, DTYPE[args.precision],
                     weights="random-init GlowTTSConfig defaults + HiFiGAN-v2 (C0=128)", weight_broadcast_s=t1 + t2,
                     weight_broadcast_bytes=b1 + b2, sentence_latency_ms_p50=float(sorted(lat)[len(lat) // 2]),
                     warmup_steps_run=nw, step_ms_p50=deltas[len(deltas) // 2], step_ms_max=deltas[-1],
                     frames=int(wav.shape[-1] // 256 - 10))
    line["rtf_x"] = value / SAMPLE_RATE
    p50 = deltas[len(deltas) // 2]
    g50 = gpu_ms[len(gpu_ms) // 2]
    line["observed"] = {"step_ms_p50": p50, "step_ms_p90": deltas[int(len(deltas) * 0.9)], "step_ms_max": deltas[-1],
                        "sentence_latency_ms_p50": float(sorted(lat)[len(lat) // 2]), "gpu_ms_per_sentence_p50": g50,
                        "warmup_steps_run": nw, "two_lanes_ms_per_sentence": lanes_ms, "node": _round(probe.summary()),
                        # which regime this process ran in (1.47 ms in some processes / boxes, 1.85 in others)
                        "mode": ("kernel-chain-bound" if p50 <= 1.12 * g50 else "host/dispatch-bound") +
                                (", fast" if p50 < 1.6 else ", SLOW")}
    # 20.4 GFLOP per sentence (FlopCounter on the reference modules); a B=1 sentence is launch/latency-bound
    # on this chip, the fraction is reported for completeness
    ach = 20.4e9 * args.steps * ctx.world / elapsed_max / 1e12
    line["roofline"] = {"bound": "mfma", "kernel": "whole sentence at B=1 (%d kernel launches per request): launch / latency-bound" % launches,
                        "achieved": ach, "peak": conv_peak(args.precision), "unit": "TFLOP/s",
                        "frac": ach / conv_peak(args.precision), "traffic": None, "launches_per_request": launches}
    line["config"]["path"] = "three calls (glow.inference, seam, vocoder.inference)" if args.unfused_sentence else \
        "Synthesizer SentencePipeline: 2 graph replays + 1 host wait per sentence"
    if ctx.world == 1 and not args.no_cpu_baseline:
        from oracle import tts_oracle as O

        def cpu_step():
            o = O.glow_tts_inference(gsd, ids, torch.tensor([T]), {}, durations=dur.view(1, 1, T))
            mel = o["model_outputs"][0].numpy()                                   # [T, C]
            voc_in = ap_v.normalize(ap_t.denormalize(mel.T))                      # synthesizer.py:414-416 (numpy seam)
            return O.hifigan_inference(hsd, "", torch.from_numpy(voc_in).unsqueeze(0), hcfg).shape[-1]

        sw = cpu_sweep(cpu_step, reps=5)
        line["cpu_baseline"] = cpu_entry(sw, "samples/s",
                                         "the same sentence (%d samples) through the oracle + numpy seam, median of %d timed reps "
                                         "at the best of the thread sweep: %d threads of %d cores, %.1f ms per sentence"
                                         % (samples, sw["reps"], sw["threads"], sw["host_cores"], sw["sec"] * 1e3))
    del glow, voc
    return line


# ---------------------------------------------------------------------------------------------------------------------
# configs[2]: HiFiGAN-v1 vocoder only, 256 x 8192-frame mels
# ---------------------------------------------------------------------------------------------------------------------
def hbm_subset_key(pc, a):
    """Launches of the vocoder that are HBM-bound rather than matrix-bound (MRF3 k=3 at C=32, conv_post,
    ups[3]); fused ResBlock pairs are keyed by ops.resblock_pair itself."""
    if pc.c_out == 1:
        return "conv_post (32->1, k=7, tanh)"
    if a.mode == 2 and pc.c_in == 64:
        return "ups[3] polyphase ConvTranspose (64->32)"
    if pc.c_in <= 32 and pc.kernel == 3 and a.mode == 0:
        return "MRF3 conv k=3 d=%d (32->32)" % pc.dilation
    return None


def wl_hifigan_v1(args, ctx):
    """BASELINE configs[2]: HiFiGAN-v1 vocoder only on precomputed 80-bin mels, batch 256 x 8192 frames per GPU
    (hifigan_config.py:95-104 generator, inference_padding 5, weight-norm folded).  The layer-by-layer live set of the
    literal shape is 6 x 69 GB, so the batch runs in slabs (HifiganGenerator.inference_slabbed); mels resident in HBM."""
    from tts_amd import ops
    from tts_amd import synthetic as W
    from tts_amd.hifigan import HifiganGenerator

    dev = ctx.dev
    cfg = dict(W.HIFIGAN_V1)
    sd, bcast_s, bcast_bytes = ctx.broadcast_weights(lambda: W.make_hifigan_state(cfg, 80, seed=1234))
    m = HifiganGenerator(80, 1, cfg["resblock_type"], cfg["resblock_dilation_sizes"], cfg["resblock_kernel_sizes"],
                         cfg["upsample_kernel_sizes"], cfg["upsample_initial_channel"], cfg["upsample_factors"],
                         inference_padding=cfg["inference_padding"])
    m.load_state_dict(sd)
    m.to(dev)
    mel = torch.randn(args.items, 80, args.frames, device=dev, generator=torch.Generator(device=dev).manual_seed(ctx.rank))
    out = torch.empty((args.items, 1, (args.frames + 10) * 256), dtype=torch.float32, device=dev)
    steps = args.hifigan_steps or args.steps
    warm = args.warmup if args.hifigan_warmup is None else args.hifigan_warmup
    for _ in range(warm):
        m.inference_slabbed(mel, out)
    timer = ops.ConvTimer(lambda pc, a: "conv")
    ops.set_conv_timer(timer)
    ctx.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.inference_slabbed(mel, out)
    ctx.fence()
    elapsed = time.perf_counter() - t0
    ops.set_conv_timer(None)
    elapsed_max = ctx.max(elapsed)

    # HBM-roofline pass for the launches that are NOT matrix-bound: one slab, MRF branches serialised on one stream,
    # HIP events around each of those launches on the launch stream.
    sub = None
    if ctx.rank == 0:
        n_slab = min(args.items, 16)
        was = m.concurrent_branches
        m.concurrent_branches = False
        m.inference(mel[:n_slab])
        torch.cuda.synchronize()
        st = ops.ConvTimer(hbm_subset_key)
        ops.set_conv_timer(st)
        m.inference(mel[:n_slab])
        torch.cuda.synchronize()
        ops.set_conv_timer(None)
        m.concurrent_branches = was
        sub_raw = st.results()
        sub = timer_table(sub_raw)
        # time-weighted mean over the whole HBM subset (total algorithmic bytes / total event time): the typical launch, next to
        # the best one
        # (HBM-priced launches only: algorithmic bytes / 8 TB/s >= algorithmic FLOP / the matrix peak — the timer also sees the
        # matrix-priced fused pairs)
        pk = conv_peak(args.precision) * 1e12
        hb = [v for v in sub_raw.values() if v["launches"] and v["bytes"] / (PEAK_HBM_GBPS * 1e9) >= v["flops"] / pk]
        sb = sum(v["bytes"] for v in hb)
        sms = sum(v["ms"] for v in hb)
        sub_mean = (sb / (sms * 1e-3) / 1e9 / PEAK_HBM_GBPS) if sms > 0 else 0.0
    if ctx.rank != 0:
        return None
    r = dict(launches=0, flops=0.0, bytes=0.0, ms=0.0)          # plain convs ("conv") + fused ResBlock pairs
    for v in timer.results().values():
        for k in r:
            r[k] += v[k]
    samples = float(out.shape[0] * out.shape[2]) * ctx.world
    value = samples * steps / elapsed_max
    line = base_line(args, ctx, "audio samples/sec (HiFiGAN-v1 vocoder only, 80-bin mels -> 22.05 kHz waveform)", value,
                     "samples/s", elapsed_max,
                     "configs[2]: HiFiGAN-v1 vocoder only, batch=%d x %d-frame mels per GPU, slabbed" % (args.items, args.frames),
                     DTYPE[args.precision], weight_broadcast_s=bcast_s, weight_broadcast_bytes=bcast_bytes,
                     fused_resblocks=bool(getattr(m, "fuse_resblocks", False)))
    line["steps"], line["warmup"], line["ms_per_step"] = steps, warm, elapsed_max / steps * 1e3
    line["rtf_x"] = value / SAMPLE_RATE
    # the MRF branches run on three HIP streams here, so per-launch event times overlap; the aggregate is priced
    # on the wall clock of the timed region instead (a lower bound on the conv kernels' own rate)
    ach = r["flops"] / elapsed_max / 1e12
    # HBM traffic of one step, measured in this run: the PMC passes run the generator on ONE 4-item slab in a child process
    # (scripts/pmc_hifigan_target.py: every conv / fused-pair / conv_post dispatch counted) and the bytes per output sample
    # are scaled to the step; next to it the layer-by-layer algorithmic figure of SURVEY §8(d) (21 237 B per sample)
    traffic, traffic_note, per_sample, traffic_excess = None, "skipped (--no-live-pmc)", None, None
    if ctx.world == 1 and not args.no_live_pmc:
        pm_items, pm_frames = 4, args.frames
        vals, why = pmc_passes([os.path.join(ROOT, "scripts", "pmc_hifigan_target.py"), args.precision, str(pm_items), str(pm_frames)],
                               ["conv1d_", "resblock_pair_", "conv_post_kernel", "replicate_pad"], timeout_s=300)
        if vals is None:
            traffic_note = why
        else:
            runs = 2                                           # the target runs the slab twice (both runs counted)
            slab_samples = float(pm_items * (pm_frames + 10) * 256)
            per_sample = (vals["FETCH_SIZE"][0] * 1024 * 2 + vals["WRITE_SIZE"][0] * 1024) / runs / slab_samples
            traffic = per_sample * samples / ctx.world
            # where the bytes above the algorithmic figure come from: per kernel family, counters of the slab vs the algorithmic
            # bytes of the same launches (input + output (+ residual / accumulate) once; ConvTimer over the same slab here)
            try:
                was = m.concurrent_branches
                m.concurrent_branches = False
                ft = ops.ConvTimer(lambda pc, a: "conv_post" if pc.c_out == 1 else "conv k%d d%d mode%d" % (pc.kernel, pc.dilation, a.mode))
                ft.select_pair = lambda pc1, a: "pair k%d d%d c%d" % (a.kernel, a.dilation, a.c)
                ops.set_conv_timer(ft)
                m.inference(mel[:pm_items])
                torch.cuda.synchronize()
                ops.set_conv_timer(None)
                m.concurrent_branches = was
                alg = {k: v["bytes"] for k, v in ft.results().items() if v["launches"]}
                meas = {}
                for fam, kib in vals["FETCH_SIZE_by_family"].items():
                    meas[fam] = meas.get(fam, 0.0) + kib * 1024 * 2 / runs
                for fam, kib in vals["WRITE_SIZE_by_family"].items():
                    meas[fam] = meas.get(fam, 0.0) + kib * 1024 / runs
                tot_alg = sum(alg.values())
                rows = sorted(((meas.get(k, 0.0) - alg.get(k, 0.0), k) for k in set(meas) | set(alg)), reverse=True)
                excess = {k: {"measured_MB": meas.get(k, 0.0) / 1e6, "algorithmic_MB": alg.get(k, 0.0) / 1e6,
                              "excess_share_of_algorithmic_total": d / tot_alg} for d, k in rows}
                DETAILS.append({"detail": "configs[2] traffic by kernel family over one %d-item slab: PMC bytes (FETCH x2 + WRITE) vs the "
                                          "algorithmic bytes of the same launches" % pm_items, "families": excess})
                top = [(k, v) for k, v in excess.items()][:3]
                traffic_excess = {"measured_over_algorithmic": sum(meas.values()) / tot_alg,
                                  "top": {k: round(v["excess_share_of_algorithmic_total"], 4) for k, v in top}}
            except Exception as e:      # an attribution table must never cost the line
                traffic_excess = {"error": "%s: %s" % (type(e).__name__, e)}
            traffic_note = ("measured in this run: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE over every conv / "
                            "fused-pair / conv_post dispatch (%d per slab) of a %d-item x %d-frame slab, %.0f B per output sample, scaled "
                            "to the step's samples" % (vals["FETCH_SIZE"][1] // runs, pm_items, pm_frames, per_sample))
    line["roofline"] = {"bound": "mfma", "kernel": "all conv launches of the generator (%s)" % conv_kernel_name(args.precision, "..."),
                        "achieved": ach, "peak": conv_peak(args.precision), "unit": "TFLOP/s",
                        "frac": ach / conv_peak(args.precision),
                        "traffic": traffic, "traffic_source": traffic_note, "traffic_excess": traffic_excess,
                        "traffic_bytes_per_sample": per_sample, "algorithmic_bytes_per_sample": r["bytes"] / (samples / ctx.world * steps),
                        "algorithmic_bytes_per_sample_layer_by_layer": 21237.0,
                        "algorithmic_gbps": r["bytes"] / elapsed_max / 1e9, "launches_timed": r["launches"],
                        "measured": "algorithmic conv FLOP of the timed steps / wall time of the timed region",
                        "hbm_subset_best_frac_of_8TBps": max([v["frac_of_8TBps"] for v in sub.values()] or [0.0]),
                        "hbm_subset_mean_frac_of_8TBps": sub_mean}
    DETAILS.append({"detail": "configs[2] hbm_subset: launches bound by HBM, not the matrix pipe (SURVEY App. A): algorithmic "
                              "bytes (input + output (+ residual / accumulate) once) / HIP-event time, one 16-item slab, "
                              "branches serialised; peak 8000 GB/s (a float4 copy reaches 6290)", "launches": sub})
    if ctx.world == 1 and not args.no_cpu_baseline:
        from oracle import tts_oracle as O

        # bounded sample of the same workload: ONE item of 1024 frames (+ the 5-frame replicate padding) through the
        # oracle's restatement of HifiganGenerator.inference (hifigan_generator.py:267-282)
        cmel = torch.randn(1, 80, 1024, generator=torch.Generator().manual_seed(0))
        csd = {k: v.float().cpu() for k, v in sd.items()}

        def cpu_run():
            return O.hifigan_inference(csd, "", cmel, cfg).shape[-1]

        sw = cpu_sweep(cpu_run, warm=lambda: O.hifigan_inference(csd, "", cmel[:, :, :64], cfg), reps=3)
        line["cpu_baseline"] = cpu_entry(sw, "samples/s",
                                         "1 item x 1024 frames (264 704 samples) of the same generator through the oracle, median "
                                         "of %d timed reps at the best of the thread sweep: %d threads of %d cores, %.2f s per run"
                                         % (sw["reps"], sw["threads"], sw["host_cores"], sw["sec"]))
    del m
    return line


# ---------------------------------------------------------------------------------------------------------------------
# MAS alone, XTTS streaming (vocoder half)
# ---------------------------------------------------------------------------------------------------------------------
def wl_mas(args, ctx):
    """The `maximum_path` operator alone (BASELINE.md "MAS" row): randn [B,257,770] fp32, ragged t_x in [200,257],
    t_y in [600,770], seed 0.  HBM-bound: 12 B per band cell (value read + in-place write + path write)."""
    import numpy as np

    from tts_amd import helpers

    B, TX, TY = args.mas_batch, 257, 770
    rng = np.random.default_rng(ctx.rank)
    tx = rng.integers(200, TX + 1, B)
    ty = rng.integers(600, TY + 1, B)
    tx[0], ty[0] = TX, TY
    mask = ((np.arange(TX)[None, :, None] < tx[:, None, None]) & (np.arange(TY)[None, None, :] < ty[:, None, None]))
    mask_t = torch.from_numpy(mask.astype(np.float32)).to(ctx.dev)
    value = torch.randn(B, TX, TY, device=ctx.dev)
    cells = float(sum(int(a) * int(b) - int(a) * (int(a) - 1) for a, b in zip(tx, ty)))   # band-limited cell count
    for _ in range(max(args.warmup, 1)):
        helpers.maximum_path(value, mask_t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.fence()
    e0.record()
    for _ in range(args.steps):
        helpers.maximum_path(value, mask_t)
    e1.record()
    ctx.fence()
    ms = ctx.max(e0.elapsed_time(e1)) / args.steps
    if ctx.rank != 0:
        return None
    line = base_line(args, ctx, "monotonic alignment search, DP cells/s (maximum_path on [%d,257,770])" % B,
                     cells * ctx.world / (ms * 1e-3), "cells/s", ms * 1e-3 * args.steps,
                     "maximum_path(value, mask), B=%d, T_x<=257, T_y<=770, ragged" % B, "f32->int32")
    gbps = 12.0 * cells / (ms * 1e-3) / 1e9
    line["roofline"] = {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                        "traffic": None,
                        "note": "the DP is a serial sweep over T_y per item (770 dependent steps): latency-bound unless "
                                "the batch fills the chip"}
    if ctx.world == 1 and not args.no_cpu_baseline:
        from oracle import mas as omas

        nb = min(B, 32)
        v, mm = value[:nb].cpu().numpy(), mask[:nb].astype(np.float32)
        c32 = float(sum(int(a) * int(b) - int(a) * (int(a) - 1) for a, b in zip(tx[:nb], ty[:nb])))
        # the reference's own Cython core (oracle/_ref, compiled from core.pyx in the build container; it travels with the
        # snapshot) when present, else the C restatement — both single-threaded, as the reference ships it
        impl, kind = ("ref", "reference") if omas.ref_maximum_path_c() is not None else ("c", "port")
        omas.maximum_path(v, mm, impl)
        reps = 5
        t0 = time.time()
        for _ in range(reps):
            omas.maximum_path(v, mm, impl)
        dt = (time.time() - t0) / reps
        line["cpu_baseline"] = {"value": c32 / dt, "unit": "cells/s", "cores": 1, "kind": kind,
                                "sample": "first %d items of the same problem, %s (single thread, as the reference ships it), %d reps, "
                                          "%.1f ms per call" % (nb, "core.pyx compiled into oracle/_ref" if impl == "ref" else
                                                                "C restatement of core.pyx", reps, dt * 1e3)}
        line["observed"] = {"ms_per_call": ms, "gpu_over_cpu": (cells / (ms * 1e-3)) / (c32 / dt)}
    return line


def wl_vits_b1(args, ctx):
    """One VITS request at a time — the reference's real call pattern (TTS/utils/synthesizer.py:384: one sentence per
    `synthesis()` call): 257 ids -> 770 frames -> 197 120 samples (8.94 s of audio), VitsArgs defaults, SDP run.  A step is one
    request; `value` = p50 wall time of a request with the waveform left on the device (host returns when the GPU is done);
    also reported: with the waveform copied to the host, and requests/s with two requests in flight (tts_amd.parallel.Lanes)."""
    from tts_amd import _lib, parallel
    from tts_amd import synthetic as W
    from tts_amd.vits import Vits
    import ctypes

    dev = ctx.dev
    sd, _, _ = ctx.broadcast_weights(lambda: W.make_vits_state({}, seed=1234))
    model = Vits({"model_args": {}})
    model.load_state_dict(sd)
    model.to(dev)
    x, xl, dur = synthetic_batch(1, args.chars, seed=ctx.rank, device=dev)
    aux = {"x_lengths": xl, "durations": dur, "run_duration_predictor": True}
    _lib.lib().ttsamd_launch_count.restype = ctypes.c_uint64
    eager = dict(aux, no_graph=True)
    model.inference(x, eager)
    torch.cuda.synchronize()
    n0 = int(_lib.lib().ttsamd_launch_count())
    model.inference(x, eager)
    torch.cuda.synchronize()
    launches = int(_lib.lib().ttsamd_launch_count()) - n0
    tw, nw = time.perf_counter(), 0
    while nw < max(args.warmup, 4) or time.perf_counter() - tw < 0.5:
        model.inference(x, aux)
        nw += 1
    ctx.fence()
    dev_ms, host_ms = [], []
    t_all = time.perf_counter()
    for _ in range(args.steps):
        t0 = time.perf_counter()
        out = model.inference(x, aux)
        torch.cuda.synchronize()
        dev_ms.append((time.perf_counter() - t0) * 1e3)
    ctx.fence()
    elapsed = time.perf_counter() - t_all
    for _ in range(min(args.steps, 30)):
        t0 = time.perf_counter()
        model.inference(x, aux)["model_outputs"].cpu()
        host_ms.append((time.perf_counter() - t0) * 1e3)
    lanes = parallel.Lanes(2, device=dev, priority=-1)
    for _ in range(6):
        lanes.run(model.inference, x, aux)
    lanes.sync()
    n = max(args.steps, 40)
    t0 = time.perf_counter()
    for _ in range(n):
        lanes.run(model.inference, x, aux)
    lanes.sync(timeout_s=60.0)
    lane_ms = (time.perf_counter() - t0) / n * 1e3
    lanes.close([model])
    samples = int(out["y_mask"].sum().item()) * 256
    dev_ms.sort()
    host_ms.sort()
    if ctx.rank != 0:
        return None
    p50 = dev_ms[len(dev_ms) // 2]
    line = base_line(args, ctx, "VITS single request latency, p50 ms (257 ids -> 197 120 samples, waveform on the device)", p50, "ms",
                     ctx.max(elapsed), "VITS B=1 request (the reference's call pattern, synthesizer.py:384): 128 chars = 257 ids, "
                     "770 frames, %d samples" % samples, DTYPE[args.precision], higher=False)
    line["rtf_x"] = samples / SAMPLE_RATE / (p50 * 1e-3)
    line["observed"] = {"request_ms_p50": p50, "request_ms_p90": dev_ms[int(len(dev_ms) * 0.9)], "request_ms_min": dev_ms[0],
                        "request_ms_max": dev_ms[-1], "request_ms_p50_waveform_on_host": host_ms[len(host_ms) // 2],
                        "two_lanes_ms_per_request": lane_ms, "two_lanes_requests_per_s": 1e3 / lane_ms,
                        "launches_per_request": launches, "warmup_steps_run": nw}
    ach = 488.8e9 / (p50 * 1e-3) / 1e12            # 488.8 GFLOP per utterance
    line["roofline"] = {"bound": "mfma", "kernel": "whole request at B=1 (%d kernel launches): latency-bound on the small-grid kernels" % launches,
                        "achieved": ach, "peak": conv_peak(args.precision), "unit": "TFLOP/s", "frac": ach / conv_peak(args.precision),
                        "traffic": None, "launches_per_request": launches}
    del model
    return line


def wl_xtts_stream(args, ctx):
    """Vocoder half of BASELINE configs[4] (XTTS-v2 streaming; the GPT-2 half is out of scope): one
    sentence of 200 synthetic GPT latents [1024], stream_chunk_size=20, overlap 1024 (xtts.py:609-616 defaults) through
    the XTTS-v2 HifiDecoder (1024 -> 512 ch, ups 8,8,2,2, d-vector 512).  A "step" is one streamed sentence; the metric
    is the wall time from "chunk's last latent available" to "chunk's waveform on the host", p50 over chunks."""
    import numpy as np

    from tts_amd import synthetic as W
    from tts_amd.xtts_decoder import HifiDecoder
    from tts_amd.xtts_stream import XttsStreamer

    dev = ctx.dev
    sd, cfg = W.make_hifi_decoder_state(seed=31)
    dec = HifiDecoder()
    dec.load_state_dict(sd)
    dec.to(dev)
    gen = torch.Generator().manual_seed(ctx.rank)
    lat = torch.randn(200, 1024, generator=gen).to(dev)
    g = torch.randn(1, 512, 1, generator=gen).to(dev)

    def run(windowed):
        st = XttsStreamer(dec, stream_chunk_size=20, overlap_wav_len=1024, windowed=windowed)
        lats, n = [], 0
        it = st.stream(iter(lat), g)
        torch.cuda.synchronize()
        while True:
            t0 = time.perf_counter()
            try:
                c = next(it)
            except StopIteration:
                break
            c = c.cpu()
            lats.append((time.perf_counter() - t0) * 1e3)
            n += c.numel()
        return lats, n, st.frames_decoded

    for _ in range(max(args.warmup, 1)):
        run(True)
        run(False)
    ctx.fence()
    t0 = time.perf_counter()
    first, rest, samples = [], [], 0
    for _ in range(args.steps):
        l, n, frames_w = run(True)
        first.append(l[0])
        rest += l[1:]
        samples += n
    ctx.fence()
    wall = ctx.max(time.perf_counter() - t0)
    ref_first, ref_rest = [], []
    for _ in range(args.steps):
        l, _, frames_f = run(False)
        ref_first.append(l[0])
        ref_rest += l[1:]
    if ctx.rank != 0:
        return None
    line = base_line(args, ctx, "XTTS-v2 streaming, vocoder half: p50 first-chunk latency (20 latents -> waveform on host)",
                     float(np.median(first)), "ms", wall,
                     "configs[4] vocoder half: 200 GPT latents/sentence, stream_chunk_size=20, overlap_wav_len=1024, "
                     "HifiDecoder 1024->512ch, tail-window schedule", "f32", higher=False,
                     p50_later_chunk_ms=float(np.median(rest)), max_later_chunk_ms=float(np.max(rest)),
                     reference_schedule_p50_first_ms=float(np.median(ref_first)),
                     reference_schedule_p50_later_ms=float(np.median(ref_rest)),
                     reference_schedule_max_later_ms=float(np.max(ref_rest)),
                     frames_vocoded_window=frames_w, frames_vocoded_reference_schedule=frames_f,
                     samples_per_s_per_gpu=samples / wall)
    if ctx.world == 1 and not args.no_cpu_baseline:
        from oracle import tts_oracle as O

        torch.set_num_threads(min(16, os.cpu_count() or 1))
        cpu_sd = {k: v.float() for k, v in sd.items()}
        with torch.no_grad():
            O.hifi_decoder_forward(cpu_sd, lat[:20].cpu()[None], g.cpu(), cfg)
            t0 = time.perf_counter()
            n = 3
            for _ in range(n):
                O.hifi_decoder_forward(cpu_sd, lat[:20].cpu()[None], g.cpu(), cfg)
            dt = (time.perf_counter() - t0) / n
        line["cpu_baseline"] = {"value": dt * 1e3, "unit": "ms", "cores": torch.get_num_threads(), "kind": "port",
                                "sample": "first chunk only (20 latents) through the oracle's HifiDecoder restatement"}
    return line


def wl_launch_check(args, ctx):
    """The launcher / process-group / broadcast / reduction skeleton every workload above runs on, with a trivial timed
    body — runs without a GPU on gloo (tests/test_parallel.py drives `bench.py --gpus 2 --workload launch_check`)."""
    from tts_amd import synthetic as W

    big = ctx.gpu            # on GPUs the blob is the real one: the 116 MB VITS state_dict of the headline workload
    sd, bcast_s, bcast_bytes = ctx.broadcast_weights(
        (lambda: W.make_vits_state({}, seed=1234)) if big else (lambda: W.make_hifigan_state(dict(W.HIFIGAN_V2), 80, seed=7)))
    digest = float(sum(float(v.double().sum()) for v in sd.values()))
    same = ctx.max(digest) == -ctx.max(-digest)                    # identical weights on every rank
    units = 1000 + ctx.rank
    ctx.fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + ctx.rank))                          # rank r is slower: the MAX over ranks must win
    ctx.fence()
    elapsed = time.perf_counter() - t0
    elapsed_max, total = ctx.max(elapsed), ctx.sum(units)
    if ctx.rank != 0:
        return None
    line = base_line(args, ctx, "launch check (no kernels)", total * args.steps / elapsed_max, "units/s", elapsed_max,
                     "launcher skeleton", "none", backend=ctx.backend, process_group=bool(ctx.pg), weights_identical=bool(same),
                     weight_broadcast_s=bcast_s, weight_broadcast_bytes=bcast_bytes, units_per_step_all_ranks=total,
                     slowest_rank_floor_s=0.01 * ctx.world * args.steps)
    # the same printer as the GPU workloads (tests/test_parallel.py checks the output contract on it): a bulky detail
    # table that must NOT end up in the last line, and the two objects every headline line carries
    DETAILS.append({"detail": "launch_check filler table", "per_kernel": {"k%03d" % i: {"avg_us": float(i), "note": "x" * 64}
                                                                          for i in range(200)}})
    line["roofline"] = {"bound": "hbm", "achieved": 0.0, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": 0.0, "traffic": None,
                        "kernel": "none (launcher skeleton)"}
    line["cpu_baseline"] = {"value": total * args.steps / elapsed_max, "unit": "units/s", "cores": 1, "kind": "port",
                            "sample": "the sleep loop itself"}
    return line


WORKLOADS = {"vits_e2e": wl_vits_e2e, "glow_hifigan_v2": wl_glow_hifigan_v2, "hifigan_v1": wl_hifigan_v1, "mas": wl_mas,
             "xtts_stream": wl_xtts_stream, "vits_b1": wl_vits_b1, "launch_check": wl_launch_check}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 2 for hifigan_v1: a step is 6 s)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 5; 1 for hifigan_v1)")
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU per step")
    ap.add_argument("--chars", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="skip the in-run rocprofv3 --pmc passes behind roofline.traffic (the committed, stamped PMC file is used)")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline run only: skip the configs[0] / configs[2] lines carried under extra_workloads at N=1")
    ap.add_argument("--serial-branches", action="store_true",
                    help="run the MRF resblock branches on one stream (for rocprof: per-kernel durations are then not "
                         "inflated by co-running kernels; the roofline pass always runs this way; with two or more lanes the "
                         "generator does so by itself)")
    ap.add_argument("--precision", default=None, choices=["h2", "x3", "f32"],
                    help="conv arithmetic: h2 = two-part fp16 split, three MFMA products per fp32 product on the large-grid launches "
                         "(default), x3 = split-bf16 kernels everywhere (six products), f32 = fp32-input MFMA kernels (bitwise "
                         "fmaf chain); all fp32-class, same parity tolerances")
    ap.add_argument("--lanes", type=int, default=2,
                    help="vits_e2e: request lanes (HIP streams) per GPU used round-robin by the steps, so that the "
                         "latency-bound text front end of one batch overlaps the waveform decoder of the previous one "
                         "(tts_amd.parallel.Lanes); 1 = one stream")
    ap.add_argument("--lane-priority", type=int, default=-1, help="HIP stream priority of the request lanes (-1 high, 0 normal)")
    ap.add_argument("--workload", default="vits_e2e", choices=sorted(WORKLOADS),
                    help="vits_e2e = BASELINE configs[1] (the headline line); glow_hifigan_v2 = configs[0]; hifigan_v1 = "
                         "configs[2], vocoder only; launch_check = the multi-rank skeleton without kernels (CPU-runnable)")
    ap.add_argument("--frames", type=int, default=8192, help="hifigan_v1: mel frames per item")
    ap.add_argument("--items", type=int, default=256, help="hifigan_v1: items per GPU per step")
    ap.add_argument("--hifigan-steps", type=int, default=None, help="hifigan_v1: timed steps (default --steps; 3 as an extra)")
    ap.add_argument("--hifigan-warmup", type=int, default=None)
    ap.add_argument("--mas-batch", type=int, default=32, help="mas: items per GPU")
    ap.add_argument("--unfused-sentence", action="store_true",
                    help="glow_hifigan_v2: time the three model calls one after the other instead of the Synthesizer's fused "
                         "sentence pipeline")
    ap.add_argument("--force-pg", action="store_true",
                    help="initialise the process group and run the weight broadcast / barriers / reductions through it even "
                         "with ONE rank (exercises the RCCL path on a single GPU)")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"],
                    help="process-group backend (default: nccl = RCCL on GPUs; gloo only for launch_check on CPU)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 2 if args.workload == "hifigan_v1" else 20
    if args.warmup is None:
        args.warmup = 1 if args.workload == "hifigan_v1" else 5

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))
    if args.workload != "launch_check" and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    try:
        _run(args)
    except BaseException as e:          # every rank says which one it is before it goes down (torchrun interleaves stderr)
        if not isinstance(e, SystemExit) or e.code not in (0, None):
            import traceback

            sys.stderr.write("[bench.py rank %s/%s host %s] %s: %s\n%s" % (
                os.environ.get("RANK", "0"), os.environ.get("WORLD_SIZE", "1"), socket.gethostname(), type(e).__name__, e,
                "" if isinstance(e, SystemExit) else traceback.format_exc()))
            sys.stderr.flush()
        raise


def _run(args):
    ctx = Ctx(args)
    if args.workload != "launch_check":
        from tts_amd import ops

        if args.precision is None:
            args.precision = ops.conv_precision()
        ops.set_conv_precision(args.precision)

    line = WORKLOADS[args.workload](args, ctx)
    emit_details(ctx)
    extras = []
    if args.workload == "vits_e2e" and ctx.world == 1 and not args.no_extras:
        # The other single-GPU BASELINE configs, each measured in its OWN fresh process after the headline's timed region.
        # Not in this process: a HIP event recorded with timing
        # (the roofline passes bracket every conv launch with them) switches its hardware queue to profiling mode for the
        # rest of the process, and every later dispatch on that queue then pays a few microseconds of completion-signal
        # handling — invisible in a 76 ms step, +75 % on the 330-launch single-sentence line (measured round 3: 3.2 ms in a
        # fresh process, 5.7 ms after the headline's roofline pass; round 2's 3.0 vs 3.86 ms discrepancy was this).
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        common = ["--precision", args.precision] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
        for name, extra_args in (("configs[0] glow_hifigan_v2", ["--workload", "glow_hifigan_v2", "--steps", "200", "--warmup", "5"]),
                                 ("mas [32,257,770] (row a1)", ["--workload", "mas", "--mas-batch", "32", "--steps", "50", "--warmup", "3"]),
                                 ("configs[2] hifigan_v1", ["--workload", "hifigan_v1", "--steps", str(args.hifigan_steps or 3),
                                                            "--warmup", str(1 if args.hifigan_warmup is None else args.hifigan_warmup),
                                                            "--items", str(args.items), "--frames", str(args.frames)] +
                                  (["--no-live-pmc"] if args.no_live_pmc else [])),
                                 ("vits_b1 single request", ["--workload", "vits_b1", "--steps", "100", "--warmup", "5", "--no-cpu-baseline"]),
                                 ("configs[4] xtts_stream (vocoder half)", ["--workload", "xtts_stream", "--steps", "5", "--warmup", "2"])):
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra_args + common, capture_output=True, text=True,
                                   timeout=600, cwd=ROOT)
                rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                for ln in rows[:-1]:
                    print(ln, flush=True)                                  # the child's detail tables (incl. its full line)
                extra = json.loads(rows[-1]) if (r.returncode == 0 and rows) else {"error": "rc=%d %s" % (r.returncode, r.stderr[-400:])}
            except Exception as e:          # an extra must never cost the headline line
                extra = {"error": "%s: %s" % (type(e).__name__, e)}
            extras.append((name, extra))
    if ctx.rank == 0:
        # Output contract (round 4): every bulky object first — tables, then each line IN FULL as a {"detail": ...} object —
        # and the compact lines LAST: configs[0], configs[2], headline (< 4 KB together), so that all three sit in whatever
        # tail of stdout the driver keeps.  The headline also carries a two-number summary of the other two configs.
        print(json.dumps({"detail": "full line of %s (every field; the compact line below drops explanatory strings only)"
                                    % line.get("config", {}).get("workload", args.workload)[:40], "line": line}), flush=True)
        for name, extra in extras:
            print(json.dumps({"extra_workload": name, "line": compact_line(extra)}), flush=True)
        if extras:
            line["other_configs"] = {name.split()[0]: {k: (float("%.5g" % v) if isinstance(v, float) else v) for k, v in (
                ("ms_per_step", e.get("ms_per_step") if e.get("unit") != "ms" else None), ("value", e.get("value")), ("unit", e.get("unit")),
                ("roofline_frac", (e.get("roofline") or {}).get("frac")),
                ("cpu_baseline_value", (e.get("cpu_baseline") or {}).get("value")), ("error", e.get("error"))) if v is not None}
                for name, e in extras}
        print(headline_json(compact_line(line, limit=None)), flush=True)
    ctx.close()


def emit_details(ctx):
    if ctx.rank == 0:
        for d in DETAILS:
            print(json.dumps(d), flush=True)
    del DETAILS[:]


HEADLINE_MAX_BYTES = 2500
EXTRA_MAX_BYTES = 1100
_ROOF_KEEP = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_of_8TBps", "frac_of_157TF", "traffic", "traffic_bytes_per_sample",
              "algorithmic_bytes_per_sample",
              "pmc_mfma_busy_frac", "pmc_kernel_cycles", "launches_timed", "avg_launch_us", "algorithmic_bytes_per_launch",
              "all_conv_launches", "hbm_subset_best_frac_of_8TBps", "hbm_subset_mean_frac_of_8TBps", "traffic_excess", "traffic_source",
              "launches_per_request")
_CPU_KEEP = ("value", "unit", "cores", "kind", "host_cores", "reps", "value_min", "value_max", "batched_x_lengths_mode", "sample")


def _round(v):
    if isinstance(v, float):
        return float("%.6g" % v)
    if isinstance(v, dict):
        return {k: _round(x) for k, x in v.items()}
    return v


def compact_line(line, limit=EXTRA_MAX_BYTES):
    """A bench line without its prose: every number stays (rounded to 6 significant digits), explanatory strings are cut or
    dropped.  With a byte `limit` (the extra-workload lines) optional fields go until it fits."""
    if "error" in line and "metric" not in line:
        return line
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data", "rtf_x", "rtf_x_per_gpu", "other_configs",
                                "arithmetic_max_err_over_sum_abs_wx") if k in line}
    cfg = dict(line.get("config", {}))
    cfg.pop("weights", None)
    out["config"] = cfg
    if "observed" in line:                  # step / request percentiles and the regime the process ran in: numbers, always kept
        out["observed"] = line["observed"]
    if "roofline" in line:
        out["roofline"] = {k: line["roofline"][k] for k in _ROOF_KEEP if k in line["roofline"]}
    if "cpu_baseline" in line:
        out["cpu_baseline"] = {k: line["cpu_baseline"][k] for k in _CPU_KEEP if k in line["cpu_baseline"]}
    out = _round(out)
    if limit is None:
        return out
    out["dtype"] = str(out.get("dtype", ""))[:4].strip()
    out["metric"] = str(out["metric"])[:80]
    out["config"] = {"workload": str(cfg.get("workload", ""))[:60]}
    for k in ("p50_later_chunk_ms", "reference_schedule_p50_first_ms"):       # the streaming line's second numbers
        if k in cfg:
            out.setdefault("observed", {})[k] = _round(cfg[k])
    for path in (("cpu_baseline", "sample"), ("roofline", "traffic_source"), ("roofline", "kernel"), ("roofline", "all_conv_launches"),
                 ("cpu_baseline", "batched_x_lengths_mode"), ("roofline", "algorithmic_bytes_per_launch"), ("cpu_baseline", "host_cores"),
                 ("data",), ("scaling",), ("vs_baseline",), ("higher_is_better",)):
        if len(json.dumps(out)) <= limit:
            break
        d = out
        for k in path[:-1]:
            d = d.get(k, {})
        d.pop(path[-1], None)
    return out


def headline_json(line):
    """The last line of the run: compact by contract.  If a field pushes it over the limit, explanatory strings
    are dropped first (never a number that the contract names)."""
    out = json.dumps(line)
    for path in (("roofline", "peak_note"), ("roofline", "measured"), ("roofline", "traffic_source"), ("cpu_baseline", "sample"),
                 ("arithmetic_max_err_over_sum_abs_wx", "operands"),
                 ("cpu_baseline", "threads_sweep_samples_per_s"), ("config", "weights"), ("roofline", "all_conv_launches"),
                 ("cpu_baseline", "batched_x_lengths_mode"), ("roofline", "kernel")):
        if len(out) <= HEADLINE_MAX_BYTES:
            break
        d = line
        for k in path[:-1]:
            d = d.get(k, {}) if isinstance(d, dict) else {}
        if isinstance(d, dict):
            d.pop(path[-1], None)
        out = json.dumps(line)
    return out


if __name__ == "__main__":
    main()
