#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): audio samples/s + real-time factor of LJSpeech-shaped VITS
end-to-end inference (text ids -> waveform, 22.05 kHz) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" = one full `Vits.inference` pass (text encoder, stochastic duration predictor, prior expansion incl.
both random draws, 4 coupling flows, HiFiGAN waveform decoder) over a batch of 32 synthetic utterances of
128 characters (257 token ids with blanks).  Token ids are resident in HBM when the
timed region starts; random-init weights of the exact `VitsArgs` default architecture (no network => no
released checkpoint).  Output length is pinned with the synthetic duration pattern 2+(t mod 3) (770 frames =
197 120 samples per utterance) so the workload is identical run to run — the duration predictor still runs
inside the timed region, nothing is skipped.

N GPUs = N independent replicas (one process per GPU), each with its own 32-utterance shard (weak scaling);
the only collective is the one-time weight broadcast from rank 0 (RCCL), outside the timed region.

Rank 0 prints ONE JSON line (metric/value/unit/... + "roofline" + "cpu_baseline", see README/DESIGN.md).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SAMPLE_RATE = 22050
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_bf16 dense peak (no sparsity)
X3_PRODUCTS = 6                 # bf16 MFMA products issued per fp32 product by the split-bf16 kernels (conv_kernel_x3.h)
DTYPE = {"x3": "f32 (conv products: both fp32 operands split 3-way into bf16, 6 products on the bf16 MFMA, fp32 "
               "accumulate; fp32-class accuracy, same parity tolerances as --precision f32)",
         "f32": "f32"}


def conv_peak(precision):
    """Peak the conv kernels are priced against, in ALGORITHMIC (fp32-equivalent) TFLOP/s: the split-bf16 kernels issue
    six bf16 MFMA products per fp32 product, so their ceiling is the bf16 dense peak / 6."""
    return PEAK_BF16_MFMA_TFLOPS / X3_PRODUCTS if precision == "x3" else PEAK_FP32_MFMA_TFLOPS


def conv_kernel_name(precision, tmpl):
    return ("ttsamd::conv1d_x3_kernel<%s>" if precision == "x3" else "ttsamd::conv1d_mfma_kernel<%s>") % tmpl
PEAK_HBM_GBPS = 8000.0


def synthetic_batch(batch, n_chars, seed, device):
    """128-char utterances -> 2*128+1 ids with the blank id interleaved (add_blank=True, vits_config.py:146;
    tokenizer.py:126-134): blank (id 0 here) at even positions, uniform random character ids at odd ones."""
    g = torch.Generator().manual_seed(seed)
    T = 2 * n_chars + 1
    x = torch.zeros(batch, T, dtype=torch.int64)
    x[:, 1::2] = torch.randint(1, 100, (batch, n_chars), generator=g)
    dur = (2 + (torch.arange(T) % 3)).float().repeat(batch, 1)
    return x.to(device), torch.full((batch,), T, dtype=torch.int64, device=device), dur.to(device)


def cpu_baseline(sd, n_chars, seconds_budget=20.0):
    """The CPU oracle (oracle/tts_oracle.py: torch fp32 restatement of the reference's modules, pinned to them)
    on this box's host cores, reference call pattern: one utterance at a time (synthesizer.py:384)."""
    from oracle import tts_oracle as O

    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    x, xl, dur = synthetic_batch(1, n_chars, 0, "cpu")
    noise_dp = torch.randn(1, 2, x.shape[1])
    with torch.no_grad():
        t0 = time.time()
        out = O.vits_inference(sd, x, xl, {}, noise_dp=noise_dp, durations=dur.view(1, 1, -1))  # warm-up
        warm = time.time() - t0
        n, t_used, samples = 0, 0.0, 0
        while n < 1 or (t_used + warm < seconds_budget and n < 8):
            t0 = time.time()
            # the oracle skips the DP when durations are injected; time it separately so no work is skipped
            O.vits_inference(sd, x, xl, {}, noise_dp=noise_dp, stop_after="prior")
            out = O.vits_inference(sd, x, xl, {}, noise_dp=noise_dp, durations=dur.view(1, 1, -1))
            t_used += time.time() - t0
            samples += out["model_outputs"].shape[-1]
            n += 1
    return {"value": samples / t_used, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": "%d x 1 utterance of %d chars (%d samples each), B=1 sentence loop as the reference does; "
                      "torch fp32 CPU ops, %d threads of %d host cores; rtf_x=%.1f"
                      % (n, n_chars, out["model_outputs"].shape[-1], threads, cores, samples / t_used / SAMPLE_RATE)}


def bench_hifigan_v1(args, world, rank, dev, dist, W, ops, parallel):
    """BASELINE configs[2]: HiFiGAN-v1 vocoder only on precomputed 80-bin mels, batch 256 x 8192 frames per GPU
    (hifigan_config.py:95-104 generator, inference_padding 5, weight-norm folded).  The layer-by-layer live set of the
    literal shape is 6 x 69 GB, so the batch runs in slabs (HifiganGenerator.inference_slabbed); mels resident in HBM."""
    from tts_amd.hifigan import HifiganGenerator

    cfg = dict(W.HIFIGAN_V1)
    sd = W.make_hifigan_state(cfg, 80, seed=1234) if rank == 0 else None
    sd = parallel.broadcast_state_dict(sd, src=0, device=dev)
    m = HifiganGenerator(80, 1, cfg["resblock_type"], cfg["resblock_dilation_sizes"], cfg["resblock_kernel_sizes"],
                         cfg["upsample_kernel_sizes"], cfg["upsample_initial_channel"], cfg["upsample_factors"],
                         inference_padding=cfg["inference_padding"])
    m.load_state_dict(sd)
    m.to(dev)
    mel = torch.randn(args.items, 80, args.frames, device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
    out = torch.empty((args.items, 1, (args.frames + 10) * 256), dtype=torch.float32, device=dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        m.inference_slabbed(mel, out)
    timer = ops.ConvTimer(lambda pc, a: "conv")
    ops.set_conv_timer(timer)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.inference_slabbed(mel, out)
    fence()
    elapsed = time.perf_counter() - t0
    ops.set_conv_timer(None)
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        r = timer.results()["conv"]
        samples = float(out.shape[0] * out.shape[2]) * world
        value = samples * args.steps / float(tt.item())
        print(json.dumps({
            "metric": "audio samples/sec (HiFiGAN-v1 vocoder only, 80-bin mels -> 22.05 kHz waveform)", "value": value,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": float(tt.item()) / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE[args.precision], "data": "synthetic", "rtf_x": value / SAMPLE_RATE,
            "config": {"workload": "configs[2]: HiFiGAN-v1 vocoder only, batch=%d x %d-frame mels per GPU, slabbed"
                                   % (args.items, args.frames), "parallelism": "replicas x%d" % world},
            # the MRF branches run on three HIP streams here, so per-launch event times overlap; the aggregate is priced
            # on the wall clock of the timed region instead (a lower bound on the conv kernels' own rate)
            "roofline": {"bound": "mfma", "kernel": "all conv launches of the generator (%s)" % conv_kernel_name(args.precision, "..."),
                         "achieved": r["flops"] / float(tt.item()) / 1e12, "peak": conv_peak(args.precision),
                         "unit": "TFLOP/s", "frac": r["flops"] / float(tt.item()) / 1e12 / conv_peak(args.precision),
                         "traffic": None, "algorithmic_gbps": r["bytes"] / float(tt.item()) / 1e9,
                         "launches_timed": r["launches"],
                         "measured": "algorithmic conv FLOP of the timed steps / wall time of the timed region"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_mas(args, world, rank, dev, dist):
    """The `maximum_path` operator alone (BASELINE.md "MAS" row): randn [32,257,770] fp32, ragged t_x in [200,257],
    t_y in [600,770], seed 0.  HBM-bound: 12 B per band cell (value read + in-place write + path write)."""
    import numpy as np

    from tts_amd import helpers

    B, TX, TY = 32, 257, 770
    rng = np.random.default_rng(rank)
    tx = rng.integers(200, TX + 1, B)
    ty = rng.integers(600, TY + 1, B)
    tx[0], ty[0] = TX, TY
    mask = ((np.arange(TX)[None, :, None] < tx[:, None, None]) & (np.arange(TY)[None, None, :] < ty[:, None, None]))
    mask_t = torch.from_numpy(mask.astype(np.float32)).to(dev)
    value = torch.randn(B, TX, TY, device=dev)
    cells = float(sum(int(a) * int(b) - int(a) * (int(a) - 1) for a, b in zip(tx, ty)))   # band-limited cell count
    for _ in range(max(args.warmup, 1)):
        helpers.maximum_path(value, mask_t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if world > 1:
        dist.barrier()
    e0.record()
    for _ in range(args.steps):
        path = helpers.maximum_path(value, mask_t)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    line = {"metric": "monotonic alignment search, DP cells/s (maximum_path on [32,257,770])", "value": cells * world / (ms * 1e-3),
            "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32->int32", "data": "synthetic",
            "config": {"workload": "maximum_path(value, mask), B=32, T_x<=257, T_y<=770, ragged"},
            "roofline": {"bound": "hbm", "achieved": 12.0 * cells / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": 12.0 * cells / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, "traffic": None,
                         "note": "one workgroup per item (32 items): the DP is a serial column sweep per item, i.e. "
                                 "latency-bound, not bandwidth-bound, at this batch size"}}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import mas as omas

        v, m = value.cpu().numpy(), mask.astype(np.float32)
        omas.maximum_path(v, m, "c")
        t0 = time.time()
        n = 3
        for _ in range(n):
            omas.maximum_path(v, m, "c")
        dt = (time.time() - t0) / n
        line["cpu_baseline"] = {"value": cells / dt, "unit": "cells/s", "cores": 1, "kind": "port",
                                "sample": "same problem, C restatement of core.pyx (single thread, as the reference ships it)"}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_xtts_stream(args, world, rank, dev, dist, W):
    """Vocoder half of BASELINE configs[4] (XTTS-v2 streaming; the GPT-2 half is out of scope): one
    sentence of 200 synthetic GPT latents [1024], stream_chunk_size=20, overlap 1024 (xtts.py:609-616 defaults) through
    the XTTS-v2 HifiDecoder (1024 -> 512 ch, ups 8,8,2,2, d-vector 512).  A "step" is one streamed sentence; the metric
    is the wall time from "chunk's last latent available" to "chunk's waveform on the host", p50 over chunks."""
    import numpy as np

    from tts_amd.xtts_decoder import HifiDecoder
    from tts_amd.xtts_stream import XttsStreamer

    sd, cfg = W.make_hifi_decoder_state(seed=31)
    dec = HifiDecoder()
    dec.load_state_dict(sd)
    dec.to(dev)
    gen = torch.Generator().manual_seed(rank)
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
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first, rest, samples = [], [], 0
    for _ in range(args.steps):
        l, n, frames_w = run(True)
        first.append(l[0])
        rest += l[1:]
        samples += n
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    ref_first, ref_rest = [], []
    for _ in range(args.steps):
        l, _, frames_f = run(False)
        ref_first.append(l[0])
        ref_rest += l[1:]
    line = {"metric": "XTTS-v2 streaming, vocoder half: p50 first-chunk latency (20 latents -> waveform on host)",
            "value": float(np.median(first)), "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[4] vocoder half: 200 GPT latents/sentence, stream_chunk_size=20, "
                                   "overlap_wav_len=1024, HifiDecoder 1024->512ch, tail-window schedule",
                       "p50_later_chunk_ms": float(np.median(rest)), "max_later_chunk_ms": float(np.max(rest)),
                       "reference_schedule_p50_first_ms": float(np.median(ref_first)),
                       "reference_schedule_p50_later_ms": float(np.median(ref_rest)),
                       "reference_schedule_max_later_ms": float(np.max(ref_rest)),
                       "frames_vocoded_window": frames_w, "frames_vocoded_reference_schedule": frames_f,
                       "samples_per_s_per_gpu": samples / wall}}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import tts_oracle as O

        torch.set_num_threads(min(os.cpu_count() or 1, 64))
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
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU per step")
    ap.add_argument("--chars", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial-branches", action="store_true",
                    help="run the MRF resblock branches on one stream (for rocprof: per-kernel durations are then not "
                         "inflated by co-running kernels; the roofline pass always runs this way)")
    ap.add_argument("--precision", default=None, choices=["x3", "f32"],
                    help="conv arithmetic: x3 = split-bf16 kernels (default, fp32-class accuracy on the bf16 MFMA), "
                         "f32 = fp32-input MFMA kernels (bitwise fmaf chain)")
    ap.add_argument("--lanes", type=int, default=2,
                    help="vits_e2e: request lanes (HIP streams) per GPU used round-robin by the steps, so that the "
                         "latency-bound text front end of one batch overlaps the waveform decoder of the previous one "
                         "(tts_amd.parallel.Lanes); 1 = one stream")
    ap.add_argument("--lane-priority", type=int, default=-1, help="HIP stream priority of the request lanes (-1 high, 0 normal)")
    ap.add_argument("--workload", default="vits_e2e", choices=["vits_e2e", "hifigan_v1", "mas", "xtts_stream"],
                    help="vits_e2e = BASELINE configs[1] (the headline line); hifigan_v1 = configs[2], vocoder only")
    ap.add_argument("--frames", type=int, default=8192, help="hifigan_v1: mel frames per item")
    ap.add_argument("--items", type=int, default=256, help="hifigan_v1: items per GPU per step")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from tts_amd import synthetic as W         # seeded synthetic checkpoint (no network => no released weights)
    from tts_amd import ops, parallel
    from tts_amd.vits import Vits

    if args.precision is None:
        args.precision = ops.conv_precision()
    ops.set_conv_precision(args.precision)

    if args.workload == "hifigan_v1":
        return bench_hifigan_v1(args, world, rank, dev, dist, W, ops, parallel)
    if args.workload == "mas":
        return bench_mas(args, world, rank, dev, dist)
    if args.workload == "xtts_stream":
        return bench_xtts_stream(args, world, rank, dev, dist, W)

    # rank 0 builds the weights, everyone else receives them in one RCCL broadcast
    sd = W.make_vits_state({}, seed=1234) if rank == 0 else None
    t0 = time.time()
    sd = parallel.broadcast_state_dict(sd, src=0, device=dev)
    bcast_s = time.time() - t0
    model = Vits({"model_args": {}})
    model.load_state_dict(sd)
    model.to(dev)

    x, xl, dur = synthetic_batch(args.batch, args.chars, seed=rank, device=dev)
    aux = {"x_lengths": xl, "durations": dur, "run_duration_predictor": True}

    lanes = parallel.Lanes(args.lanes, device=dev, priority=args.lane_priority) if args.lanes > 1 else None

    def step():
        if lanes is not None:
            return lanes.run(model.inference, x, aux)
        return model.inference(x, aux)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.lanes if lanes is not None else 0):   # prime every lane (hipGraph capture, allocator pools): untimed
        step()
    for _ in range(args.warmup):
        out = step()
    # live roofline measurement: HIP events around every launch of the dominant kernel instantiation
    # (ResBlock k=11 d=1 convs of the 256/128-channel stages, conv1d_mfma_kernel<11,1,2,2,2,2,0>) and around all conv launches of the
    # waveform decoder in aggregate, on the stream they are launched on, during the timed region.
    def select(pc, a):
        # every launch that dispatches to the 128x128-block k=11 d=1 NORMAL instantiation:
        # the ResBlock1 k=11 d=1 convs of the 256- and 128-channel MRF stages (8 per step)
        if pc.kernel == 11 and pc.dilation == 1 and a.mode == 0 and ((pc.c_out + 31) // 32) % 4 == 0:
            return "dominant"
        # every other waveform-decoder / flow conv; the tiny text-side launches are left untouched
        return "other_conv" if 2.0 * pc.c_out * pc.c_in * pc.kernel * a.t_out * a.batch >= 1e9 else None

    if args.serial_branches:
        model.waveform_decoder.concurrent_branches = False
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0

    # Roofline pass (rank 0 only, AFTER the timed region, same inputs): in the timed region the three MRF resblock
    # branches run on three HIP streams, so an event pair around one launch also counts the kernels co-running with it.
    # Here the branches are serialised on one stream and every conv launch of the decoder / flows is bracketed by HIP
    # events on the stream it is launched on.
    timer = ops.ConvTimer(select)
    roof_steps = 2
    lanes = None          # the roofline pass runs on the default stream, one request at a time
    if rank == 0:
        was = model.waveform_decoder.concurrent_branches
        model.waveform_decoder.concurrent_branches = False
        step()
        torch.cuda.synchronize()
        ops.set_conv_timer(timer)
        for _ in range(roof_steps):
            step()
        torch.cuda.synchronize()
        ops.set_conv_timer(None)
        model.waveform_decoder.concurrent_branches = was

    samples_per_step = int(out["y_mask"].sum().item()) * 256        # valid output samples of this rank's shard
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    ss = torch.tensor([float(samples_per_step)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ss, op=dist.ReduceOp.SUM)
    elapsed_max, total_samples_per_step = float(tt.item()), float(ss.item())

    if rank == 0:
        res = timer.results()
        dom = res.get("dominant", dict(launches=0, flops=0.0, bytes=0.0, ms=1.0))
        allc = dict(launches=0, flops=0.0, bytes=0.0, ms=0.0)
        for r in res.values():
            for k in allc:
                allc[k] += r[k]
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["launches"] else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_dominant_%s.json" % args.precision)
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        value = total_samples_per_step * args.steps / elapsed_max
        line = {
            "metric": "audio samples/sec (LJSpeech VITS -> HiFiGAN decoder, 22.05 kHz, end-to-end inference)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE[args.precision], "data": "synthetic",
            "rtf_x": value / SAMPLE_RATE, "rtf_x_per_gpu": value / SAMPLE_RATE / world,
            "config": {"workload": "configs[1]: LJSpeech VITS end-to-end, batch=%d random %d-char utterances per GPU "
                                   "(257 ids, 770 frames, 197120 samples each), 22.05 kHz" % (args.batch, args.chars),
                       "utterances_per_gpu": args.batch, "chars": args.chars, "parallelism": "replicas x%d" % world,
                       "weights": "random-init VitsArgs defaults (29.1 M params), broadcast from rank 0 in %.3f s" % bcast_s,
                       "mrf_branch_streams": 1 if args.serial_branches else 3, "request_lanes": args.lanes},
            "roofline": {
                "bound": "mfma",
                "kernel": conv_kernel_name(args.precision, "11,1,1,4,4,1,0" if args.precision == "x3" else "11,1,2,2,2,2,0")
                          + " (ResBlock1 k=11 d=1 convs, 256->256 and 128->128)",
                "achieved": ach, "peak": conv_peak(args.precision), "unit": "TFLOP/s",
                "frac": ach / conv_peak(args.precision), "traffic": traffic,
                "peak_note": ("algorithmic fp32 FLOP (2*c_out*c_in*k*t_out*B per launch) / launch time; peak = bf16 dense "
                              "MFMA peak 2500 TF / 6 bf16 products per fp32 product; on the matrix pipe itself: %.0f of "
                              "2500 bf16 TFLOP/s" % (ach * X3_PRODUCTS)) if args.precision == "x3" else
                             "fp32-input MFMA peak (= fp32 vector peak); exact fp32 arithmetic",
                "launches_timed": dom["launches"], "avg_launch_us": dom["ms"] * 1e3 / max(dom["launches"], 1),
                "algorithmic_gbps": dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 if dom["launches"] else 0.0,
                "all_conv_launches": {"launches": allc["launches"],
                                      "tflops": allc["flops"] / (allc["ms"] * 1e-3) / 1e12 if allc["ms"] else 0.0,
                                      "algorithmic_gbps": allc["bytes"] / (allc["ms"] * 1e-3) / 1e9 if allc["ms"] else 0.0,
                                      "ms_per_step": allc["ms"] / roof_steps},
                "measured": "%d extra steps after the timed region with the MRF branch streams serialised (in the timed "
                            "region three branch streams overlap and inflate per-launch event times); HIP events on the "
                            "launch stream" % roof_steps,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, args.chars)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
