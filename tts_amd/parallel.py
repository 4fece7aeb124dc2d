"""Multi-GPU: utterances are independent (the reference itself loops sentences, synthesizer.py:384), so
the path shards as replicas — one process per GPU, a contiguous slab of utterances per rank, NO data-path
collective.  The only exchange is a one-time broadcast of the weights from rank 0 (RCCL over xGMI on GPUs,
gloo in the CPU tests): the whole reference-layout state_dict travels as ONE flat fp32 blob (~116 MB for
VITS) instead of hundreds of small broadcasts — a single large transfer is what a point-to-point xGMI
ring/tree wants.  SURVEY.md §8(e).
"""
import threading

import torch
import torch.distributed as dist


def flatten_state_dict(sd):
    """-> (blob fp32 [N], manifest [(name, shape, dtype_str, offset, numel)]).  Non-float tensors are
    carried as fp32 values (exact for the small integer buffers a checkpoint may hold)."""
    manifest, parts, off = [], [], 0
    for name in sorted(sd):
        t = sd[name]
        n = t.numel()
        manifest.append((name, tuple(t.shape), str(t.dtype).replace("torch.", ""), off, n))
        parts.append(t.detach().reshape(-1).to(torch.float32).cpu())
        off += n
    blob = torch.cat(parts) if parts else torch.zeros(0)
    return blob, manifest


def unflatten_state_dict(blob, manifest):
    sd = {}
    for name, shape, dtype, off, n in manifest:
        sd[name] = blob[off:off + n].reshape(shape).to(getattr(torch, dtype)).cpu()
    return sd


def broadcast_state_dict(sd, src=0, device=None, force=False):
    """Rank `src` passes its state_dict, the others pass None; every rank returns an identical CPU copy.
    One object broadcast (the manifest, a few KB) + one tensor broadcast (the blob).
    force=True runs the collectives even in a group of ONE rank (the RCCL communicator, the broadcast kernel and the
    blob's round trip through the device are then exercised on a single GPU: tests/test_rccl_gpu.py)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return sd
    rank = dist.get_rank()
    if rank == src:
        blob, manifest = flatten_state_dict(sd)
        meta = [manifest]
    else:
        blob, meta = None, [None]
    dist.broadcast_object_list(meta, src=src)
    manifest = meta[0]
    total = sum(m[4] for m in manifest)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    if rank == src:
        buf = blob.to(dev)
    else:
        buf = torch.empty(total, dtype=torch.float32, device=dev)
    dist.broadcast(buf, src=src)
    return unflatten_state_dict(buf.cpu(), manifest)


def shard_range(n_items, rank=None, world_size=None):
    """Contiguous slab [lo, hi) of `n_items` utterances owned by `rank` (sizes differ by at most one)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_length(lengths, world_size):
    """Length-balanced partition: sort utterances by length (longest first) and deal them round-robin, so
    every rank gets the same count (+-1) and a near-equal total number of frames.
    -> list of index lists, one per rank."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return [order[r::world_size] for r in range(world_size)]


_TLS = threading.local()      # .lanes: lanes of the Lanes object whose run() is issuing this THREAD's current request


def active_lanes():
    """How many request lanes the request being issued shares the GPU with (0 outside `Lanes.run`).  The HiFiGAN generator
    reads it: with two or more requests in flight the lanes already supply the concurrency and its MRF branches run on ONE
    stream (measured round 3, two lanes, per request: B=1 3.47 vs 4.41 ms, B=4 10.2 vs 11.6, B=16 38.2 vs 41.2, B=32 75.4 vs
    77.3 — serial vs three branch streams); a lone request keeps its three branch streams (B=1 4.50 vs 4.72 ms)."""
    return getattr(_TLS, "lanes", 0)


class Lanes:
    """Request lanes inside one GPU process: N HIP streams used round-robin, one request (batch) per lane at a time.
    A request's text front end is a few milliseconds of tiny, latency-bound launches with a host sync in the middle
    (`y_lengths.max()`), its waveform decoder 80+ ms of chip-filling convs; on ONE stream the chip idles through every
    front end.  With two lanes the front end (and flows) of request i+1 run while the decoder of request i occupies the
    matrix pipes — launches are asynchronous and the mid-request host sync only waits for its own lane's stream, so a
    single host thread drives both.  Results are ordered by `sync()` (or by the caller waiting on the lane's event).

        lanes = Lanes(2)
        outs = [lanes.run(model.inference, x_i, aux_i) for ...]     # returns immediately after issuing
        lanes.sync()
    """

    def __init__(self, n=2, device=None, priority=-1):
        # priority -1 (high), measured round 3 with two lanes: B=32 76.5 vs 77.9 ms/step, B=1 3.52 vs 4.05 ms/request against
        # priority 0.  (Only while a lane's request stays on its own stream, as it does inside a lane — see active_lanes():
        # with normal-priority MRF branch streams next to a high-priority lane every stage pays cross-priority event waits,
        # B=1 5.1 ms/request.)
        # High-priority lane streams: a request's own launches (text front end, flows, up-sampling convs) then win free CU
        # slots over the other lane's resblock convs, which run on the generator's normal-priority branch streams —
        # otherwise every one of the front end's ~170 dependent launches queues behind a chip-filling decoder kernel.
        from . import _lib

        if int(n) <= 1:
            priority = 0       # a lone lane keeps its MRF branch streams (normal priority): same priority, no cross-priority waits
        self._owned = [_lib.OwnedStream(device, priority) for _ in range(max(1, int(n)))]    # dedicated, never pool-aliased
        self.streams = [o.stream for o in self._owned]
        self._next = 0
        self._outs = [None] * len(self.streams)   # keep each lane's last result alive until the lane is reused

    def run(self, fn, *args, **kwargs):
        i = self._next
        self._next = (i + 1) % len(self.streams)
        st = self.streams[i]
        st.wait_stream(torch.cuda.current_stream())        # inputs produced on the caller's stream
        was, _TLS.lanes = getattr(_TLS, "lanes", 0), len(self.streams)
        try:
            with torch.cuda.stream(st):
                out = fn(*args, **kwargs)
        finally:
            _TLS.lanes = was
        self._outs[i] = out
        return out

    def close(self, models=()):
        """Tear the lanes down: wait for them, then drop what `models` (Vits / GlowTTS / HifiganGenerator objects) keep per
        lane stream — captured graphs and MRF branch-stream sets — so that nothing stays keyed by a stream that is about to
        be destroyed.  (Graphs hold a reference to their stream's owner, so forgetting this leaks streams, it does not crash.)"""
        for st in self.streams:
            st.synchronize()
        for o in self._owned:
            for m in models:
                # a Synthesizer brings its models and its SentencePipeline (whose captured tail is keyed by the lane's stream too)
                holders = [m, getattr(m, "waveform_decoder", None), getattr(m, "model_g", None), getattr(m, "pipeline", None)]
                for sub in (getattr(m, "tts_model", None), getattr(m, "vocoder_model", None)):
                    holders += [sub, getattr(sub, "waveform_decoder", None), getattr(sub, "model_g", None)]
                for holder in holders:
                    if holder is None:
                        continue
                    for name in ("_front", "_tail", "_graph", "_scratch"):      # graphs first, then the scratch sets they read
                        cache = getattr(holder, name, None)
                        if cache is not None and hasattr(cache, "purge_stream"):
                            cache.purge_stream(o.handle)
                    if hasattr(holder, "release_streams"):
                        holder.release_streams(o.handle)
        self._outs = [None] * len(self.streams)
        self.streams, self._owned = [], []

    def sync(self, timeout_s=None):
        """Wait for every lane.  With `timeout_s` the wait polls the lanes' streams and raises TtsAmdError when the deadline
        passes instead of blocking for ever on a stalled stream — a serving loop can then drop the request and rebuild
        its lanes (`close(models)` first) (a blocking hipStreamSynchronize cannot be interrupted)."""
        if timeout_s is None:
            for st in self.streams:
                st.synchronize()
            return
        import time

        from . import _lib

        deadline = time.monotonic() + float(timeout_s)
        for i, st in enumerate(self.streams):
            while not st.query():
                if time.monotonic() > deadline:
                    raise _lib.TtsAmdError("request lane %d did not finish within %.1f s (stalled stream?)" % (i, timeout_s))
                time.sleep(0.0002)
