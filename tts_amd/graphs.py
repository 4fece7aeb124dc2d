"""hipGraph capture of launch-bound segments (the text encoder + duration predictor is ~170 tiny launches whose
host-side issue cost exceeds their GPU time).  A segment is captured per input shape on torch's capture stream —
every kernel of this package is launched on `torch.cuda.current_stream()`, so plain HIP stream capture records them —
and replayed as ONE graph launch.  Inputs are copied into static buffers; outputs alias static buffers and are valid
until the next replay of the same graph.

Capture is not free (two warm-up runs, a capture run, and `torch.cuda.graph` entry synchronises the device and empties
the allocator cache — which also stalls the other request lane), and real traffic brings a new text length with almost
every request.  So a shape is captured only once it has been seen `capture_after` times; until then — and whenever
capture fails — the segment runs as eager launches."""
import collections
import gc

import torch


class GraphedSegment:
    _capture = {}      # device index -> the ONE dedicated stream every segment is warmed up and captured on

    @classmethod
    def capture_stream(cls, device):
        """Warm-up and capture run on the same dedicated stream, so everything a segment creates per stream on first use
        (the HiFiGAN generator's MRF branch streams are keyed by the current stream) exists before capture begins, and is
        re-used by every later capture instead of piling up per throw-away stream."""
        from . import _lib

        idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        if idx not in cls._capture:
            cls._capture[idx] = _lib.OwnedStream(torch.device("cuda", idx))
        return cls._capture[idx].stream

    def __init__(self, fn, example_inputs, stable=()):
        from . import _lib

        self.stream = torch.cuda.current_stream()
        # the torch view does not keep a dedicated (lane) stream alive: hold its owner, so that the stream this graph
        # replays on — and release() synchronises — exists for as long as the graph does (a dropped `Lanes` object would
        # otherwise destroy it under the cache entry)
        self.owner = _lib.stream_owner(self.stream.cuda_stream)
        # `stable` inputs live at the same address on every call (another graph's static outputs, a model's per-stream
        # scratch): the capture reads them in place — no staging copy per replay (the cache keys on their addresses)
        self.stable = frozenset(stable)
        self.static_in = [t if i in self.stable else t.clone() for i, t in enumerate(example_inputs)]
        self.copied = [i for i, t in enumerate(self.static_in) if i not in self.stable and t.numel()]
        side = self.capture_stream(example_inputs[0].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up outside capture: one-time hipFuncSetAttribute calls, allocator
            for _ in range(2):
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # No garbage collection inside the capture: finalisers of cyclic garbage left by earlier requests (a dropped model's
        # graphs, streams, events) call HIP, and under global capture mode an unsafe call from this thread aborts the process
        # from inside a C++ destructor (seen once in eight full test runs: "Fatal Python error: Aborted", GC running under
        # torch.cuda.current_stream() in the middle of a capture).  torch.cuda.graph() itself no longer collects on entry.
        gc.collect()
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            # thread_local: only this thread's calls are policed while it captures — another host thread (a second serving
            # thread, a data loader) may allocate or synchronise meanwhile without invalidating the capture
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                self.static_out = fn(*self.static_in)
        finally:
            if gc_was_enabled:
                gc.enable()

    def __call__(self, *inputs):
        if len(self.copied) == 1:
            i = self.copied[0]
            self.static_in[i].copy_(inputs[i])
        elif self.copied:                      # all staging copies of the replay as ONE launch
            from . import ops

            ops.copy_into([self.static_in[i] for i in self.copied], [inputs[i] for i in self.copied])
        self.graph.replay()
        return self.static_out

    def release(self):
        """Before the graph, its private pool and its static outputs go away: the stream that replays it may still have
        consumers of those buffers queued (request lanes)."""
        try:
            self.stream.synchronize()
        except Exception:                 # the stream is gone (torn down behind our back): a device-wide wait covers it
            torch.cuda.synchronize()


class GraphCache:
    """Shape-keyed LRU cache of captured segments with an eager escape hatch (disabled, shape not yet hot, capture failed)."""

    def __init__(self, fn, max_entries=16, capture_after=2):
        self.fn, self.max_entries, self.capture_after = fn, max_entries, capture_after
        self.entries = collections.OrderedDict()    # key -> GraphedSegment, least recently used first
        self.stable_ptrs = {}                       # key -> addresses of the in-place inputs the capture reads (purge_addresses)
        self.hits = collections.OrderedDict()       # key -> times seen (not captured yet)
        self.failed = set()                         # keys whose capture raised: eager from then on
        self.enabled = True
        self.last_static = False                    # the last call returned a captured graph's static buffers (fixed addresses)
        self.stats = {"replays": 0, "eager": 0, "captures": 0, "capture_failures": 0, "evictions": 0}

    def clear(self):
        """Drop every captured graph (they hold raw pointers to the weights they were captured with: call this whenever
        the weights are re-packed)."""
        for seg in self.entries.values():
            seg.release()
        self.entries.clear()
        self.stable_ptrs.clear()
        self.hits.clear()
        self.failed.clear()

    def _drop(self, key):
        self.stable_ptrs.pop(key, None)
        self.entries.pop(key).release()

    def purge_addresses(self, ptrs):
        """Drop the graphs that read one of these device addresses in place (a StreamScratch set that is being evicted: the
        graph would otherwise sit in the cache as a dead entry — its key can never match again — pinning its private pool)."""
        ptrs = frozenset(ptrs)
        for k in [k for k, p in self.stable_ptrs.items() if p & ptrs]:
            self._drop(k)
            self.stats["evictions"] += 1

    def purge_stream(self, handle):
        """Drop the graphs captured for one stream (a request lane that is being torn down)."""
        for k in [k for k in self.entries if k[1] == handle]:
            self._drop(k)
        for k in [k for k in self.hits if k[1] == handle]:
            self.hits.pop(k)

    def __call__(self, *inputs, key=None, stable=()):
        """`key`: extra hashable state the captured launches depend on (scalars baked into the graph).
        `stable`: indices of inputs that live at a fixed address (read in place by the capture, no per-replay copy)."""
        self.last_static = False
        if not self.enabled:
            return self.fn(*inputs)
        # one capture per (shape, stream): concurrent request lanes (parallel.Lanes) replay on their own streams and must
        # not share the static input / output buffers of a graph
        # ... and per lane count: the HiFiGAN generator bakes "MRF branches on three streams / on one" (parallel.active_lanes)
        # into whatever it captures, so a graph captured inside a multi-lane request is not replayed for a lone one
        from . import parallel

        base = (key, torch.cuda.current_stream().cuda_stream, parallel.active_lanes() > 1) + \
            tuple((tuple(t.shape), t.dtype) for t in inputs)
        key = base + tuple(inputs[i].data_ptr() for i in stable)       # a capture is tied to its in-place inputs' addresses
        seg = self.entries.get(key)
        if seg is not None:
            self.entries.move_to_end(key)
            self.stats["replays"] += 1
            self.last_static = True
            return seg(*inputs)
        if base in self.failed:
            self.stats["eager"] += 1
            return self.fn(*inputs)
        n = self.hits.get(base, 0) + 1                                  # "hot" is a property of the shape, not of the addresses
        if n < self.capture_after:
            self.hits[base] = n
            self.hits.move_to_end(base)
            while len(self.hits) > 8 * self.max_entries:
                self.hits.popitem(last=False)
            self.stats["eager"] += 1
            return self.fn(*inputs)
        while len(self.entries) >= self.max_entries:
            self._drop(next(iter(self.entries)))
            self.stats["evictions"] += 1
        try:
            seg = GraphedSegment(self.fn, inputs, stable)
        except Exception:                            # capture is an optimisation: never fail the request over it
            self.failed.add(base)
            self.stats["capture_failures"] += 1
            torch.cuda.synchronize()
            self.stats["eager"] += 1
            return self.fn(*inputs)
        self.entries[key] = seg
        # (the address of the input AND of its storage: a stable input may be a slice / view of a scratch buffer)
        self.stable_ptrs[key] = frozenset(p for i in stable for p in (inputs[i].data_ptr(), inputs[i].untyped_storage().data_ptr()))
        # the shape has to earn its next capture again: after an LRU eviction (or a change of an in-place input's address) it is
        # captured on its `capture_after`-th fresh sighting, not on the very next one — a capture costs a collection, two warm-up
        # runs and a device synchronise, and diverse traffic would otherwise thrash on them
        self.hits.pop(base, None)
        self.stats["captures"] += 1
        self.last_static = True
        return seg(*inputs)


class _WeakList:
    """The GraphCaches a StreamScratch notifies, held weakly (`append`, `+=`, iteration over the live ones)."""

    def __init__(self, items=()):
        import weakref

        self._ref = weakref.ref
        self._items = [weakref.ref(i) for i in items]

    def append(self, item):
        self._items.append(self._ref(item))

    def __iadd__(self, items):
        for i in items:
            self.append(i)
        return self

    def __iter__(self):
        live = [(r, r()) for r in self._items]
        self._items = [r for r, o in live if o is not None]
        return iter([o for _, o in live if o is not None])

    def __len__(self):
        return len(list(iter(self)))


class StreamScratch:
    """Fixed device buffers of a model, one set per (current stream, key): what a request writes BEFORE replaying a graph
    (masks, durations, noise) lands at the same address every time, so the graph reads it in place (`stable` inputs of
    GraphCache) instead of through a staging copy.  Per stream because request lanes run concurrently.  LRU-bounded."""

    def __init__(self, max_entries=64, dependents=()):
        # sized above the graph caches that key on these buffers (2 x 16 + the sentence pipeline's 12): a live graph's scratch
        # set is not the first thing to go; `dependents`: GraphCaches told to drop their graphs over a set that is evicted
        self.max_entries = max_entries
        self.dependents = _WeakList(dependents)        # weak: a pipeline that is dropped must not stay referenced from the model
        self.sets = collections.OrderedDict()

    @staticmethod
    def _addresses(obj):
        if torch.is_tensor(obj):
            return [obj.data_ptr(), obj.untyped_storage().data_ptr()]
        if isinstance(obj, dict):
            obj = obj.values()
        if isinstance(obj, (str, bytes)):
            return []
        if isinstance(obj, (list, tuple)) or hasattr(obj, "__iter__"):
            out = []
            for v in obj:
                out += StreamScratch._addresses(v)
            return out
        return []

    def _evict(self, k):
        old = self.sets.pop(k)
        ptrs = self._addresses(old)
        for cache in self.dependents:
            cache.purge_addresses(ptrs)

    def get(self, key, make):
        k = (torch.cuda.current_stream().cuda_stream,) + tuple(key)
        s = self.sets.get(k)
        if s is None:
            while len(self.sets) >= self.max_entries:
                self._evict(next(iter(self.sets)))
            s = self.sets[k] = make()
        else:
            self.sets.move_to_end(k)
        return s

    def purge_stream(self, handle):
        """Drop the sets of one stream (a request lane that is being torn down), and the graphs reading them."""
        for k in [k for k in self.sets if k[0] == handle]:
            self._evict(k)

    def clear(self):
        for k in list(self.sets):
            self._evict(k)
