"""GPU parity: HIP HiFiGAN generator vs the CPU oracle (oracle/tts_oracle.py, pinned to the
reference modules).  Tolerance: 1e-4 absolute RMS (north_star) AND 1e-5 relative RMS (internal bar)."""
import os

import pytest
import torch

from oracle import tts_oracle as O
from oracle import weights as W
from tts_amd.hifigan import HifiganGenerator

pytestmark = pytest.mark.gpu


def _errs(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    rms = float((a - b).pow(2).mean().sqrt())
    return rms, rms / float(b.pow(2).mean().sqrt() + 1e-30)


def _make(cfg, in_ch, gpu, sd, **kw):
    m = HifiganGenerator(in_ch, 1, cfg["resblock_type"], cfg["resblock_dilation_sizes"], cfg["resblock_kernel_sizes"],
                         cfg["upsample_kernel_sizes"], cfg["upsample_initial_channel"], cfg["upsample_factors"],
                         inference_padding=cfg.get("inference_padding", 5), **kw)
    m.load_state_dict(sd)
    return m.to(gpu)


@pytest.mark.parametrize("variant", ["v1_c64", "v2", "rb2", "v1_full", "v1_full_long", "v1_full_batch"])
def test_hifigan_inference_matches_oracle(gpu, variant):
    torch.set_num_threads(8)
    cfg = dict(W.HIFIGAN_V1)
    T, B = 40, 2
    if variant == "v1_c64":
        cfg["upsample_initial_channel"] = 64
    elif variant == "v2":
        cfg = dict(W.HIFIGAN_V2)
    elif variant == "rb2":
        cfg.update(upsample_initial_channel=64, resblock_type="2", resblock_dilation_sizes=[[1, 3]] * 3)
    elif variant == "v1_full_long":       # full-width v1 (512 channels), several time tiles at every stage, two items
        T, B = 150, 2
        torch.set_num_threads(min(64, os.cpu_count() or 8))
    elif variant == "v1_full_batch":      # full-width v1, 8 x 640 columns at the 256-channel stage: its k = 3 / k = 7 pairs run fused
        T, B = 70, 8
        torch.set_num_threads(min(64, os.cpu_count() or 8))
    else:
        T, B = 24, 1
    sd = O.make_hifigan_state(cfg, 80, seed=5)
    x = torch.randn(B, 80, T, generator=torch.Generator().manual_seed(0))
    want = O.hifigan_inference(sd, "", x, cfg)
    m = _make(cfg, 80, gpu, sd)
    got = m.inference(x.to(gpu))
    assert got.shape == want.shape == (B, 1, (T + 10) * 256)
    rms, rel = _errs(got, want)
    assert rms < 1e-4 and rel < 1e-5, (rms, rel)
    if variant == "v1_full_batch":
        from tts_amd import ops
        if ops.conv_precision() == "h2":  # ... and differ from the two-launch form only at fp32 rounding level (it IS another kernel)
            assert m._fuse_limit(256, B * (T + 10) * 8) == 7 and m._fuse_limit(256, 4000) == 0
            m.fuse_max_kernel = {128: 7, 256: 0}
            two = m.inference(x.to(gpu))
            assert not torch.equal(two, got) and _errs(got, two)[1] < 2e-6, _errs(got, two)


def test_vits_decoder_shape_contract(gpu):
    """tests/tts_tests2/test_delightful_tts_layers.py:58-88 shape contract: 32 frames -> [1,1,8192];
    VITS decoder flavour (in=192, no pre/post weight-norm, no post bias, padding 0; vits.py:704-718)."""
    cfg = dict(W.HIFIGAN_V1, inference_padding=0)
    sd = O.make_hifigan_state(cfg, 192, seed=7, pre_wn=False, post_wn=False, post_bias=False)
    x = torch.randn(1, 192, 32, generator=torch.Generator().manual_seed(0))
    want = O.hifigan_forward(sd, "", x, cfg)
    got = _make(cfg, 192, gpu, sd, conv_pre_weight_norm=False, conv_post_weight_norm=False, conv_post_bias=False)(x.to(gpu))
    assert got.shape == (1, 1, 8192)
    rms, rel = _errs(got, want)
    assert rms < 1e-4 and rel < 1e-5, (rms, rel)


def test_xtts_hifi_decoder_matches_oracle_and_reference_golden(gpu):
    """XTTS HifiDecoder vocoder half (xtts/hifigan_decoder.py:615-701): two linear interpolations of the GPT latents,
    d-vector conditioning at conv_pre and after every upsampling layer.  Checked against the oracle and against the
    fixture produced by the real reference generator."""
    import os

    import numpy as np

    from tts_amd.xtts_decoder import HifiDecoder

    sd, cfg = W.make_hifi_decoder_state(decoder_input_dim=96, d_vector_dim=32, upsample_initial_channel=64, seed=31)
    lat = torch.randn(2, 9, 96, generator=torch.Generator().manual_seed(5))
    g = torch.randn(2, 32, 1, generator=torch.Generator().manual_seed(6))
    want = O.hifi_decoder_forward(sd, lat, g, cfg)
    dec = HifiDecoder(decoder_input_dim=96, upsample_initial_channel_decoder=64, d_vector_dim=32)
    dec.load_state_dict(sd)
    dec.cuda()
    got = dec.inference(lat.to(gpu), g.to(gpu))
    assert got.shape == want.shape == (2, 1, int(int(9 * 4) * (24000 / 22050)) * 256)
    rms, rel = _errs(got, want)
    assert rms < 1e-4 and rel < 1e-5, (rms, rel)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "xtts_hifi_decoder.npz"))["wav"]
    rms, rel = _errs(got, torch.from_numpy(gold))
    assert rms < 1e-4 and rel < 1e-5, (rms, rel)
    # interpolation alone is (near-)exact
    from tts_amd import ops

    z = torch.randn(3, 5, 37, generator=torch.Generator().manual_seed(7))
    for s in (4.0, 24000 / 22050, 0.5):
        ref = torch.nn.functional.interpolate(z, scale_factor=[s], mode="linear")
        out = ops.linear_interp(z.to(gpu), s)
        assert out.shape == ref.shape and float((out.cpu() - ref).abs().max()) < 1e-6


@pytest.mark.parametrize("length_scale", [1.0, 1.25])
def test_xtts_streaming_chunker(gpu, length_scale):
    """Vocoder half of Xtts.inference_stream (xtts.py:653-687): chunks from the tail-window streamer equal (to fp32 re-association)
    the reference schedule (re-vocode the whole prefix per chunk) on the HIP path, and both match the oracle's
    restatement run on the CPU; the window does O(n) generator work instead of O(n^2)."""
    from tts_amd.xtts_decoder import HifiDecoder
    from tts_amd.xtts_stream import XttsStreamer

    sd, cfg = W.make_hifi_decoder_state(decoder_input_dim=96, d_vector_dim=32, upsample_initial_channel=64, seed=31)
    steps = list(torch.randn(47, 96, generator=torch.Generator().manual_seed(11)))
    g = torch.randn(1, 32, 1, generator=torch.Generator().manual_seed(12))
    want = O.xtts_stream_decode(sd, steps, g, cfg, stream_chunk_size=10, overlap_wav_len=1024, length_scale=length_scale)
    dec = HifiDecoder(decoder_input_dim=96, upsample_initial_channel_decoder=64, d_vector_dim=32)
    dec.load_state_dict(sd)
    dec.cuda()
    dev_steps = [s.to(gpu) for s in steps]
    win = XttsStreamer(dec, stream_chunk_size=10, overlap_wav_len=1024, length_scale=length_scale, windowed=True)
    full = XttsStreamer(dec, stream_chunk_size=10, overlap_wav_len=1024, length_scale=length_scale, windowed=False)
    a = [c.cpu() for c in win.stream(iter(dev_steps), g.to(gpu))]
    b = [c.cpu() for c in full.stream(iter(dev_steps), g.to(gpu))]
    assert len(a) == len(b) == len(want) == 5            # 4 full chunks + the final flush with the 7 leftover
    for ca, cb, cw in zip(a, b, want):
        assert ca.shape == cb.shape == cw.shape
        # (not bitwise: the window's short launches and the prefix's long ones may take different conv tile families — the
        # small-grid kernels cut K into slices — which re-associates the fp32 sums)
        if ca.numel():
            assert _errs(ca, cb)[1] < 2e-6
        if cw.numel():
            rms, rel = _errs(ca, cw)
            assert rms < 1e-4 and rel < 1e-5, (rms, rel)
    assert win.frames_decoded < 0.6 * full.frames_decoded


def test_inference_slabbed_equals_inference(gpu):
    """BASELINE configs[2] runs through `inference_slabbed` (the batch cut into slabs that fit HBM): items are independent,
    so a 3-slab run reproduces the unslabbed call — device output, host output and preallocated `out` — up to the fp32
    re-association of a different conv tile family (the launcher picks the small-grid kernels, which cut K into slices, from
    the launch's total block count: a 1-item slab and the 7-item batch differ there); the device and the host path of the
    SAME slabbing are bit-identical."""
    cfg = dict(W.HIFIGAN_V1, upsample_initial_channel=64)
    sd = O.make_hifigan_state(cfg, 80, seed=21)
    m = _make(cfg, 80, gpu, sd)
    mel = torch.randn(7, 80, 40, generator=torch.Generator().manual_seed(4))
    want = m.inference(mel.to(gpu))
    widest = max((64 >> (i + 1)) * h for i, h in enumerate((8, 64, 128, 256)))
    per_item = 6 * 4 * widest * (40 + 10)
    got = m.inference_slabbed(mel.to(gpu), max_live_bytes=3 * per_item)          # slabs of 3, 3, 1 items
    assert _errs(got, want)[1] < 2e-6
    host = torch.empty(want.shape, dtype=torch.float32)
    m.inference_slabbed(mel, out=host, max_live_bytes=3 * per_item)              # host mels -> host waveforms
    torch.cuda.synchronize()
    assert torch.equal(host, got.cpu())


def test_inference_slabbed_full_width_against_oracle(gpu):
    """A parity point on the configs[2] path at the generator's real width (HiFiGAN-v1, 512 channels): five 700-frame mels
    (190 K samples each — every stage runs many time tiles of the large-grid kernels) through `inference_slabbed` cut into
    slabs of 2, 2, 1 items; the first item of the first slab, the item at a slab boundary and the lone item of the last slab
    against oracle runs on those items alone, and the whole call against the unslabbed one."""
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    cfg = dict(W.HIFIGAN_V1)
    sd = O.make_hifigan_state(cfg, 80, seed=23)
    m = _make(cfg, 80, gpu, sd)
    B, T = 5, 700
    mel = torch.randn(B, 80, T, generator=torch.Generator().manual_seed(6))
    widest = max((512 >> (i + 1)) * h for i, h in enumerate((8, 64, 128, 256)))
    per_item = 6 * 4 * widest * (T + 10)
    got = m.inference_slabbed(mel.to(gpu), max_live_bytes=2 * per_item + 1)
    assert got.shape == (B, 1, (T + 10) * 256)
    whole = m.inference(mel.to(gpu))
    assert _errs(got, whole)[1] < 2e-6          # tile family may differ with the batch (fp32 reassociation), nothing more
    for b in (0, 2, 4):
        want = O.hifigan_inference(sd, "", mel[b:b + 1], cfg)
        rms, rel = _errs(got[b:b + 1], want)
        assert rms < 1e-4 and rel < 1e-5, (b, rms, rel)


@pytest.mark.parametrize("host", ["python", "handle"])
def test_single_item_inference_graph_equals_eager(gpu, host):
    """`inference` on one item replays as a hipGraph per 32-frame length bucket (the item runs ragged-exact inside the padded
    tensor): same waveform as the eager launches at the true length — for several lengths sharing a bucket, across capture
    and replay, and after a weight re-pack (captured graphs are dropped with the weights they point to).  "python": the Python
    host's own graph cache; "handle": the default route through the vocoder handle (use_native), staged into its static buffers."""
    cfg = dict(W.HIFIGAN_V2)
    sd = W.make_hifigan_state(cfg, 80, seed=3)
    m = _make(cfg, 80, gpu, sd)
    m.use_native = host == "handle"
    g = torch.Generator().manual_seed(4)
    for T in (41, 64, 50, 41, 41):
        c = torch.randn(1, 80, T, generator=g).to(gpu)
        m.use_graphs = False
        want = m.inference(c)
        m.use_graphs = True
        for _ in range(3):
            got = m.inference(c)
            assert got.shape == want.shape == (1, 1, (T + 10) * 256)
            assert _errs(got, want)[1] < 2e-6
    if host == "python":
        assert m._graph.stats["captures"] >= 1 and m._graph.stats["replays"] >= 6 and len(m._graph.entries) <= 2
    else:
        assert len(m._native) == 1 and m._graph.stats["captures"] == 0
    m.load_state_dict(W.make_hifigan_state(cfg, 80, seed=5))          # re-pack: graphs of the old weights must not replay
    c = torch.randn(1, 80, 41, generator=g).to(gpu)
    m.use_graphs = False
    want = m.inference(c)
    m.use_graphs = True
    for _ in range(3):
        assert _errs(m.inference(c), want)[1] < 2e-6


def test_hifigan_with_untuned_kernel_sizes_dilations_and_upsamplers(gpu):
    """A generator no default config describes — ResBlock kernels 5 / 9 / 3 at dilations (1,2,4) (2,6,3) (3,12,1), upsampling
    (stride, kernel) = (3,7) (2,4) (4,4) (2,6) — takes the generic conv / polyphase ConvTranspose fallback wherever there is no
    tuned instantiation (ResBlock pairs stay unfused there) and matches the oracle (hifigan_generator.py:199-233 accepts any)."""
    torch.set_num_threads(8)
    cfg = dict(W.HIFIGAN_V1, upsample_initial_channel=64, resblock_kernel_sizes=[5, 9, 3],
               resblock_dilation_sizes=[[1, 2, 4], [2, 6, 3], [3, 12, 1]], upsample_factors=[3, 2, 4, 2], upsample_kernel_sizes=[7, 4, 4, 6])
    # (the same config as tests/golden/cases.py: hifigan_untuned, where the oracle is pinned to the reference module)
    sd = O.make_hifigan_state(cfg, 80, seed=29)
    x = torch.randn(2, 80, 33, generator=torch.Generator().manual_seed(2))
    want = O.hifigan_inference(sd, "", x, cfg)
    m = _make(cfg, 80, gpu, sd)
    got = m.inference(x.to(gpu))
    assert got.shape == want.shape == (2, 1, (33 + 10) * 48)
    rms, rel = _errs(got, want)
    assert rms < 1e-4 and rel < 1e-5, (rms, rel)
    # ... and the REFERENCE module's own output for this config (tests/golden/hifigan_untuned.npz, made by make_golden.py)
    import os

    import numpy as np

    ref = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), "golden", "hifigan_untuned.npz"))["wav"])
    rms, rel = _errs(got, ref)
    assert rms < 1e-4 and rel < 1e-5, (rms, rel)
    # ragged-exact rows of a batch equal their own single runs here too (k - stride is even for every stage)
    lens = torch.tensor([33, 21])
    rag = m.inference(x.to(gpu), lengths=lens.to(gpu))
    one = m.inference(x[1:2, :, :21].to(gpu))
    assert _errs(rag[1:2, :, : (21 + 10) * 48], one)[1] < 2e-6


@pytest.mark.parametrize("precision", ["h2", "x3", "f32"])
def test_hifigan_trained_like_weights_match_the_reference_golden(gpu, precision):
    """Stand-in for the released checkpoints (unreachable offline): weight-norm gains spread over 1e-2 .. 1e1 per
    channel, intermediate tensors spanning three decades inside a tile, conv weight rows mixing magnitudes 1e-1 .. 1e2, mel channels
    from 1e-5 to 1e2 (tests/golden/cases.py: trained_like_hifigan_state) — against the REFERENCE module's own output
    (tests/golden/hifigan_trained_like.npz, made by make_golden.py from TTS/vocoder/models/hifigan_generator.py) on every conv
    arithmetic: the three-product fp16 kernels (the 128-, 64- and 32-channel stages of this 2 x 130-frame batch run on them), the
    six-product bf16 kernels, the fp32-input MFMA kernels; and the unscaled twin of the same network (leaky ReLU is positively
    homogeneous: same function) as a second witness."""
    import numpy as np

    from tests.golden import cases
    from tts_amd import ops

    torch.set_num_threads(8)
    cfg = dict(cases.HIFIGAN_TRAINED_LIKE)
    sd, sx = cases.trained_like_hifigan_state(cfg, 77)
    x = torch.randn(2, 80, 130, generator=torch.Generator().manual_seed(5)) * sx.view(1, 80, 1)
    ref = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), "golden", "hifigan_trained_like.npz"))["wav"])
    was = ops.conv_precision()
    ops.set_conv_precision(precision)
    try:
        m = _make(cfg, 80, gpu, sd)
        m.use_graphs = False
        got = m.inference(x.to(gpu))
        m0 = _make(cfg, 80, gpu, O.make_hifigan_state(cfg, 80, seed=77))
        m0.use_graphs = False
        got0 = m0.inference((x / sx.view(1, 80, 1)).to(gpu))
    finally:
        ops.set_conv_precision(was)
    assert torch.isfinite(got).all() and got.shape == ref.shape
    rms, rel = _errs(got, ref)
    print("trained-like weights, %s: rms %.3e rel %.3e vs the reference; rel %.3e vs the unscaled network" % (precision, rms, rel, _errs(got, got0)[1]))
    assert rms < 1e-4 and rel < 1e-5, (precision, rms, rel)
    assert _errs(got, got0)[1] < 1e-5


@pytest.mark.parametrize("variant", ["v2_sentence", "v1_c256_batch", "v1_full_batch", "rb2_c64", "untuned"])
def test_native_vocoder_handle_equals_the_python_driven_path(gpu, variant):
    """The model-level C ABI (include/tts_amd.h: ttsamd_hifigan_{create,load,finalize,forward,destroy}; csrc/hifigan_model.hip)
    through ctypes: weights handed over in the reference's state_dict layout, folded / re-ordered / packed in C++, the launch
    sequence issued in C++ — against the oracle at the usual tolerance, and BITWISE equal to the Python-driven generator when both
    get the same folded weights (same kernels, same tiles), eager and through the handle's own hipGraph replay; ragged batches too.
    With the raw weight-norm parameters (the handle folds them itself, summing the norm in a different order than torch) the two
    agree to 3e-6."""
    from tts_amd import ops
    from tts_amd.hifigan import NativeHifigan

    torch.set_num_threads(8)
    cfg = dict(W.HIFIGAN_V1)
    B, T = 2, 40
    if variant == "v2_sentence":                 # a single sentence: grouped MRF launches, small-grid kernels
        cfg, B, T = dict(W.HIFIGAN_V2), 1, 61
    elif variant == "v1_c256_batch":             # large grids: three-product kernels on the 128- / 64- / 32-channel stages
        cfg["upsample_initial_channel"] = 256
        B, T = 3, 130
    elif variant == "v1_full_batch":             # full width: the 256-channel stage's fused k = 3 / k = 7 pairs (>= 4096 columns)
        B, T = 8, 60
        torch.set_num_threads(min(64, os.cpu_count() or 8))
    elif variant == "rb2_c64":
        cfg.update(upsample_initial_channel=64, resblock_type="2", resblock_dilation_sizes=[[1, 3]] * 3)
    else:                                        # kernel sizes / strides without tuned instantiations: generic kernels
        cfg = dict(W.HIFIGAN_V1, upsample_initial_channel=64, resblock_kernel_sizes=[5, 9, 3],
                   resblock_dilation_sizes=[[1, 2, 4], [2, 6, 3], [3, 12, 1]], upsample_factors=[3, 2, 4, 2], upsample_kernel_sizes=[7, 4, 4, 6])
    sd = O.make_hifigan_state(cfg, 80, seed=21)
    x = torch.randn(B, 80, T, generator=torch.Generator().manual_seed(22))
    want = O.hifigan_inference(sd, "", x, cfg)
    m = _make(cfg, 80, gpu, sd)
    m.use_graphs, m.concurrent_branches = False, False
    ref = m.inference(x.to(gpu))
    # (1) raw reference-layout state_dict: the handle folds the weight norm itself
    nat = NativeHifigan(m, sd)
    got = nat.forward(x.to(gpu))
    assert got.shape == want.shape == ref.shape and nat.output_samples(T) == want.shape[-1]
    rms, rel = _errs(got, want)
    assert rms < 1e-4 and rel < 1e-5, (variant, rms, rel)
    # (the handle sums the weight norm in its own order: weights differ from torch's fold in the last bit; the exact-fp32 path —
    # TTSAMD_CONV_PRECISION=f32 — passes that on undamped: 1.2e-6 there, 3-6e-7 on the split arithmetics)
    assert _errs(got, ref)[1] < 3e-6
    nat.close()
    # (2) the weights torch folded: bit for bit the Python-driven path
    folded = {}
    for k in sd:
        if k.endswith(".parametrizations.weight.original0"):
            name = k[: -len(".parametrizations.weight.original0")]
            folded[name + ".weight"] = ops.fold_weight_norm(sd, name)
        elif not k.endswith(".parametrizations.weight.original1"):
            folded[k] = sd[k]
    nat = NativeHifigan(m, folded)
    got = nat.forward(x.to(gpu))
    assert torch.equal(got, ref), float((got - ref).abs().max())
    xg = x.to(gpu)
    out = torch.empty_like(ref)
    for _ in range(3):                            # first call: eager + capture, then replays into the same buffers
        out.zero_()
        nat.forward(xg, use_graph=True, out=out)
        assert torch.equal(out, ref)
    if m.exact_hop and B > 1:                     # ragged-exact batching through the handle == through the Python host
        lens = torch.tensor(([T, max(1, T - 13), max(1, T // 2)] + [max(1, T - 5 * i) for i in range(3, B)])[:B]).to(gpu)
        assert torch.equal(nat.forward(xg, lengths=lens), m.inference(xg, lengths=lens))
    # errors come back as codes + messages, nothing crashes
    import pytest as _pt

    from tts_amd import _lib

    with _pt.raises(_lib.TtsAmdError):
        NativeHifigan(m, {k: v for k, v in folded.items() if not k.startswith("conv_post")})
    nat.close()
    nat.close()                                   # idempotent
