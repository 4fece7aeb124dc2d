"""GPU parity of the fused ResBlock1-iteration kernel (ttsamd_resblock_pair, tts_amd/csrc/resblock_kernel_x3.h):
  * BITWISE equal to the two ttsamd_conv1d launches it replaces (same split-bf16 products in the same order), over
    every (channels, kernel, dilation) instantiation, ragged masks, the MRF accumulate / average epilogue, tensors
    shorter than one tile and tile-edge lengths;
  * within the conv tolerance (1e-5 relative RMS) of torch's fp32 CPU ops for hifigan_generator.py:90-98;
  * the HiFiGAN generator gives identical waveforms with fusion on and off."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tts_oracle as O
from oracle import weights as W
from tts_amd import _lib, ops
from tts_amd.hifigan import HifiganGenerator

pytestmark = pytest.mark.gpu
SLOPE = 0.1


@pytest.fixture(autouse=True)
def same_summation_order():
    """The bitwise comparisons below need the two-launch baseline to sum in the fused kernel's order: small test tensors
    would otherwise take the K-split small-grid kernels (a different, equally valid, fp32 summation order)."""
    was, was_p = ops.set_conv_small_grid(1), ops.conv_precision()
    ops.set_conv_precision("x3")          # the bitwise claims are the six-product kernels'; the three-product ones: *_h2 tests below
    yield
    ops.set_conv_small_grid(was)
    ops.set_conv_precision(was_p)


def _pair(C, K, D, seed, gpu):
    g = torch.Generator().manual_seed(seed)
    w1 = torch.randn(C, C, K, generator=g) / np.sqrt(C * K)
    w2 = torch.randn(C, C, K, generator=g) / np.sqrt(C * K)
    b1, b2 = 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    return (w1, b1, w2, b2), ops.PackedConv(w1, b1, gpu, dilation=D), ops.PackedConv(w2, b2, gpu, dilation=1), g


def _unfused(pc1, pc2, x, mask, accum, div):
    tmp, y = torch.empty_like(x), torch.empty_like(x)
    ops.conv1d(pc1, x, tmp, in_act=ops.ACT_LRELU, in_slope=SLOPE, in_mask=mask)
    ops.conv1d(pc2, tmp, y, in_act=ops.ACT_LRELU, in_slope=SLOPE, res=x, accum=accum, out_div=div, in_mask=mask)
    return y


def _torch_ref(w, x, mask, accum, div, K, D):
    w1, b1, w2, b2 = w
    m = 1.0 if mask is None else mask[:, None]
    xt = F.conv1d(F.leaky_relu(x * m, SLOPE), w1, b1, padding=(K - 1) * D // 2, dilation=D)
    xt = F.conv1d(F.leaky_relu(xt * m, SLOPE), w2, b2, padding=(K - 1) // 2)
    y = xt + x
    if accum is not None:
        y = accum + y
    return y / div if div else y


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


ALL = [(C, K, D) for C in (32, 64, 128) for K in (3, 7, 11) for D in (1, 3, 5)] + \
      [(16, 3, 1), (16, 7, 3), (16, 11, 5), (8, 3, 5), (8, 7, 1), (8, 11, 3)]      # HiFiGAN-v2's late stages: zero-padded tile


@pytest.mark.parametrize("ckd", ALL, ids=lambda c: "c%d_k%d_d%d" % c)
def test_fused_pair_bitwise_equals_two_convs(gpu, ckd):
    C, K, D = ckd
    B, T = 2, 700 + 13 * K + D                       # several tiles, ragged last tile
    w, pc1, pc2, g = _pair(C, K, D, C + K + D, gpu)
    assert ops.resblock_pair_supported(pc1, pc2)
    x = torch.randn(B, C, T, generator=g).to(gpu)
    y = torch.full((B, C, T), float("nan"), device=gpu)
    ops.resblock_pair(pc1, pc2, x, y, slope=SLOPE)
    want = _unfused(pc1, pc2, x, None, None, 0.0)
    assert torch.equal(y, want), float((y - want).abs().max())
    assert _rel(y, _torch_ref(w, x.cpu(), None, None, 0.0, K, D)) < 1e-5


@pytest.mark.parametrize("case", [(32, 11, 5, 3, 1000, True, True, 3.0), (32, 3, 1, 2, 256, True, False, 0.0),
                                  (64, 7, 3, 3, 517, True, True, 3.0), (64, 11, 1, 1, 247, False, True, 0.0),
                                  (128, 3, 5, 2, 300, True, True, 3.0), (32, 7, 1, 4, 5, True, True, 3.0),
                                  (64, 3, 1, 1, 1, False, False, 0.0), (32, 11, 3, 1, 246, False, False, 0.0),
                                  (32, 11, 3, 1, 247, False, True, 3.0), (16, 7, 5, 3, 700, True, True, 3.0),
                                  (8, 11, 1, 2, 1000, True, True, 3.0)])
def test_fused_pair_mask_accum_div_and_edges(gpu, case):
    """Ragged length masks (both convs see x*mask / mid*mask), the MRF accumulate + true division of the block's last
    iteration, tensors shorter than the halo, and lengths that end exactly on / one past a tile edge (kBN = 256-(K-1))."""
    C, K, D, B, T, has_mask, has_acc, div = case
    w, pc1, pc2, g = _pair(C, K, D, sum(case[:5]), gpu)
    x = torch.randn(B, C, T, generator=g)
    lens = torch.tensor([max(1, T - 97 * i) for i in range(B)])
    mask = (torch.arange(T)[None, :] < lens[:, None]).float() if has_mask else None
    acc = torch.randn(B, C, T, generator=g) if has_acc else None
    dev = lambda t: None if t is None else t.to(gpu)  # noqa: E731
    y = torch.full((B, C, T), float("nan"), device=gpu)
    ops.resblock_pair(pc1, pc2, dev(x), y, slope=SLOPE, mask=dev(mask), accum=dev(acc), out_div=div)
    want = _unfused(pc1, pc2, dev(x), dev(mask), dev(acc), div)
    assert torch.equal(y, want), float((y - want).abs().max())
    assert _rel(y, _torch_ref(w, x, mask, acc, div, K, D)) < 1e-5


class _H2:
    """three-product arithmetic (precision "h2") with the large-grid kernels forced onto the test shapes: small-grid tiles off
    for the convs, variant 1 (the default-width tiles) for the fused pair"""

    def __enter__(self):
        self.was, self.was_p = ops.set_conv_small_grid(0), ops.conv_precision()
        ops.set_conv_precision("h2")

    def __exit__(self, *exc):
        ops.set_conv_small_grid(self.was)
        ops.set_conv_precision(self.was_p)


@pytest.mark.parametrize("ckd", ALL, ids=lambda c: "c%d_k%d_d%d" % c)
def test_fused_pair_h2_equals_two_convs_to_rounding(gpu, ckd):
    """The three-product fused kernel (resblock_kernel_h2.h: one activation exponent per block tile) against the two three-product
    conv launches (one exponent per 16-channel chunk tile) — fp32 rounding level, not bitwise —, against torch at the conv
    tolerance, and different in its bits from the six-product kernel (i.e. it is the kernel that ran)."""
    C, K, D = ckd
    B, T = 2, 700 + 13 * K + D
    w, pc1, pc2, g = _pair(C, K, D, C + K + D, gpu)
    x = torch.randn(B, C, T, generator=g).to(gpu)
    y6 = torch.empty_like(x)
    ops.resblock_pair(pc1, pc2, x, y6, slope=SLOPE, variant=1)
    with _H2():
        assert ops.resblock_pair_supported(pc1, pc2)
        y = torch.full((B, C, T), float("nan"), device=gpu)
        ops.resblock_pair(pc1, pc2, x, y, slope=SLOPE, variant=1)
        want = _unfused(pc1, pc2, x, None, None, 0.0)
    assert _rel(y, want) < 2e-6, _rel(y, want)
    assert _rel(y, _torch_ref(w, x.cpu(), None, None, 0.0, K, D)) < 1e-5
    assert not torch.equal(y, y6) and _rel(y, y6) < 2e-6


@pytest.mark.parametrize("case", [(32, 11, 5, 3, 1000, True, True, 3.0), (32, 3, 1, 2, 256, True, False, 0.0),
                                  (64, 7, 3, 3, 517, True, True, 3.0), (64, 11, 1, 1, 247, False, True, 0.0),
                                  (128, 3, 5, 2, 300, True, True, 3.0), (32, 7, 1, 4, 5, True, True, 3.0),
                                  (64, 3, 1, 1, 1, False, False, 0.0), (32, 11, 3, 1, 246, False, False, 0.0),
                                  (32, 11, 3, 1, 247, False, True, 3.0), (16, 7, 5, 3, 700, True, True, 3.0),
                                  (8, 11, 1, 2, 1000, True, True, 3.0)])
def test_fused_pair_h2_mask_accum_div_and_edges(gpu, case):
    """The edge cases of test_fused_pair_mask_accum_div_and_edges on the three-product kernel, plus a wide-range input (item
    magnitudes 1e-3 .. 1e2: every block takes its own exponent) against an fp64 evaluation."""
    C, K, D, B, T, has_mask, has_acc, div = case
    w, pc1, pc2, g = _pair(C, K, D, sum(case[:5]), gpu)
    x = torch.randn(B, C, T, generator=g) * (10.0 ** torch.linspace(-3, 2, B))[:, None, None]
    lens = torch.tensor([max(1, T - 97 * i) for i in range(B)])
    mask = (torch.arange(T)[None, :] < lens[:, None]).float() if has_mask else None
    acc = torch.randn(B, C, T, generator=g) if has_acc else None
    dev = lambda t: None if t is None else t.to(gpu)  # noqa: E731
    with _H2():
        y = torch.full((B, C, T), float("nan"), device=gpu)
        ops.resblock_pair(pc1, pc2, dev(x), y, slope=SLOPE, mask=dev(mask), accum=dev(acc), out_div=div, variant=1)
        want = _unfused(pc1, pc2, dev(x), dev(mask), dev(acc), div)
    assert torch.isfinite(y).all()
    assert _rel(y, want) < 2e-6, _rel(y, want)
    w64 = tuple(t.double() for t in w)
    ref = _torch_ref(w64, x.double(), None if mask is None else mask.double(), None if acc is None else acc.double(), div, K, D)
    for i in range(B):                                  # per item: the small-magnitude items keep their own relative accuracy
        assert _rel(y[i], ref[i]) < 1e-5, (i, _rel(y[i], ref[i]))


@pytest.mark.parametrize("kd", [(K, D) for K in (3, 7, 11) for D in (1, 3, 5)], ids=lambda c: "k%d_d%d" % c)
def test_fused_pair_h2_256_channels(gpu, kd):
    """The 256-channel pair exists on the three-product arithmetic only (8 waves, one block per CU): against the two three-product
    conv launches at fp32 rounding level, against torch at the conv tolerance; ragged masks + accumulate + division + a wide-range
    input (item magnitudes 1e-3 .. 1e2) against an fp64 evaluation; the six-product dispatch does not offer it."""
    K, D = kd
    C, B, T = 256, 3, 300 + 17 * K + D
    w, pc1, pc2, g = _pair(C, K, D, C + K + D, gpu)
    assert not ops.resblock_pair_supported(pc1, pc2)                   # six products (the suite's default): two launches
    x = torch.randn(B, C, T, generator=g) * (10.0 ** torch.linspace(-3, 2, B))[:, None, None]
    lens = torch.tensor([max(1, T - 97 * i) for i in range(B)])
    mask = (torch.arange(T)[None, :] < lens[:, None]).float()
    acc = torch.randn(B, C, T, generator=g)
    with _H2():
        assert ops.resblock_pair_supported(pc1, pc2)
        y0 = torch.full((B, C, T), float("nan"), device=gpu)
        ops.resblock_pair(pc1, pc2, x.to(gpu), y0, slope=SLOPE)
        want0 = _unfused(pc1, pc2, x.to(gpu), None, None, 0.0)
        y = torch.full((B, C, T), float("nan"), device=gpu)
        ops.resblock_pair(pc1, pc2, x.to(gpu), y, slope=SLOPE, mask=mask.to(gpu), accum=acc.to(gpu), out_div=3.0)
        want = _unfused(pc1, pc2, x.to(gpu), mask.to(gpu), acc.to(gpu), 3.0)
    assert torch.isfinite(y0).all() and torch.isfinite(y).all()
    assert _rel(y0, want0) < 2e-6 and _rel(y, want) < 2e-6, (_rel(y0, want0), _rel(y, want))
    w64 = tuple(t.double() for t in w)
    ref0 = _torch_ref(w64, x.double(), None, None, 0.0, K, D)
    ref = _torch_ref(w64, x.double(), mask.double(), acc.double(), 3.0, K, D)
    for i in range(B):                                  # per item: the small-magnitude items keep their own relative accuracy
        assert _rel(y0[i], ref0[i]) < 1e-5 and _rel(y[i], ref[i]) < 1e-5, (i, _rel(y0[i], ref0[i]), _rel(y[i], ref[i]))


@pytest.mark.parametrize("ck", [(8, 3), (16, 7), (8, 11)])
def test_fused_pair_h2_rows_beyond_the_tensor_read_as_zero(gpu, ck):
    """8- and 16-channel pairs run on a padded tile: the rows beyond the tensor's own channels must come back as zeros through the
    buffer range check, not as the bytes that follow the tensor (here: NaN and 1e30) — see the conv test of the same name."""
    C, K = ck
    B, T = 1, 900
    w, pc1, pc2, g = _pair(C, K, 3, C + K, gpu)
    x = torch.randn(B, C, T, generator=g)
    n = B * C * T
    big = torch.full((n + 40 * T + 4096,), float("nan"), device=gpu)
    big[n + 3::2] = 1e30
    xg = big[:n].view(B, C, T)
    xg.copy_(x.to(gpu))
    with _H2():
        y = torch.full((B, C, T), float("nan"), device=gpu)
        ops.resblock_pair(pc1, pc2, xg, y, slope=SLOPE)
    assert torch.isfinite(y).all() and _rel(y, _torch_ref(w, x, None, None, 0.0, K, 3)) < 1e-5


@pytest.mark.parametrize("case", [(32, 11, 5, 3, 60000), (64, 3, 1, 2, 120000), (64, 7, 3, 5, 30011), (128, 3, 3, 4, 20000)])
def test_fused_pair_h2_many_tiles_default_dispatch(gpu, case):
    """Large tensors take the three-product kernel by default (no variant): many more (item, tile) pairs than resident blocks,
    ragged masks + accumulate + division, both tiles of the 64-channel pair."""
    C, K, D, B, T = case
    w, pc1, pc2, g = _pair(C, K, D, sum(case), gpu)
    x = torch.randn(B, C, T, generator=g).to(gpu)
    lens = torch.tensor([T - 1234 * i for i in range(B)])
    mask = (torch.arange(T)[None, :] < lens[:, None]).float().to(gpu)
    acc = torch.randn(B, C, T, generator=g).to(gpu)
    y6 = torch.empty_like(x)
    ops.resblock_pair(pc1, pc2, x, y6, slope=SLOPE, mask=mask, accum=acc, out_div=3.0)
    with _H2():
        y = torch.full((B, C, T), float("nan"), device=gpu)
        ops.resblock_pair(pc1, pc2, x, y, slope=SLOPE, mask=mask, accum=acc, out_div=3.0)
        y2 = torch.full((B, C, T), float("nan"), device=gpu)
        ops.resblock_pair(pc1, pc2, x, y2, slope=SLOPE, mask=mask, accum=acc, out_div=3.0, variant=1)
    assert not torch.equal(y, y6) and _rel(y, y6) < 2e-6
    assert _rel(y2, y6) < 2e-6
    assert _rel(y, _torch_ref(w, x.cpu(), mask.cpu(), acc.cpu(), 3.0, K, D)) < 1e-5


@pytest.mark.parametrize("case", [(32, 11, 5, 3, 60000), (64, 3, 1, 2, 120000), (64, 7, 3, 5, 30011), (128, 11, 1, 4, 20000)])
def test_fused_pair_persistent_blocks_many_tiles(gpu, case):
    """Many more (item, tile) pairs than resident blocks, ragged masks + accumulate + division: still bitwise the two-launch
    result, on the default and the alternative tile."""
    C, K, D, B, T = case
    w, pc1, pc2, g = _pair(C, K, D, sum(case), gpu)
    x = torch.randn(B, C, T, generator=g).to(gpu)
    lens = torch.tensor([T - 1234 * i for i in range(B)])
    mask = (torch.arange(T)[None, :] < lens[:, None]).float().to(gpu)
    acc = torch.randn(B, C, T, generator=g).to(gpu)
    y = torch.full((B, C, T), float("nan"), device=gpu)
    ops.resblock_pair(pc1, pc2, x, y, slope=SLOPE, mask=mask, accum=acc, out_div=3.0)
    want = _unfused(pc1, pc2, x, mask, acc, 3.0)
    assert torch.equal(y, want), float((y - want).abs().max())
    y2 = torch.full((B, C, T), float("nan"), device=gpu)
    ops.resblock_pair(pc1, pc2, x, y2, slope=SLOPE, mask=mask, accum=acc, out_div=3.0, variant=1)
    assert torch.equal(y2, want)


def test_fused_pair_alternative_tile_and_limits(gpu):
    w, pc1, pc2, g = _pair(64, 11, 5, 9, gpu)
    x = torch.randn(2, 64, 900, generator=g).to(gpu)
    ya, yb = torch.empty_like(x), torch.empty_like(x)
    ops.resblock_pair(pc1, pc2, x, ya, slope=SLOPE)
    ops.resblock_pair(pc1, pc2, x, yb, slope=SLOPE, variant=1)         # 8-wave / 256-column tile
    assert torch.equal(ya, yb)
    with pytest.raises(_lib.TtsAmdError):
        ops.resblock_pair(pc1, pc2, x, x, slope=SLOPE)                 # in place is refused (tiles read x's halo)
    _, p1, p2, _ = _pair(16, 3, 1, 1, gpu)
    assert ops.resblock_pair_supported(p1, p2)                         # C = 16 / 8 (HiFiGAN-v2 tail stages): padded 32-row tile
    _, p1, p2, _ = _pair(48, 3, 1, 1, gpu)
    assert not ops.resblock_pair_supported(p1, p2)                     # no instantiation: the unfused path
    with pytest.raises(_lib.TtsAmdError):
        ops.resblock_pair(p1, p2, x[:, :48].contiguous(), ya[:, :48].contiguous(), slope=SLOPE)
    was = ops.conv_precision()
    ops.set_conv_precision("f32")
    try:
        assert not ops.resblock_pair_supported(pc1, pc2)               # exact-fp32 path keeps the two-launch form
    finally:
        ops.set_conv_precision(was)


@pytest.mark.parametrize("ragged", [False, True])
def test_hifigan_fused_equals_unfused(gpu, ragged):
    """Whole generator (v1, C0 = 256: stages of 128, 64, 32, 16 channels), fusion on every supported stage vs off:
    identical waveform bits; and against the oracle at the usual tolerance."""
    cfg = dict(W.HIFIGAN_V1, upsample_initial_channel=256)
    sd = O.make_hifigan_state(cfg, 80, seed=5)
    m = HifiganGenerator(80, 1, cfg["resblock_type"], cfg["resblock_dilation_sizes"], cfg["resblock_kernel_sizes"],
                         cfg["upsample_kernel_sizes"], cfg["upsample_initial_channel"], cfg["upsample_factors"],
                         inference_padding=cfg["inference_padding"])
    m.load_state_dict(sd)
    m.to(gpu)
    mel = torch.randn(3, 80, 23, generator=torch.Generator().manual_seed(6))
    lengths = torch.tensor([23, 17, 9]) if ragged else None
    m.fuse_resblocks, m.fuse_channels, m.fuse_max_kernel = True, (16, 32, 64, 128), {}
    fused = m.inference(mel.to(gpu), lengths=lengths)
    m.fuse_resblocks = False
    plain = m.inference(mel.to(gpu), lengths=lengths)
    assert torch.equal(fused, plain)
    if not ragged:
        want = O.hifigan_inference(sd, "", mel, cfg)
        a, b = fused.double().cpu(), want.double()
        rms = float((a - b).pow(2).mean().sqrt())
        assert rms < 1e-4 and rms / float(b.pow(2).mean().sqrt()) < 1e-5


@pytest.mark.parametrize("case", [(64, 1, 1, 2624, False), (32, 3, 2, 5000, True), (16, 5, 1, 41984, False), (8, 1, 3, 700, True),
                                  (32, 5, 1, 37, False)])
def test_grouped_branches_bitwise_equal_the_single_pairs(gpu, case):
    """ttsamd_resblock_group: the k = 3 / 7 / 11 branches of one MRF stage in ONE launch (blockIdx.y = branch) write exactly the
    bits of three ttsamd_resblock_pair launches — ragged masks, tensors shorter than a tile, two-branch groups (an absent slot),
    different inputs per branch — and ttsamd_sum_div averages them in the reference's order (hifigan_generator.py:255-261)."""
    C, D, B, T, masked = case
    g = torch.Generator().manual_seed(C + D + T)
    pairs = []
    for K in (11, 3, 7):                                   # any order: the slot follows the kernel size
        _, pc1, pc2, _ = _pair(C, K, D, 100 + K, gpu)
        pairs.append((pc1, pc2))
    xs = [torch.randn(B, C, T, generator=g).to(gpu) for _ in pairs]
    lens = torch.tensor([T, max(1, T - 29), max(1, T // 3)][:B])
    mask = (torch.arange(T)[None, :] < lens[:, None]).float().to(gpu) if masked else None
    assert ops.resblock_group_supported(pairs, B, C, T)
    want = [ops.resblock_pair(pc1, pc2, x, torch.empty_like(x), slope=SLOPE, mask=mask) for (pc1, pc2), x in zip(pairs, xs)]
    got = [torch.full_like(x, float("nan")) for x in xs]
    ops.resblock_group(pairs, xs, got, slope=SLOPE, mask=mask)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    two = [torch.full_like(x, float("nan")) for x in xs[:2]]
    ops.resblock_group(pairs[:2], xs[:2], two, slope=SLOPE, mask=mask)
    assert torch.equal(two[0], want[0]) and torch.equal(two[1], want[1])
    if T % 4 == 0:
        y = torch.empty_like(xs[0])
        ops.sum_div(got, y, 3.0)
        w = [t.cpu() for t in want]          # true division on the CPU (torch's GPU division by a scalar multiplies by 1 / div)
        assert torch.equal(y.cpu(), ((w[0] + w[1]) + w[2]) / 3.0)
        ops.sum_div(got[:2], y, 2.0)
        assert torch.equal(y.cpu(), (w[0] + w[1]) / 2.0)
    # not a small-grid shape / a kernel size without a slot: refused, the caller launches the pairs one by one
    assert not ops.resblock_group_supported(pairs, 32, C, 200000)
    _, q1, q2, _ = _pair(C, 3, D, 7, gpu)
    assert not ops.resblock_group_supported(pairs[:2] + [(q1, q2)], B, C, T)


@pytest.mark.parametrize("ragged", [False, True])
def test_hifigan_grouped_stages_equal_branch_streams(gpu, ragged):
    """HiFiGAN-v2 on a single sentence: every MRF stage as grouped launches (3 per stage + the average) vs the nine launches on
    branch streams with the accumulate chained through the last convs — identical waveform bits, also through graph replay."""
    cfg = dict(W.HIFIGAN_V2)
    sd = O.make_hifigan_state(cfg, 80, seed=15)
    m = HifiganGenerator(80, 1, cfg["resblock_type"], cfg["resblock_dilation_sizes"], cfg["resblock_kernel_sizes"],
                         cfg["upsample_kernel_sizes"], cfg["upsample_initial_channel"], cfg["upsample_factors"],
                         inference_padding=cfg["inference_padding"])
    m.load_state_dict(sd)
    m.to(gpu)
    B = 2 if ragged else 1
    mel = torch.randn(B, 80, 61, generator=torch.Generator().manual_seed(16))
    lengths = torch.tensor([61, 40]).to(gpu) if ragged else None
    m.use_graphs = False
    assert m.group_branches
    grouped = m.inference(mel.to(gpu), lengths=lengths)
    m.group_branches = False
    plain = m.inference(mel.to(gpu), lengths=lengths)
    assert torch.equal(grouped, plain)
    m.group_branches, m.use_graphs = True, True
    for _ in range(3):
        assert torch.equal(m.inference(mel.to(gpu), lengths=lengths), plain)
