"""GPU parity: fused MFMA Conv1d / polyphase ConvTranspose1d vs torch fp32 CPU ops (floating point:
tolerance = 1e-5 relative RMS, i.e. fp32 reordering noise; north_star bar is 1e-4 absolute RMS)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tts_amd import ops

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(params=["h2", "x3", "f32"], autouse=True)
def conv_precision(request):
    """Every conv test runs on all three arithmetic paths at the SAME tolerance: the three-product two-part-fp16 kernels (default
    for large grids; forced onto the small test shapes here by switching the small-grid tiles off — tests that sweep the small-grid
    modes set them themselves), the six-product split-bf16 kernels, and the fp32-input MFMA kernels."""
    before = ops.conv_precision()
    ops.set_conv_precision(request.param)
    was = ops.set_conv_small_grid(0) if request.param == "h2" else None
    yield request.param
    if was is not None:
        ops.set_conv_small_grid(was)
    ops.set_conv_precision(before)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


CASES = [  # (B, Cin, Cout, K, D, T)
    (2, 32, 32, 3, 1, 700), (2, 32, 32, 7, 5, 1000), (1, 64, 64, 11, 3, 600), (2, 128, 128, 7, 1, 300),
    (1, 256, 256, 11, 5, 150), (2, 80, 512, 7, 1, 47), (2, 192, 384, 5, 1, 77), (3, 192, 96, 1, 1, 33),
    (2, 96, 192, 1, 1, 130), (2, 768, 192, 3, 1, 257), (1, 32, 1, 7, 1, 5000), (2, 192, 29, 1, 1, 50),
    (2, 3, 20, 3, 9, 40), (1, 17, 40, 5, 1, 1),
    # mid-size grids (48 .. 128 of the 128 x 128-class blocks): the 128 x 64 tile on eight waves (conv1d_h2_launch_mid)
    (1, 256, 256, 11, 1, 6160), (1, 128, 128, 7, 3, 8001), (2, 64, 128, 3, 5, 4000),
]


@pytest.mark.parametrize("case", CASES)
def test_conv1d_matches_torch(gpu, case):
    B, Cin, Cout, K, D, T = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / np.sqrt(Cin * K)
    b = torch.randn(Cout, generator=g)
    want = F.conv1d(F.leaky_relu(x, 0.1), w, b, padding=(K - 1) * D // 2, dilation=D)
    pc = ops.PackedConv(w, b, gpu, dilation=D)
    y = torch.full((B, Cout, T), float("nan"), device=gpu)
    ops.conv1d(pc, x.to(gpu), y, in_act=ops.ACT_LRELU, in_slope=0.1)
    assert _rel(y, want) < TOL


@pytest.mark.parametrize("case", [(1, 20, 32, 3, 1, 300), (2, 3, 32, 2, 1, 500), (1, 40, 64, 7, 3, 260), (1, 17, 128, 11, 1, 200)])
def test_rows_and_chunks_beyond_c_in_read_as_zero_not_as_whatever_follows_the_tensor(gpu, case):
    """The K loop walks 16-channel chunks and requests up to three chunks ahead: rows >= c_in (the rest of a partial chunk, chunks
    past the end) must come back as ZEROS through the buffer range check — not as the bytes that follow the tensor in memory.
    (Round 6 moved the row / chunk offsets of the staging loads into the load's scalar operand; the range check of a raw buffer
    access on gfx950 covers voffset + soffset — scripts/ubench/soffset_range.hip — and this test holds the kernels to it.  With the
    weight image zero-padded a stray read would vanish from the products unless the bytes are NaN / Inf — and the per-tile exponent
    would follow their magnitude.)  Here the tensor sits at the head of an allocation whose tail is NaN and 1e30: the result must
    not notice."""
    B, Cin, Cout, K, D, T = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / np.sqrt(Cin * K)
    b = torch.randn(Cout, generator=g)
    pad = (K - 1) * D // 2
    want = F.conv1d(F.pad(F.leaky_relu(x, 0.1), (pad, (K - 1) * D - pad)), w, b, dilation=D)
    n = B * Cin * T
    big = torch.full((n + 80 * T + 4096,), float("nan"), device=gpu)
    big[n + 7::2] = 1e30
    xg = big[:n].view(B, Cin, T)
    xg.copy_(x.to(gpu))
    pc = ops.PackedConv(w, b, gpu, dilation=D)
    y = torch.full((B, Cout, want.shape[2]), float("nan"), device=gpu)
    ops.conv1d(pc, xg, y, in_act=ops.ACT_LRELU, in_slope=0.1, t_out=want.shape[2])
    assert torch.isfinite(y).all() and _rel(y, want) < TOL


def test_conv1d_epilogue_res_accum_mask_div(gpu):
    g = torch.Generator().manual_seed(1)
    B, C, T, K = 2, 64, 333, 7
    x, res, acc = (torch.randn(B, C, T, generator=g) for _ in range(3))
    mask = (torch.arange(T)[None, :] < torch.tensor([333, 200])[:, None]).float()
    w = torch.randn(C, C, K, generator=g) / np.sqrt(C * K)
    b = torch.randn(C, generator=g)
    want = (acc + (F.conv1d(x * mask[:, None], w, b, padding=3) + res)) * mask[:, None] / 3.0
    pc = ops.PackedConv(w, b, gpu)
    y = torch.empty(B, C, T, device=gpu)
    ops.conv1d(pc, x.to(gpu), y, in_mask=mask.to(gpu), res=res.to(gpu), accum=acc.to(gpu), out_mask=mask.to(gpu),
               out_div=3.0)
    assert _rel(y, want) < TOL


# Launches with many more tiles than resident blocks, every optional operand.  (B, Cin, Cout, K, D, T, res, accum, mask)
MULTI_TILE = [
    (4, 32, 32, 3, 1, 100003, True, False, False), (3, 32, 32, 11, 5, 150000, False, True, False),
    (4, 64, 64, 7, 3, 40001, True, True, True), (4, 128, 128, 3, 1, 20011, True, False, False),
    (5, 16, 128, 3, 1, 16500, True, True, False), (2, 256, 256, 3, 1, 17000, True, False, True),
    (3, 48, 64, 1, 1, 50000, True, False, False), (6, 128, 128, 11, 1, 12000, False, False, False),
]


# Grids whose block count is a multiple of 8 take the XCD-aware block -> tile mapping (conv_tile_of_block): m-block = XCD
# index mod m-blocks, contiguous (item, time tile) ranges per XCD.  1, 2, 4 and 8 m-blocks, and a grid that is not a
# multiple of 8 (plain mapping).  (B, Cin, Cout, K, D, T)
XCD_CASES = [(4, 32, 128, 3, 1, 16384), (2, 32, 256, 3, 1, 16384), (2, 16, 512, 3, 1, 4096), (1, 16, 1024, 1, 1, 1024),
             (3, 32, 256, 3, 1, 5000), (8, 16, 64, 7, 1, 8192), (8, 16, 32, 7, 1, 16384),
             # round 6: small weight images take the x-local order (all m-blocks of a tile pair on one XCD) at ANY m-block count (3, 24);
             # images above TTSAMD_XLOCAL_WBYTES keep the weights-local order (2 m-blocks), extended to 16 m-blocks (two per XCD)
             (8, 32, 384, 3, 1, 1024), (1, 16, 3072, 1, 1, 1024), (1, 256, 256, 11, 1, 1024), (1, 512, 2048, 1, 1, 1024)]


@pytest.mark.parametrize("case", XCD_CASES)
def test_conv1d_xcd_tile_mapping(gpu, case):
    B, Cin, Cout, K, D, T = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / np.sqrt(Cin * K)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, T, generator=g)
    want = F.conv1d(x, w, b, padding=(K - 1) * D // 2, dilation=D) + res
    y = torch.full((B, Cout, T), float("nan"), device=gpu)
    ops.conv1d(ops.PackedConv(w, b, gpu, dilation=D), x.to(gpu), y, res=res.to(gpu))
    assert _rel(y, want) < TOL
    # every (item, row, column) written exactly by its own tile: per-item, per-row-block errors stay at rounding level
    err = (y.cpu() - want).abs().amax(dim=2)
    assert float(err.max()) < 1e-4


@pytest.mark.parametrize("case", MULTI_TILE)
def test_conv1d_multi_tile(gpu, case, conv_precision):
    B, Cin, Cout, K, D, T, has_res, has_acc, has_mask = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / np.sqrt(Cin * K)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, T, generator=g) if has_res else None
    acc = torch.randn(B, Cout, T, generator=g) if has_acc else None
    lens = torch.tensor([T - 37 * i for i in range(B)])
    mask = (torch.arange(T)[None, :] < lens[:, None]).float() if has_mask else None
    xin = x * mask[:, None] if has_mask else x
    want = F.conv1d(F.leaky_relu(xin, 0.1), w, b, padding=(K - 1) * D // 2, dilation=D)
    if has_res:
        want = want + res
    if has_acc:
        want = acc + want
    if has_mask:
        want = want * mask[:, None]
    pc = ops.PackedConv(w, b, gpu, dilation=D)
    dev = lambda t: None if t is None else t.to(gpu)
    y = torch.full((B, Cout, T), float("nan"), device=gpu)
    ops.conv1d(pc, x.to(gpu), y, in_act=ops.ACT_LRELU, in_slope=0.1, res=dev(res), accum=dev(acc), in_mask=dev(mask),
               out_mask=dev(mask))
    assert _rel(y, want) < TOL


@pytest.mark.parametrize("case", [(2, 64, 32, 8, 50), (1, 128, 64, 2, 301), (2, 32, 16, 2, 64), (1, 512, 256, 8, 20),
                                  (1, 512, 256, 8, 770), (1, 256, 128, 8, 1500)])     # the last two: the mid-size (eight-wave) tile
def test_conv_transpose_polyphase(gpu, case):
    B, Cin, Cout, u, T = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cin, Cout, 2 * u, generator=g) / np.sqrt(Cin * 2)
    b = torch.randn(Cout, generator=g)
    want = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=u, padding=u // 2)
    wp, bp = ops.convt_polyphase_weight(w, b, u)
    pc = ops.PackedConv(wp, bp, gpu, pad_left=1)
    y = torch.full((B, Cout, T * u), float("nan"), device=gpu)
    ops.conv1d(pc, x.to(gpu), y, in_act=ops.ACT_LRELU, in_slope=0.1, mode=ops.CONV_SHUFFLE, shuffle_u=u,
               shuffle_pad=u // 2)
    assert want.shape == y.shape
    assert _rel(y, want) < TOL


def test_gate_and_couple_modes(gpu):
    g = torch.Generator().manual_seed(3)
    B, H, T = 2, 192, 90
    x = torch.randn(B, H, T, generator=g)
    w = torch.randn(2 * H, H, 5, generator=g) / np.sqrt(H * 5)
    b = torch.randn(2 * H, generator=g) * 0.1
    xin = F.conv1d(x, w, b, padding=2)
    want = torch.tanh(xin[:, :H]) * torch.sigmoid(xin[:, H:])
    wg, bg = ops.gate_permute(w, b, H)
    y = torch.empty(B, H, T, device=gpu)
    ops.conv1d(ops.PackedConv(wg, bg, gpu), x.to(gpu), y, mode=ops.CONV_GATE)
    assert _rel(y, want) < TOL
    # mean-only coupling: x1 = (x1 - post(h)*mask) * mask, in place on the upper half of z
    z = torch.randn(B, H, T, generator=g)
    mask = (torch.arange(T)[None, :] < torch.tensor([90, 61])[:, None]).float()
    wpost = torch.randn(H // 2, H, 1, generator=g) / np.sqrt(H)
    bpost = torch.randn(H // 2, generator=g) * 0.1
    m = F.conv1d(x, wpost, bpost) * mask[:, None]
    want_z = z.clone()
    want_z[:, H // 2:] = (z[:, H // 2:] - m) * mask[:, None]
    zg = z.to(gpu)
    ops.conv1d(ops.PackedConv(wpost, bpost, gpu), x.to(gpu), zg, mode=ops.CONV_COUPLE, res=zg, out_mask=mask.to(gpu),
               y_row_offset=H // 2, res_row_offset=H // 2)
    assert _rel(zg, want_z) < TOL


def test_replicate_pad(gpu):
    x = torch.randn(3, 5, 17)
    y = torch.empty(3, 5, 27, device=gpu)
    ops.replicate_pad(x.to(gpu), y, 5)
    assert torch.equal(y.cpu(), F.pad(x, (5, 5), mode="replicate"))


# ---- split-bf16 arithmetic under adversarial operands (only randn had been tried) ----------------------------
def _f32_from_bits(sign, exp, mant):
    bits = (sign.astype(np.uint32) << 31) | (exp.astype(np.uint32) << 23) | mant.astype(np.uint32)
    return torch.from_numpy(bits.view(np.float32).copy())


def _max_residual_values(rng, shape, e_lo, e_hi):
    """fp32 values whose 3-way bf16 split has the LARGEST residuals: after the leading 8 significant bits, each further
    8-bit group sits next to the round-to-nearest tie (0x7F.. / 0x80..), so |x - bf16(x)| and |r1 - bf16(r1)| are ~1/2 ulp
    of the part before them — the products the kernel drops (w2 x3, w3 x2, w3 x3) are then as large as they can get."""
    n = int(np.prod(shape))
    hi = rng.integers(0, 128, n)                               # 7 explicit mantissa bits of part 1
    mid = rng.choice([0x7F, 0x80, 0x7E, 0x81], n)              # next 8 bits: next to the tie
    lo = rng.choice([0x7F, 0x80, 0xFF, 0x01], n)               # last 8 bits
    mant = (hi << 16) | (mid << 8) | lo
    return _f32_from_bits(rng.integers(0, 2, n), rng.integers(e_lo, e_hi + 1, n), mant).reshape(shape)


def _conv64(x, w, b, K, D):
    return F.conv1d(x.double(), w.double(), None if b is None else b.double(), padding=(K - 1) * D // 2, dilation=D)


@pytest.mark.parametrize("case", [(2, 64, 64, 7, 1, 400), (1, 256, 256, 11, 1, 260), (2, 32, 32, 3, 5, 900)])
def test_split_bf16_maximal_residual_operands_vs_fp64(gpu, case, conv_precision):
    """Both operands built to maximise the dropped products; the error against an fp64 conv must stay fp32-class: below
    2^-20 of sum|w x| on BOTH arithmetic paths (the classical worst-case bound of an fp32 dot product of this length is
    K * 2^-24 = 2^-15..2^-12.5 of sum|w x|; a random-walk estimate sqrt(K) * 2^-24 = 2^-19.6..2^-18.3).
    Measured on MI355X (max over outputs, K = 448 / 2816 / 96): split-bf16 5.0e-7 / 3.5e-7 / ~2e-7, the exact fp32-input
    MFMA path (a plain fmaf chain) 5.3e-7 at K = 2816 — i.e. the split path is no worse than sequential fp32 FMA — and
    torch's fp32 CPU conv (blocked, pairwise-like summation) 2.3e-7 / 8.4e-8."""
    B, Cin, Cout, K, D, T = case
    rng = np.random.default_rng(sum(case))
    x = _max_residual_values(rng, (B, Cin, T), 120, 130)        # |x| in [2^-7, 2^4)
    w = _max_residual_values(rng, (Cout, Cin, K), 115, 122)     # |w| in [2^-12, 2^-4)
    want = _conv64(x, w, None, K, D)
    scale = _conv64(x.abs(), w.abs(), None, K, D)               # sum |w x| per output
    y = torch.empty(B, Cout, T, device=gpu)
    ops.conv1d(ops.PackedConv(w, None, gpu, dilation=D), x.to(gpu), y)
    err_gpu = float(((y.cpu().double() - want).abs() / scale).max())
    err_cpu = float(((F.conv1d(x, w, None, padding=(K - 1) * D // 2, dilation=D).double() - want).abs() / scale).max())
    print("max |err| / sum|wx|: gpu %.3e, torch fp32 cpu %.3e (%s)" % (err_gpu, err_cpu, conv_precision))
    assert err_gpu < 2.0 ** -20, (err_gpu, err_cpu)
    assert _rel(y, want) < TOL


def test_split_bf16_cancellation_k2816_vs_fp64(gpu, conv_precision):
    """K = 256 channels x 11 taps = 2816 products of alternating sign and near-equal magnitude per output: the result is
    ~1e-3 of sum|w x|, so any systematic product error would surface.  Error measured against fp64, relative to sum|w x|."""
    B, C, K, T = 1, 256, 11, 300
    rng = np.random.default_rng(2816)
    sign = np.where((np.arange(C)[:, None] + np.arange(T)[None, :]) % 2 == 0, 1.0, -1.0)
    x = torch.from_numpy((sign * (1.0 + 1e-3 * rng.standard_normal((C, T)))).astype(np.float32))[None]
    w = torch.from_numpy((0.05 * (1.0 + 1e-3 * rng.standard_normal((C, C, K)))).astype(np.float32))
    want = _conv64(x, w, None, K, 1)
    scale = _conv64(x.abs(), w.abs(), None, K, 1)
    y = torch.empty(B, C, T, device=gpu)
    ops.conv1d(ops.PackedConv(w, None, gpu), x.to(gpu), y)
    err_gpu = float(((y.cpu().double() - want).abs() / scale).max())
    err_cpu = float(((F.conv1d(x, w, None, padding=5).double() - want).abs() / scale).max())
    print("cancellation: |result|/sum|wx| median %.2e; max err/sum|wx|: gpu %.3e, torch fp32 cpu %.3e"
          % (float((want.abs() / scale).median()), err_gpu, err_cpu))
    assert err_gpu < 2.0 ** -20, (err_gpu, err_cpu)


def test_split_bf16_tiny_activations_flush_bound(gpu, conv_precision):
    """|x| in [1e-40, 1e-36]: the 2nd / 3rd bf16 parts of such values fall below bf16's normal range (1.18e-38).  Whatever
    the hardware does with them (keep as denormals or flush), the error is bounded by the flushed parts themselves:
    |err| <= 2 * 2^-126 * sum|w| per output (+ fp32-level relative error) — and a batch that mixes such values with
    ordinary ones keeps the ordinary tolerance."""
    B, C, K, T = 1, 64, 7, 500
    rng = np.random.default_rng(40)
    mag = 10.0 ** rng.uniform(-40, -36, (B, C, T))
    x = torch.from_numpy((mag * rng.choice([-1.0, 1.0], mag.shape)).astype(np.float32))
    w = torch.randn(C, C, K, generator=torch.Generator().manual_seed(1)) / np.sqrt(C * K)
    want = _conv64(x, w, None, K, 1)
    y = torch.empty(B, C, T, device=gpu)
    ops.conv1d(ops.PackedConv(w, None, gpu), x.to(gpu), y)
    bound = 2.0 * 2.0 ** -126 * float(w.abs().sum(dim=(1, 2)).max()) + 1e-5 * float(want.abs().max())
    err = float((y.cpu().double() - want).abs().max())
    print("tiny activations: max |err| %.3e (bound %.3e, max |want| %.3e)" % (err, bound, float(want.abs().max())))
    assert err <= bound
    xm = torch.randn(B, C, T, generator=torch.Generator().manual_seed(2))
    xm[:, ::3] = x[:, ::3]                                       # a third of the channels tiny, the rest ordinary
    ops.conv1d(ops.PackedConv(w, None, gpu), xm.to(gpu), y)
    assert _rel(y, _conv64(xm, w, None, K, 1)) < TOL


def test_h2_running_exponent_rescales_and_wide_range(gpu):
    """Three-product kernels: (i) channel magnitudes growing by 2^20 over the reduction force the block's running activation
    exponent to be renewed (accumulators rescaled) several times; (ii) a tile mixing magnitudes 1e-5 .. 1e2; (iii) weight rows
    spanning 1e-2 .. 1e1 (weight-norm gains of a trained model) — all against an fp64 conv, relative to sum|w x| per output."""
    was_p, was_g = ops.conv_precision(), ops.set_conv_small_grid(0)
    ops.set_conv_precision("h2")
    try:
        rng = np.random.default_rng(5)
        for name, C, K, T in (("growing", 256, 7, 300), ("shrinking", 256, 11, 200), ("wide", 128, 11, 300), ("gains", 128, 3, 500)):
            x = torch.randn(2, C, T, generator=torch.Generator().manual_seed(C + K))
            w = torch.randn(C, C, K, generator=torch.Generator().manual_seed(K)) / np.sqrt(C * K)
            if name == "growing":
                x = x * (2.0 ** (torch.arange(C).float() / C * 20.0 - 10.0))[None, :, None]
            elif name == "shrinking":
                x = x * (2.0 ** (10.0 - torch.arange(C).float() / C * 30.0))[None, :, None]
            elif name == "wide":
                x = torch.from_numpy((10.0 ** rng.uniform(-5, 2, (2, C, T)) * rng.choice([-1.0, 1.0], (2, C, T))).astype(np.float32))
            else:
                w = w * torch.from_numpy(10.0 ** rng.uniform(-2, 1, (C, 1, 1))).float()
            res = torch.randn(2, C, T, generator=torch.Generator().manual_seed(3))
            want = _conv64(x, w, None, K, 1) + res.double()
            scale = _conv64(x.abs(), w.abs(), None, K, 1) + res.abs().double()
            y = torch.empty(2, C, T, device=gpu)
            ops.conv1d(ops.PackedConv(w, None, gpu), x.to(gpu), y, res=res.to(gpu))
            err = float(((y.cpu().double() - want).abs() / scale).max())
            print("%s: max |err| / (sum|wx| + |res|) = %.3e" % (name, err))
            assert err < 2.0 ** -20, (name, err)
    finally:
        ops.set_conv_precision(was_p)
        ops.set_conv_small_grid(was_g)


def test_h2_residual_fold_guard_and_zero_chunks(gpu):
    """Three-product kernels, corners of the scaling: tiny activations x tiny weights next to an ORDINARY residual (the residual
    cannot be folded into accumulators whose unit is 2^-(activation + row exponent): the epilogue adds it instead); all-zero
    leading chunks (masked / padded channels: exponent set by the first chunk with data); an all-zero input."""
    was_p, was_g = ops.conv_precision(), ops.set_conv_small_grid(0)
    ops.set_conv_precision("h2")
    try:
        C, K, T = 128, 7, 300
        g = torch.Generator().manual_seed(9)
        w = torch.randn(C, C, K, generator=g) / np.sqrt(C * K)
        res = torch.randn(1, C, T, generator=g)
        x = torch.randn(1, C, T, generator=g)
        for name, xs, ws in (("tiny x tiny + ordinary residual", 1e-25, 1e-12), ("zero leading chunks", 1.0, 1.0), ("all zero", 0.0, 1.0)):
            xx = x * xs
            if name.startswith("zero"):
                xx = xx.clone()
                xx[:, :48] = 0.0
            ww = w * ws
            want = _conv64(xx, ww, None, K, 1) + res.double()
            y = torch.empty(1, C, T, device=gpu)
            ops.conv1d(ops.PackedConv(ww, None, gpu), xx.to(gpu), y, res=res.to(gpu))
            assert torch.isfinite(y).all(), name
            assert _rel(y, want) < TOL, name
            y2 = torch.empty(1, C, T, device=gpu)            # the conv part alone keeps its own relative accuracy
            ops.conv1d(ops.PackedConv(ww, None, gpu), xx.to(gpu), y2)
            if xs:
                assert _rel(y2, _conv64(xx, ww, None, K, 1)) < TOL, name
            else:
                assert float(y2.abs().max()) == 0.0
    finally:
        ops.set_conv_precision(was_p)
        ops.set_conv_small_grid(was_g)


@pytest.mark.parametrize("case", [(2, 32, 5001, True, True), (3, 8, 250, False, True), (1, 32, 3, True, False), (2, 64, 247, True, True),
                                  (1, 16, 249, False, False), (2, 32, 1, True, True), (1, 32, 100000, False, True)])
def test_single_output_channel_streaming_conv(gpu, case, conv_precision):
    """C -> 1, k = 7 (HiFiGAN conv_post) takes the streaming kernel (conv_post.hip: 16-byte row loads, neighbour columns by
    DPP shifts, exact fp32 FMA chain): leaky-ReLU(0.01) in, tanh out, optional length mask and bias; lengths around the
    248-column wave tile, shorter than one lane's 4 columns, and rows whose 16-byte loads are not 16-byte aligned."""
    B, C, T, has_mask, has_bias = case
    g = torch.Generator().manual_seed(sum(case[:3]))
    x = torch.randn(B, C, T, generator=g)
    w = torch.randn(1, C, 7, generator=g) / np.sqrt(C * 7)
    b = torch.randn(1, generator=g) if has_bias else None
    lens = torch.tensor([max(1, T - 61 * i) for i in range(B)])
    mask = (torch.arange(T)[None, :] < lens[:, None]).float() if has_mask else None
    xin = x * mask[:, None] if has_mask else x
    want = torch.tanh(F.conv1d(F.leaky_relu(xin, 0.01), w, b, padding=3))
    y = torch.full((B, 1, T), float("nan"), device=gpu)
    ops.conv1d(ops.PackedConv(w, b, gpu), x.to(gpu), y, in_act=ops.ACT_LRELU, in_slope=0.01, out_act=ops.ACT_TANH,
               in_mask=None if mask is None else mask.to(gpu))
    assert float((y.cpu() - want).abs().max()) < 2e-6


@pytest.mark.parametrize("case", [(1, 768, 192, 3, 257, 0), (1, 192, 768, 3, 257, 0), (1, 192, 384, 5, 770, 1), (2, 192, 192, 1, 159, 0),
                                  (1, 200, 100, 7, 300, 0), (1, 144, 64, 3, 65, 0), (3, 256, 256, 3, 40, 0),
                                  (1, 192, 576, 1, 257, 0), (1, 256, 192, 1, 257, 0), (2, 208, 64, 1, 130, 0), (1, 400, 128, 1, 70, 0),
                                  (1, 192, 192, 5, 257, 0), (1, 130, 64, 5, 64, 0), (1, 192, 384, 3, 300, 1), (2, 256, 256, 5, 63, 1),
                                  (1, 512, 256, 3, 770, 0), (1, 256, 256, 3, 6000, 0), (1, 192, 256, 1, 5000, 0),
                                  (1, 80, 192, 1, 159, 0), (1, 80, 128, 7, 338, 0), (1, 192, 512, 7, 770, 0), (1, 192, 80, 1, 64, 0),
                                  (1, 256, 1, 1, 64, 0), (1, 100, 104, 3, 33, 1), (1, 48, 40, 5, 20, 1), (2, 64, 96, 7, 31, 0)])
def test_small_grid_tiles_and_k_split(gpu, case, conv_precision):
    """Launches that underfill the chip (single-sentence shapes): mode 1 (64-column tiles, one 32x32 tile per wave) is
    bitwise the large-grid result; mode 2 adds wave groups that split the K loop (chunk counts that do not divide by the
    group count included); mode 3 takes the small-grid kernels of conv_kernel_x3s.h (single-iteration 1x1 convs at <= 192 channels,
    multi-iteration ones with a partly filled last iteration, halo rounds of k = 3 / 5, paired gate rows, eight K slices at
    >= 512 channels, 64x64 tiles on the longer launches); mode 4 (default) takes the one-shot kernels of conv_kernel_x3o.h
    first (a wave per K slice, 4 / 8 / 12 / 16 of them, the epilogue spread over four waves).
    All stay within the conv tolerance of torch and of mode 0."""
    B, Cin, Cout, K, T, gate = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / np.sqrt(Cin * K)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, T, generator=g)
    pre = F.conv1d(F.leaky_relu(x, 0.1), w, b, padding=(K - 1) // 2)
    if gate:
        H = Cout // 2
        want = torch.tanh(pre[:, :H]) * torch.sigmoid(pre[:, H:])
        wp, bp = ops.gate_permute(w, b, H)
        pc = ops.PackedConv(wp, bp, gpu)
    else:
        want = pre + res
        pc = ops.PackedConv(w, b, gpu)
    outs = {}
    for mode in (0, 1, 2, 3, 4):
        was = ops.set_conv_small_grid(mode)
        try:
            y = torch.full(want.shape, float("nan"), device=gpu)
            if gate:
                ops.conv1d(pc, x.to(gpu), y, in_act=ops.ACT_LRELU, in_slope=0.1, mode=ops.CONV_GATE)
            else:
                ops.conv1d(pc, x.to(gpu), y, in_act=ops.ACT_LRELU, in_slope=0.1, res=res.to(gpu))
        finally:
            ops.set_conv_small_grid(was)
        assert _rel(y, want) < TOL, mode
        outs[mode] = y
    if conv_precision == "x3":
        assert torch.equal(outs[0], outs[1])
        for mode in (2, 3, 4):
            assert _rel(outs[mode], outs[0]) < 2e-6, mode


@pytest.mark.parametrize("T", [159, 31, 770])
def test_one_shot_kernel_fused_epilogues(gpu, T, conv_precision):
    """The one-shot small-grid kernel (conv_kernel_x3o.h) under every fused 1x1 epilogue of the flows — WaveNet res/skip split,
    mean-only coupling, Glow's affine coupling both ways with a partly filled last pair tile (80 channels) — with per-item
    row biases and a ragged mask, against torch; and against the looping kernels (small-grid mode 3) and the large-grid kernels (mode 0:
    under precision "h2" the three-product kernel with every paired-row / split epilogue)."""
    g = torch.Generator().manual_seed(T)
    B, H = 2, 192
    dev = gpu
    mask = (torch.arange(T)[None, :] < torch.tensor([T, max(1, T - 7)])[:, None]).float()
    acts = torch.randn(B, H, T, generator=g)
    outs = {}
    for mode in (0, 3, 4):           # 0: the large-grid kernels under the same epilogues (the three-product kernel under "h2")
        was = ops.set_conv_small_grid(mode)
        try:
            # res/skip: rows < H: x = (x + v) * mask, rows >= H: out = out + v
            w = torch.randn(2 * H, H, 1, generator=torch.Generator().manual_seed(1)) / np.sqrt(H)
            b = torch.randn(2 * H, generator=torch.Generator().manual_seed(2)) * 0.1
            x0 = torch.randn(B, H, T, generator=torch.Generator().manual_seed(3))
            o0 = torch.randn(B, H, T, generator=torch.Generator().manual_seed(4))
            v = F.conv1d(acts, w, b)
            want_x, want_o = (x0 + v[:, :H]) * mask[:, None], o0 + v[:, H:]
            xg, og = x0.to(dev), o0.to(dev)
            ops.conv1d(ops.PackedConv(w, b, dev), acts.to(dev), xg, mode=ops.CONV_RES_SKIP, res=xg, out_mask=mask.to(dev), y2=og,
                       accum=og, split_row=H)
            assert _rel(xg, want_x) < TOL and _rel(og, want_o) < TOL, mode
            # Glow affine coupling, reverse and forward, 80 coupled channels of a 160-channel tensor, in place
            half = 80
            we = torch.randn(2 * half, H, 1, generator=torch.Generator().manual_seed(5)) / np.sqrt(H)
            be = torch.randn(2 * half, generator=torch.Generator().manual_seed(6)) * 0.1
            wp, bp = ops.pair_permute(we, be, half, half)
            pc = ops.PackedConv(wp, bp, dev)
            z = torch.randn(B, 2 * half, T, generator=torch.Generator().manual_seed(7))
            ts = F.conv1d(acts, we, be)
            t_, s_ = ts[:, :half], ts[:, half:]
            want_r, want_f = z.clone(), z.clone()
            want_r[:, half:] = (z[:, half:] - t_) * torch.exp(-s_) * mask[:, None]
            want_f[:, half:] = (t_ + torch.exp(s_) * z[:, half:]) * mask[:, None]
            for cmode, want in ((ops.CONV_COUPLE_AFFINE, want_r), (ops.CONV_COUPLE_AFFINE_FWD, want_f)):
                zg = z.to(dev)
                ops.conv1d(pc, acts.to(dev), zg, mode=cmode, res=zg, res_row_offset=half, y_row_offset=half, out_mask=mask.to(dev),
                           split_row=half)
                assert _rel(zg, want) < TOL, (mode, cmode)
                outs[(mode, cmode)] = zg
            # gate with a per-item row bias (speaker conditioning)
            wg = torch.randn(2 * H, H, 5, generator=torch.Generator().manual_seed(8)) / np.sqrt(5 * H)
            bg = torch.randn(2 * H, generator=torch.Generator().manual_seed(9)) * 0.1
            rb = torch.randn(B, 2 * H, generator=torch.Generator().manual_seed(10)) * 0.3
            pre = F.conv1d(acts, wg, bg, padding=2) + rb[:, :, None]
            want_g = torch.tanh(pre[:, :H]) * torch.sigmoid(pre[:, H:])
            wgp, bgp = ops.gate_permute(wg, bg, H)
            idx = torch.tensor(ops.pair_index(H, H))
            y = torch.empty(B, H, T, device=dev)
            ops.conv1d(ops.PackedConv(wgp, bgp, dev), acts.to(dev), y, mode=ops.CONV_GATE, row_bias=rb[:, idx].contiguous().to(dev))
            assert _rel(y, want_g) < TOL, mode
            outs[(mode, "gate")] = y
        finally:
            ops.set_conv_small_grid(was)
    if conv_precision == "x3":
        for k in (ops.CONV_COUPLE_AFFINE, ops.CONV_COUPLE_AFFINE_FWD, "gate"):
            assert _rel(outs[(4, k)], outs[(3, k)]) < 2e-6, k
    for k in (ops.CONV_COUPLE_AFFINE, ops.CONV_COUPLE_AFFINE_FWD, "gate"):
        assert _rel(outs[(0, k)], outs[(3, k)]) < 4e-6, k


@pytest.mark.parametrize("case", [(2, 40, 48, 9, 2, 130), (1, 192, 70, 13, 1, 77), (1, 16, 16, 4, 1, 50), (2, 33, 20, 31, 27, 900),
                                  (1, 64, 64, 3, 2, 200), (1, 96, 32, 6, 7, 65)])
def test_generic_conv_any_kernel_and_dilation(gpu, case):
    """Shapes without a tuned instantiation (ttsamd_conv1d_tuned == 0: k = 9 d = 2, k = 13, even kernels, k = 31 d = 27, a tuned
    kernel size at an untuned dilation) run on the generic split-bf16 kernel with the full NORMAL epilogue — against torch."""
    from tts_amd import _lib

    B, Cin, Cout, K, D, T = case
    assert _lib.lib().ttsamd_conv1d_supported(K, D) == 1 and _lib.lib().ttsamd_conv1d_tuned(K, D) == 0
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / np.sqrt(Cin * K)
    b = torch.randn(Cout, generator=g)
    pad = (K - 1) * D // 2
    t_out = T + 2 * pad - (K - 1) * D
    res = torch.randn(B, Cout, t_out, generator=g)
    acc = torch.randn(B, Cout, t_out, generator=g)
    lens = torch.tensor([T, max(1, T - 13)])[:B]
    mask = (torch.arange(T)[None, :] < lens[:, None]).float()
    omask = mask[:, :t_out] if t_out <= T else torch.ones(B, t_out)
    want = (F.conv1d(F.leaky_relu(x * mask[:, None], 0.1), w, b, padding=pad, dilation=D) + res + acc) * omask[:, None] / 3.0
    y = torch.full(want.shape, float("nan"), device=gpu)
    pc = ops.PackedConv(w, b, gpu, dilation=D)
    assert not pc.tuned
    ops.conv1d(pc, x.to(gpu), y, t_out=t_out, in_act=ops.ACT_LRELU, in_slope=0.1, in_mask=mask.to(gpu), res=res.to(gpu),
               accum=acc.to(gpu), out_mask=omask.contiguous().to(gpu), out_div=3.0)
    assert _rel(y, want) < TOL
    assert _lib.lib().ttsamd_conv1d_supported(32, 1) == 0 and _lib.lib().ttsamd_conv1d_supported(3, 28) == 0


def test_generic_gate_conv_and_polyphase_transposed_conv(gpu):
    """A WaveNet gate conv at k = 7, dilation 2 (VitsArgs kernel_size / dilation_rate are free parameters) and ConvTranspose1d
    layers whose kernel is not twice the stride — k = 7 u = 3 (three taps per phase, k - u even), k = 4 u = 4 (one tap),
    k = 6 u = 4 — in polyphase form on the generic kernel, against torch."""
    g = torch.Generator().manual_seed(11)
    B, H, T = 2, 64, 90
    x = torch.randn(B, H, T, generator=g)
    w = torch.randn(2 * H, H, 7, generator=g) / np.sqrt(H * 7)
    b = torch.randn(2 * H, generator=g) * 0.1
    pre = F.conv1d(x, w, b, padding=6, dilation=2)
    want = torch.tanh(pre[:, :H]) * torch.sigmoid(pre[:, H:])
    wg, bg = ops.gate_permute(w, b, H)
    y = torch.empty(B, H, T, device=gpu)
    ops.conv1d(ops.PackedConv(wg, bg, gpu, dilation=2), x.to(gpu), y, mode=ops.CONV_GATE)
    assert _rel(y, want) < TOL
    for k, u, cin, cout in ((7, 3, 48, 24), (4, 4, 32, 16), (6, 4, 40, 20), (16, 8, 64, 32)):
        wt = torch.randn(cin, cout, k, generator=g) / np.sqrt(cin * k)
        bt = torch.randn(cout, generator=g) * 0.1
        xin = torch.randn(B, cin, 37, generator=g)
        pad = (k - u) // 2
        want = F.conv_transpose1d(F.leaky_relu(xin, 0.1), wt, bt, u, pad)
        wp, bp = ops.convt_polyphase_weight(wt, bt, u)
        pc = ops.PackedConv(wp, bp, gpu, pad_left=wp.shape[2] - 1)
        y = torch.full(want.shape, float("nan"), device=gpu)
        ops.conv1d(pc, xin.to(gpu), y, t_out=37 + wp.shape[2] - 1, in_act=ops.ACT_LRELU, in_slope=0.1, mode=ops.CONV_SHUFFLE,
                   shuffle_u=u, shuffle_pad=pad)
        assert y.shape == want.shape and _rel(y, want) < TOL, (k, u)
