"""GPU parity: HIP monotonic alignment search vs the CPU oracle — bit-exact (integer path)."""
import numpy as np
import pytest
import torch

from oracle import mas
from tts_amd import helpers

pytestmark = pytest.mark.gpu


def _problem(rng, B, TX, TY, ties=False, full=False):
    tx = np.full(B, TX) if full else rng.integers(max(1, TX // 2), TX + 1, B)
    ty = np.full(B, TY) if full else rng.integers(max(TX, TY // 2), TY + 1, B)
    tx[0], ty[0] = TX, TY
    ty = np.maximum(ty, tx)
    mask = (mas.sequence_mask(tx, TX)[:, :, None] & mas.sequence_mask(ty, TY)[:, None, :]).astype(np.float32)
    v = (rng.integers(-2, 3, (B, TX, TY)) if ties else rng.standard_normal((B, TX, TY))).astype(np.float32)
    return v, mask, tx.astype(np.int32), ty.astype(np.int32)


SHAPES = [(4, 17, 40), (3, 64, 64), (2, 65, 200), (5, 1, 9), (2, 7, 7), (1, 130, 257), (3, 1, 1),
          (2, 300, 641), (1, 520, 1100), (1, 1030, 1200), (32, 257, 770)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("ties", [False, True])
def test_maximum_path_bit_exact(gpu, shape, ties):
    rng = np.random.default_rng(sum(shape) * 2 + ties)
    v, mask, tx, ty = _problem(rng, *shape, ties=ties)
    want = mas.maximum_path(v, mask, "c")
    got = helpers.maximum_path(torch.from_numpy(v).to(gpu), torch.from_numpy(mask).to(gpu))
    assert got.dtype == torch.float32 and got.shape == tuple(v.shape)
    assert np.array_equal(got.cpu().numpy().astype(np.int32), want)


@pytest.mark.parametrize("shape", [(4, 17, 40), (2, 130, 257), (3, 257, 770)])
def test_maximum_path_c_mirror_in_place_values(gpu, shape):
    """Device mirror of core.pyx:42: values updated in place bit-identically, paths pre-zeroed."""
    rng = np.random.default_rng(5)
    v, mask, tx, ty = _problem(rng, *shape)
    v = v * mask
    want_v = v.copy()
    want_p = np.zeros(v.shape, np.int32)
    mas.maximum_path_c(want_p, want_v, tx, ty)
    dv = torch.from_numpy(v).to(gpu)
    dp = torch.zeros(v.shape, dtype=torch.int32, device=gpu)
    helpers.maximum_path_c(dp, dv, torch.from_numpy(tx).to(gpu), torch.from_numpy(ty).to(gpu))
    assert np.array_equal(dp.cpu().numpy(), want_p)
    assert np.array_equal(dv.cpu().numpy().view(np.uint32), want_v.view(np.uint32))


@pytest.mark.parametrize("shape", [(3, 40, 25), (2, 130, 64), (2, 257, 100)])
def test_items_with_fewer_columns_than_rows(gpu, shape):
    """t_y < t_x: core.pyx:21-27 then fills no cell at all (the band is empty) and the backtrack (core.pyx:32-37) walks the
    UNMODIFIED values; no caller of the reference produces such items (a mel is longer than its text), the kernels still have to
    agree with the Cython core on them, ragged lengths included."""
    B, TX, TY = shape
    rng = np.random.default_rng(TX + TY)
    tx = rng.integers(TY + 1, TX + 1, B)
    ty = rng.integers(max(1, TY // 2), TY + 1, B)
    tx[0], ty[0] = TX, TY
    mask = (mas.sequence_mask(tx, TX)[:, :, None] & mas.sequence_mask(ty, TY)[:, None, :]).astype(np.float32)
    v = rng.standard_normal((B, TX, TY)).astype(np.float32)
    want = mas.maximum_path(v, mask, "c")
    got = helpers.maximum_path(torch.from_numpy(v).to(gpu), torch.from_numpy(mask).to(gpu))
    assert np.array_equal(got.cpu().numpy().astype(np.int32), want)


def test_non_rectangular_mask_is_applied(gpu):
    """value*mask happens inside the kernel (helpers.py:184) even for masks with holes."""
    rng = np.random.default_rng(9)
    v, mask, tx, ty = _problem(rng, 3, 40, 90, full=True)
    mask[:, 5:9, 20:30] = 0.0  # holes inside the band; lengths (row/col 0 sums) unchanged
    want = mas.maximum_path(v, mask, "c")
    got = helpers.maximum_path(torch.from_numpy(v).to(gpu), torch.from_numpy(mask).to(gpu))
    assert np.array_equal(got.cpu().numpy().astype(np.int32), want)


def test_full_size_property(gpu):
    """BASELINE MAS shape 32x257x770: size-independent properties (one cell per column, monotone)."""
    rng = np.random.default_rng(0)
    v, mask, tx, ty = _problem(rng, 32, 257, 770)
    got = helpers.maximum_path(torch.from_numpy(v).to(gpu), torch.from_numpy(mask).to(gpu)).cpu().numpy()
    for i in range(32):
        cols = got[i].sum(0)
        assert (cols[: ty[i]] == 1).all() and (cols[ty[i]:] == 0).all()
        rows = got[i].argmax(0)[: ty[i]]
        assert rows[0] == 0 and rows[-1] == tx[i] - 1
        assert ((np.diff(rows) == 0) | (np.diff(rows) == 1)).all()


def test_mas_logp_and_attention_match_oracle(gpu):
    """vits.py:909-919: logp from (z_p, m_p, logs_p) then maximum_path — logp to fp32 tolerance, and the path
    bit-exact with the CPU MAS oracle run on the SAME logp (MAS is exact given its input)."""
    import torch

    from oracle import tts_oracle as O

    g = torch.Generator().manual_seed(11)
    B, C, Tx, Ty = 3, 192, 45, 131
    z = torch.randn(B, C, Ty, generator=g)
    m = torch.randn(B, C, Tx, generator=g)
    logs = 0.3 * torch.randn(B, C, Tx, generator=g)
    for glow in (False, True):
        want = O.mas_logp(z, m, logs, glow)
        got = helpers.mas_logp(z.to(gpu), m.to(gpu), logs.to(gpu), glow)
        rel = float((got.cpu().double() - want.double()).pow(2).mean().sqrt() / want.double().pow(2).mean().sqrt())
        assert rel < 1e-6, rel
    xl, yl = np.array([45, 30, 7], np.int32), np.array([131, 100, 9], np.int32)
    xm = torch.from_numpy(mas.sequence_mask(xl, Tx)).float()
    ym = torch.from_numpy(mas.sequence_mask(yl, Ty)).float()
    attn = helpers.mas_attention(z.to(gpu), m.to(gpu), logs.to(gpu), xm.to(gpu), ym.unsqueeze(1).to(gpu))
    mask = (xm[:, :, None] * ym[:, None, :]).numpy()
    ref = mas.maximum_path(got.cpu().numpy(), mask, "c")
    assert np.array_equal(attn.cpu().numpy().astype(np.int32), ref)
    assert np.array_equal(attn.sum(1).cpu().numpy(), ym.numpy())      # every valid frame is aligned to exactly one token


@pytest.mark.parametrize("shape", [(1, 2100, 2300), (2, 2500, 2560)])
def test_maximum_path_beyond_2048_rows(gpu, shape):
    """core.pyx:11-47 has no bound on t_x: beyond 32 row groups the column-stepping kernel takes over — bit-exact like
    the others, ragged items, mask applied inside, in-place mirror included."""
    rng = np.random.default_rng(shape[1])
    v, mask, tx, ty = _problem(rng, *shape)
    want = mas.maximum_path(v, mask, "c")
    got = helpers.maximum_path(torch.from_numpy(v).to(gpu), torch.from_numpy(mask).to(gpu))
    assert np.array_equal(got.cpu().numpy().astype(np.int32), want)
    vm = v * mask
    want_v, want_p = vm.copy(), np.zeros(v.shape, np.int32)
    mas.maximum_path_c(want_p, want_v, tx, ty)
    dv = torch.from_numpy(vm).to(gpu)
    dp = torch.zeros(v.shape, dtype=torch.int32, device=gpu)
    helpers.maximum_path_c(dp, dv, torch.from_numpy(tx).to(gpu), torch.from_numpy(ty).to(gpu))
    assert np.array_equal(dp.cpu().numpy(), want_p)
    assert np.array_equal(dv.cpu().numpy().view(np.uint32), want_v.view(np.uint32))


@pytest.mark.parametrize("env", [{"TTSAMD_MAS_FORCE_BIG": "1"}, {"TTSAMD_MAS_FORCE_BIG": "2"}, {"TTSAMD_MAS_MW": "1"},
                                 {"TTSAMD_MAS_SINGLE_WAVE": "1"}, {"TTSAMD_MAS_BT": "1"}],
                         ids=["any_tx_lds", "any_tx_workspace", "mw_round3_column_step", "one_dp_wave", "round2_backtrack_walk"])
def test_any_tx_kernel_on_the_small_cases(gpu, env):
    """The kernels the default dispatch does not pick for these shapes, forced onto them in a fresh process: the any-T_x kernel
    (TTSAMD_MAS_FORCE_BIG=1: column state in LDS, =2: in the workspace, the T_x > 16 384 arrangement), the round-3 column
    step of the skewed pipeline (TTSAMD_MAS_MW=1; the default since round 6 is the branch-free step), the one-DP-wave kernel,
    the round-2 backtrack walk (TTSAMD_MAS_BT=1; default since round 6: conditions folded into the window words).
    Same bit-exact cases, ties, ragged items, the in-place mirror."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
from oracle import mas
from tts_amd import helpers
from tests.test_mas_gpu import _problem
gpu = torch.device("cuda:0")
for shape in [(4, 17, 40), (3, 64, 64), (2, 65, 200), (5, 1, 9), (2, 7, 7), (1, 130, 257), (3, 1, 1), (2, 300, 641), (8, 257, 770)]:
    for ties in (False, True):
        rng = np.random.default_rng(sum(shape) * 2 + ties)
        v, mask, tx, ty = _problem(rng, *shape, ties=ties)
        want = mas.maximum_path(v, mask, "c")
        got = helpers.maximum_path(torch.from_numpy(v).to(gpu), torch.from_numpy(mask).to(gpu))
        assert np.array_equal(got.cpu().numpy().astype(np.int32), want), (shape, ties)
    vm = v * mask
    want_v, want_p = vm.copy(), np.zeros(v.shape, np.int32)
    mas.maximum_path_c(want_p, want_v, tx, ty)
    dv = torch.from_numpy(vm).to(gpu)
    dp = torch.zeros(v.shape, dtype=torch.int32, device=gpu)
    helpers.maximum_path_c(dp, dv, torch.from_numpy(tx).to(gpu), torch.from_numpy(ty).to(gpu))
    assert np.array_equal(dp.cpu().numpy(), want_p), shape
    assert np.array_equal(dv.cpu().numpy().view(np.uint32), want_v.view(np.uint32)), shape
print("any-T_x kernel OK")
''' % root
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=280,
                       env=dict(os.environ, **env), cwd=root)
    assert p.returncode == 0 and "any-T_x kernel OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
