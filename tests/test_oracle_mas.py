"""Pins the MAS / alignment-helper oracle (CPU): C restatement == the reference's own Cython core
(oracle/_ref, compiled from /root/reference) == numpy restatement, plus the structural facts the
reference's own tests assert (tests/tts_tests/test_helpers.py:24-30,71-88)."""
import numpy as np
import pytest

from oracle import mas


def _problem(rng, B, TX, TY, ragged=True, ties=False):
    tx = rng.integers(max(1, TX // 2), TX + 1, B) if ragged else np.full(B, TX)
    ty = rng.integers(max(TX, TY // 2), TY + 1, B) if ragged else np.full(B, TY)
    tx[0], ty[0] = TX, TY
    ty = np.maximum(ty, tx)
    mask = (mas.sequence_mask(tx, TX)[:, :, None] & mas.sequence_mask(ty, TY)[:, None, :]).astype(np.float32)
    if ties:
        v = rng.integers(-2, 3, (B, TX, TY)).astype(np.float32)
    else:
        v = rng.standard_normal((B, TX, TY)).astype(np.float32)
    return v, mask, tx.astype(np.int32), ty.astype(np.int32)


@pytest.mark.parametrize("shape", [(4, 17, 40), (3, 64, 64), (2, 65, 200), (5, 1, 9), (2, 7, 7), (1, 130, 257)])
@pytest.mark.parametrize("ties", [False, True])
def test_c_restatement_matches_reference_cython_and_numpy(shape, ties):
    rng = np.random.default_rng(sum(shape) + ties)
    v, mask, tx, ty = _problem(rng, *shape, ties=ties)
    a = mas.maximum_path(v, mask, "c")
    ref = mas.ref_maximum_path_c()
    if ref is not None:
        b = mas.maximum_path(v, mask, "ref")
        assert np.array_equal(a, b)
        # in-place DP values are bit-identical too
        v1, v2 = (v * mask).copy(), (v * mask).copy()
        p1, p2 = np.zeros(v.shape, np.int32), np.zeros(v.shape, np.int32)
        mas.maximum_path_c(p1, v1, tx, ty)
        ref(p2, v2, tx, ty)
        assert np.array_equal(v1.view(np.uint32), v2.view(np.uint32))
    if not ties:  # numpy fallback breaks ties the other way (>= vs <), helpers.py:214 vs core.pyx:36
        c = mas.maximum_path_numpy(v, mask)
        assert np.array_equal(a.astype(np.float32), c)
    # structure: binary, one cell per valid column, monotone, ends at (t_x-1, t_y-1)
    assert set(np.unique(a)) <= {0, 1}
    for i in range(shape[0]):
        cols = a[i].sum(0)
        assert (cols[: ty[i]] == 1).all() and (cols[ty[i]:] == 0).all()
        rows = a[i].argmax(0)[: ty[i]]
        assert rows[0] == 0 and rows[-1] == tx[i] - 1
        assert ((np.diff(rows) == 0) | (np.diff(rows) == 1)).all()


@pytest.mark.parametrize("shape", [(3, 40, 25), (2, 130, 64), (2, 257, 100)])
def test_c_restatement_matches_reference_on_items_with_fewer_columns_than_rows(shape):
    """t_y < t_x (no caller produces it; core.pyx:21-27 fills nothing, :32-37 walks the unmodified values): the restatement
    follows the Cython core there too — the GPU test of the same name compares the kernels with it."""
    ref = mas.ref_maximum_path_c()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box and no prebuilt core)")
    B, TX, TY = shape
    rng = np.random.default_rng(TX + TY)
    tx = rng.integers(TY + 1, TX + 1, B)
    ty = rng.integers(max(1, TY // 2), TY + 1, B)
    tx[0], ty[0] = TX, TY
    mask = (mas.sequence_mask(tx, TX)[:, :, None] & mas.sequence_mask(ty, TY)[:, None, :]).astype(np.float32)
    v = rng.standard_normal((B, TX, TY)).astype(np.float32)
    assert np.array_equal(mas.maximum_path(v, mask, "c"), mas.maximum_path(v, mask, "ref"))


def test_reference_cython_core_was_built():
    import os

    if not os.path.isdir("/root/reference"):
        pytest.skip("reference tree absent (GPU box)")
    assert mas.ref_maximum_path_c() is not None, "run `make -C oracle` to build oracle/_ref"


def test_sequence_mask_reference_test_vector():
    # tests/tts_tests/test_helpers.py:24-30
    lengths = np.array([4, 2, 5])
    m = mas.sequence_mask(lengths)
    assert m.shape == (3, 5) and m[0].sum() == 4 and m[1].sum() == 2 and m[2].sum() == 5


def test_generate_path_reference_structure():
    # tests/tts_tests/test_helpers.py:71-88: row t holds `dur[t]` ones starting at cumsum offset
    rng = np.random.default_rng(0)
    durations = rng.integers(1, 5, (3, 21))
    durations[2, 15:] = 0
    x_len = np.array([21, 21, 15])
    y_len = durations.sum(1)
    mask = (mas.sequence_mask(x_len, 21)[:, :, None] & mas.sequence_mask(y_len)[:, None, :]).astype(np.float32)
    path = mas.generate_path(durations.astype(np.float32), mask)
    for b in range(3):
        c = 0
        for t in range(21):
            assert (path[b, t, c : c + durations[b, t]] == 1).all()
            assert path[b, t].sum() == durations[b, t]
            c += durations[b, t]
