"""Pins oracle/tts_oracle.py (CPU): (i) against the REAL reference modules when /root/reference is
present (build container), (ii) against the committed fixtures tests/golden/*.npz generated from
those modules by tests/golden/make_golden.py (works on the GPU box too)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_shim
from tests.golden import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REL_TOL = 2e-5  # fp32 CPU restatement vs reference modules (different op grouping / other host CPU)


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_oracle_matches_golden_fixture(name):
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    out = cases.CASES[name]("oracle")
    assert set(gold.files) == set(out)
    if "durations" in out:  # integer path first (ceil cliff)
        assert np.array_equal(out["durations"].numpy(), gold["durations"])
    for k in gold.files:
        assert tuple(out[k].shape) == gold[k].shape, k
        assert _rel(out[k], torch.from_numpy(gold[k])) < REL_TOL, k


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree absent (GPU box)")
@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_oracle_matches_live_reference_modules(name):
    torch.set_num_threads(1)
    ref = cases.CASES[name]("ref")
    out = cases.CASES[name]("oracle")
    for k in ref:
        assert _rel(out[k], ref[k]) < 1e-6, k


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree absent (GPU box)")
def test_weight_factory_layout_matches_reference_state_dicts():
    """Seeded factories produce exactly the reference modules' state_dict keys and shapes
    (load_state_dict(strict=True) inside oracle/ref_models.py would raise otherwise)."""
    from oracle import ref_models as RM
    from oracle import weights as W

    RM.RefVits(W.make_vits_state(dict(upsample_initial_channel_decoder=32)), dict(upsample_initial_channel_decoder=32))
    RM.RefGlow(W.make_glow_state(dict(num_flow_blocks_dec=2)), dict(num_flow_blocks_dec=2))


def _reference_handle_chunks():
    """The body of Xtts.handle_chunks compiled straight from the reference file (the class itself cannot be imported
    here: GPT-2/tokenizer dependencies)."""
    import ast

    src = open(os.path.join(ref_shim.REF_ROOT, "TTS/tts/models/xtts.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "handle_chunks")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "xtts.handle_chunks", "exec"), ns)  # noqa: S102
    return lambda *a: ns["handle_chunks"](None, *a)


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree absent (GPU box)")
@pytest.mark.parametrize("overlap", [64, 1024])
def test_oracle_handle_chunks_matches_reference(overlap):
    from oracle import tts_oracle as O

    ref = _reference_handle_chunks()
    g = torch.Generator().manual_seed(overlap)
    lens = [5000, 9000, 9000 + overlap // 2, 14000, 14000]      # growing prefix, a short chunk, the final flush
    full = torch.randn(lens[-1] + 100, generator=g)
    pa, oa, pb, ob = None, None, None, None
    for n in lens:
        wa = (full[:n] + 0.01 * torch.randn(n, generator=g)).clone()    # the tail of the prefix changes as it grows
        wb = wa.clone()
        ca, pa, oa = ref(wa, pa, oa, overlap)
        cb, pb, ob = O.xtts_handle_chunks(wb, pb, ob, overlap)
        assert torch.equal(ca, cb) and torch.equal(pa, pb)
        assert (oa is None) == (ob is None) and (oa is None or torch.equal(oa, ob))


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree absent (GPU box)")
def test_oracle_forward_mas_matches_reference_method():
    """oracle.vits_forward_mas against the body of the reference's `Vits.forward_mas` (vits.py:909-942) compiled straight
    from the reference file (the class itself needs coqpit / trainer to import) with a stub `self` whose duration predictor
    returns zeros — the alignment half is all that is restated — and the reference's own `maximum_path`."""
    import ast
    import math
    import types

    from oracle import mas
    from oracle import tts_oracle as O

    src = open(os.path.join(ref_shim.REF_ROOT, "TTS/tts/models/vits.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "forward_mas")
    ref_helpers = ref_shim.ref("TTS.tts.utils.helpers")
    ns = {"torch": torch, "math": math, "maximum_path": ref_helpers.maximum_path}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "vits.forward_mas", "exec"), ns)  # noqa: S102
    stub = types.SimpleNamespace(args=types.SimpleNamespace(use_sdp=True, detach_dp_input=True),
                                 duration_predictor=lambda x, x_mask, *a, **k: torch.zeros(x.shape[0]))
    g = torch.Generator().manual_seed(4)
    B, C, Tx, Ty = 3, 16, 11, 37
    z_p = torch.randn(B, C, Ty, generator=g)
    m_p = torch.randn(B, C, Tx, generator=g)
    logs_p = 0.3 * torch.randn(B, C, Tx, generator=g)
    x_mask = O.sequence_mask(torch.tensor([11, 7, 3]), Tx).unsqueeze(1).float()
    y_mask = O.sequence_mask(torch.tensor([37, 20, 9]), Ty).unsqueeze(1).float()
    _, attn = ns["forward_mas"](stub, {}, z_p, m_p, logs_p, torch.zeros(B, C, Tx), x_mask, y_mask, None, None)

    def cpu_mas(value, mask):
        return torch.from_numpy(mas.maximum_path(value.numpy(), mask.numpy(), "c")).to(value.dtype)

    out = O.vits_forward_mas(z_p, m_p, logs_p, x_mask, y_mask, cpu_mas)
    assert torch.equal(out["attn"], attn)
    assert torch.equal(out["attn_durations"], attn.sum(3))


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree absent (GPU box)")
def test_oracle_interpolate_vocoder_input_matches_reference():
    """oracle.interpolate_vocoder_input against the reference function compiled straight from
    TTS/vocoder/utils/generic_utils.py:11-29 (the module itself imports librosa-dependent code)."""
    import ast
    import contextlib
    import io

    from oracle import tts_oracle as O

    src = open(os.path.join(ref_shim.REF_ROOT, "TTS/vocoder/utils/generic_utils.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "interpolate_vocoder_input")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "generic_utils.interpolate_vocoder_input", "exec"), ns)  # noqa: S102
    rng = np.random.default_rng(0)
    for t, ratio in ((37, 24000 / 22050), (120, 16000 / 22050), (5, 44100 / 22050)):
        spec = rng.standard_normal((80, t)).astype(np.float32)
        with contextlib.redirect_stdout(io.StringIO()):
            want = ns["interpolate_vocoder_input"]([1, ratio], spec)
        got = O.interpolate_vocoder_input([1, ratio], spec)
        assert got.shape == want.shape and torch.equal(got, want)
