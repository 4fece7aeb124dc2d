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
