"""GPU parity: HIP HiFiGAN generator vs the CPU oracle (oracle/tts_oracle.py, pinned to the
reference modules).  Tolerance: 1e-4 absolute RMS (north_star) AND 1e-5 relative RMS (internal bar)."""
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


@pytest.mark.parametrize("variant", ["v1_c64", "v2", "rb2", "v1_full"])
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
    else:
        T, B = 24, 1
    sd = O.make_hifigan_state(cfg, 80, seed=5)
    x = torch.randn(B, 80, T, generator=torch.Generator().manual_seed(0))
    want = O.hifigan_inference(sd, "", x, cfg)
    got = _make(cfg, 80, gpu, sd).inference(x.to(gpu))
    assert got.shape == want.shape == (B, 1, (T + 10) * 256)
    rms, rel = _errs(got, want)
    assert rms < 1e-4 and rel < 1e-5, (rms, rel)


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
