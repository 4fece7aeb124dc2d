"""GPU parity of the model-level C ABI of the two acoustic models (include/tts_amd.h: ttsamd_vits_*, ttsamd_glowtts_*;
csrc/vits_model.hip, csrc/glow_model.hip) through ctypes (tts_amd/native.py):

  * BITWISE equal to the Python-driven path (tts_amd.Vits / tts_amd.GlowTTS over the kernel-level ABI) when both get the same folded
    weights — the handle issues the same kernel-level calls with the same arguments;
  * against the CPU oracle (oracle/tts_oracle.py, pinned to the reference modules) at the usual tolerances with the RAW
    reference-layout state_dict (weight norm folded in C++, its norm summed in a different order than torch's);
  * the front end's hipGraph replay == its eager run; errors come back as codes (nothing crashes)."""
import pytest
import torch

from oracle import tts_oracle as O
from oracle import weights as W
from tts_amd import _lib, ops
from tts_amd.glow_tts import GlowTTS
from tts_amd.native import NativeGlowTTS, NativeVits
from tts_amd.vits import Vits

pytestmark = pytest.mark.gpu


def _errs(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    rms = float((a - b).pow(2).mean().sqrt())
    return rms, rms / float(b.pow(2).mean().sqrt() + 1e-30)


def _folded(sd):
    out = {}
    for k in sd:
        if k.endswith(".parametrizations.weight.original0"):
            name = k[: -len(".parametrizations.weight.original0")]
            out[name + ".weight"] = ops.fold_weight_norm(sd, name)
        elif not k.endswith(".parametrizations.weight.original1"):
            out[k] = sd[k]
    return out


@pytest.mark.parametrize("variant", ["sdp_small", "dp_small", "sdp_c256"])
def test_native_vits_handle_equals_the_python_driven_path(gpu, variant):
    torch.set_num_threads(8)
    args = dict(upsample_initial_channel_decoder=256 if variant == "sdp_c256" else 64, use_sdp=variant != "dp_small")
    sd = W.make_vits_state(args, seed=91)
    g = torch.Generator().manual_seed(5)
    B, T = (2, 57) if variant == "sdp_c256" else (3, 41)
    x = torch.randint(0, 100, (B, T), generator=g)
    xl = torch.tensor([T, 29, 12][:B])
    noise_dp = torch.randn(B, 2, T, generator=g)
    m = Vits({"model_args": args})
    m.load_state_dict(_folded(sd))
    m.to(gpu)
    m.use_graphs = False
    m.use_native = False
    m.waveform_decoder.use_graphs, m.waveform_decoder.concurrent_branches = False, False
    xg, xlg, ndg = x.to(gpu), xl.to(gpu), noise_dp.to(gpu)
    pre = m.inference(xg, {"x_lengths": xlg, "noise_dp": ndg, "return_extras": True})
    t_dec = pre["z"].shape[2]
    noise_z = torch.randn(B, 192, t_dec, generator=g).to(gpu)
    ref = m.inference(xg, {"x_lengths": xlg, "noise_dp": ndg, "noise_z": noise_z, "return_extras": True})
    nat = NativeVits(m, _folded(sd))
    td, ylens = nat.encode(xg, xlg, ndg)
    assert td == t_dec and ylens == [int(v) for v in ref["y_lengths"].cpu()]
    got = nat.decode(td, noise_z, extras=True)
    for k in ("model_outputs", "alignments", "durations", "z", "z_p", "m_p", "logs_p", "y_mask", "y_lengths", "x", "logw"):
        assert got[k].shape == ref[k].shape, (k, got[k].shape, ref[k].shape)
        assert torch.equal(got[k], ref[k]), (k, float((got[k].double() - ref[k].double()).abs().max()))
    # the front end as a hipGraph: first call eager + capture, then replays
    for _ in range(3):
        td2, _ = nat.encode(xg, xlg, ndg, use_graph=True)
        again = nat.decode(td2, noise_z)
        assert td2 == t_dec and torch.equal(again["model_outputs"], ref["model_outputs"]) and torch.equal(again["durations"], ref["durations"])
    # injected durations (vits.py:1141-1143), with and without the predictor in the pass
    dur = ref["durations"].clone()
    ref_inj = m.inference(xg, {"x_lengths": xlg, "noise_dp": ndg, "noise_z": noise_z, "durations": dur, "run_duration_predictor": True})
    for run_dp in (True, False):
        td3, _ = nat.encode(xg, xlg, ndg if run_dp else None, durations=dur, run_duration_predictor=run_dp)
        assert torch.equal(nat.decode(td3, noise_z)["model_outputs"], ref_inj["model_outputs"])
    nat.close()
    # the raw reference-layout state_dict (the handle folds the weight norm itself) against the oracle; the oracle's integer
    # durations are injected (ceil() cliff), the duration predictor still runs and is compared through logw
    pre0 = O.vits_inference(sd, x, xl, args, noise_dp=noise_dp, stop_after="prior", noise_z=torch.zeros(B, 192, 1))
    t_o = int(pre0["y_lengths"].max())
    nz_o = torch.randn(B, 192, t_o, generator=g)
    want = O.vits_inference(sd, x, xl, args, noise_dp=noise_dp, noise_z=nz_o)
    nat = NativeVits(m, sd)
    td, _ = nat.encode(xg, xlg, ndg, durations=want["durations"].to(gpu), run_duration_predictor=True)
    assert td == t_o
    got = nat.decode(td, nz_o.to(gpu), extras=True)
    rms, rel = _errs(got["model_outputs"], want["model_outputs"])
    assert rms < 1e-4 and rel < 1e-5, (rms, rel)
    assert _errs(got["logw"], want["logw"])[1] < 1e-5
    for k in ("z", "z_p", "m_p", "logs_p"):
        assert _errs(got[k], want[k])[1] < 1e-5, k
    assert torch.equal(got["alignments"].cpu(), want["alignments"]) and torch.equal(got["y_mask"].cpu(), want["y_mask"])
    # errors are codes + messages
    with pytest.raises(_lib.TtsAmdError):
        NativeVits(m, {k: v for k, v in sd.items() if not k.startswith("flow.flows.2.")})
    if args["use_sdp"]:                                # the SDP needs its noise draw: a NULL pointer is an error code, not a crash
        import ctypes

        td_c = ctypes.c_int32(0)
        rc = _lib.lib().ttsamd_vits_encode(nat._h, _lib.P(xg), _lib.P(xlg), B, T, None, None, 0, None, ctypes.byref(td_c), 0, _lib.stream_ptr())
        assert rc == -1 and b"noise_dp" in _lib.lib().ttsamd_last_error()
    nat.close()
    nat.close()


@pytest.mark.parametrize("c0", [64, 256])
def test_native_vits_single_request_replays_its_tail_as_a_graph(gpu, c0):
    """A single request through the handle with use_graph: front end and tail (prior expansion, flows, waveform decoder at the
    32-frame bucket, ragged-exact) replay as hipGraphs, outputs cut to the true extent — bitwise the Python host's graphed request
    (tts_amd.Vits with `_front` / `_tail` captured, text bucket off), over several requests of different noise and two lengths that
    share a bucket."""
    args = dict(upsample_initial_channel_decoder=c0)
    sd = _folded(W.make_vits_state(args, seed=17))
    g = torch.Generator().manual_seed(9)
    m = Vits({"model_args": args})
    m.load_state_dict(sd)
    m.to(gpu)
    m.use_native, m.text_bucket = False, 1
    nat = NativeVits(m, sd)
    for T in (33, 33, 35, 33):
        x = torch.randint(0, 100, (1, T), generator=g).to(gpu)
        nd = torch.randn(1, 2, T, generator=g).to(gpu)
        pre = m.inference(x, {"noise_dp": nd, "return_extras": True})
        t_dec = pre["z"].shape[2]
        nz = torch.randn(1, 192, t_dec, generator=g).to(gpu)
        ref = m.inference(x, {"noise_dp": nd, "noise_z": nz, "return_extras": True})
        td, _ = nat.encode(x, None, nd, use_graph=True)
        assert td == t_dec
        got = nat.decode(td, nz, extras=True, use_graph=True)
        for k in ("model_outputs", "alignments", "durations", "z", "z_p", "m_p", "logs_p", "y_mask", "y_lengths", "x", "logw"):
            assert got[k].shape == ref[k].shape, (k, got[k].shape, ref[k].shape)
            assert torch.equal(got[k], ref[k]), (T, k, float((got[k].double() - ref[k].double()).abs().max()))
    nat.close()
    # the same through the product class: `Vits.inference` routed to the handle (inputs staged into per-stream buffers, token axis
    # padded to the 16-token bucket, outputs cut in the handle's copy-out) against the Python host's graphed request, a batch too
    m.text_bucket = 16
    for B, T in ((1, 33), (1, 37), (1, 33), (3, 37), (3, 37)):
        x = torch.randint(0, 100, (B, T), generator=g).to(gpu)
        xl = torch.tensor([T, T - 5, T - 11][:B]).to(gpu)
        nd = torch.randn(B, 2, T, generator=g).to(gpu)
        m.use_native = False
        pre = m.inference(x, {"x_lengths": xl, "noise_dp": nd, "return_extras": True})
        nz = torch.randn(B, 192, pre["z"].shape[2], generator=g).to(gpu)
        aux = {"x_lengths": xl, "noise_dp": nd, "noise_z": nz, "return_extras": True}
        ref = m.inference(x, aux)
        m.use_native, m.native_single_requests = True, True
        got = m.inference(x, aux)
        m.use_native = False
        for k in ("model_outputs", "alignments", "durations", "z", "z_p", "m_p", "logs_p", "y_mask", "y_lengths", "x", "logw"):
            assert got[k].shape == ref[k].shape, (B, T, k, got[k].shape, ref[k].shape)
            assert torch.equal(got[k], ref[k]), (B, T, k, float((got[k].double() - ref[k].double()).abs().max()))
    assert len(m._native) == 1


@pytest.mark.parametrize("variant", ["default", "relwin", "not_mean_only"])
def test_native_glowtts_handle_equals_the_python_driven_path(gpu, variant):
    torch.set_num_threads(8)
    args = dict(num_flow_blocks_dec=3, inference_noise_scale=0.4)
    args["encoder_params"] = dict(O.GLOW_DEFAULTS["encoder_params"], num_layers=2)
    if variant == "relwin":
        args["encoder_params"].update(rel_attn_window_size=4, layer_norm_type="2")
    if variant == "not_mean_only":
        args["mean_only"] = False
    sd = W.make_glow_state(dict(args, num_chars=130), seed=33)
    # store_inverse(): the 4 x 4 inverses as the checkpoint of an eval-loaded model carries them — both hosts then read the same matrix
    sdf = _folded(sd)
    for k in list(sdf):
        if k.startswith("decoder.flows.") and k.endswith(".weight") and sdf[k].dim() == 2 and sdf[k].shape == (4, 4):
            sdf[k[: -len("weight")] + "weight_inv"] = torch.inverse(sdf[k].float())
    g = torch.Generator().manual_seed(3)
    B, T = 3, 37
    x = torch.randint(0, 130, (B, T), generator=g)
    xl = torch.tensor([37, 30, 11])
    m = GlowTTS(dict(args, num_chars=130))
    m.load_state_dict(sdf)
    m.to(gpu)
    m.use_graphs = False
    m.use_native = False
    xg, xlg = x.to(gpu), xl.to(gpu)
    pre = m.inference(xg, {"x_lengths": xlg})
    t_dec = int(pre["y_lengths"].max())
    noise = torch.randn(B, 80, t_dec, generator=g).to(gpu)
    ref = m.inference(xg, {"x_lengths": xlg, "noise": noise})
    nat = NativeGlowTTS(m, sdf)
    for use_graph in (False, True, True):
        td, ylens = nat.encode(xg, xlg, use_graph=use_graph)
        assert td == t_dec and ylens == [int(v) for v in ref["y_lengths"].cpu()]
        got = nat.decode(td, noise)
        for k in ("model_outputs", "y_mean", "y_log_scale", "alignments", "durations_log", "total_durations_log", "y_lengths", "durations"):
            assert got[k].shape == ref[k].shape, (k, got[k].shape, ref[k].shape)
            assert torch.equal(got[k], ref[k]), (k, float((got[k].double() - ref[k].double()).abs().max()))
    # ragged-exact batching (padded tokens own no frames)
    td, _ = nat.encode(xg, xlg, ragged_exact=True)
    nz_r = torch.randn(B, 80, td, generator=g).to(gpu)
    ref_r = m.inference(xg, {"x_lengths": xlg, "noise": nz_r, "ragged_exact": True})
    assert torch.equal(nat.decode(td, nz_r)["model_outputs"], ref_r["model_outputs"])
    nat.close()
    # raw state_dict (weight norm folded and the 4 x 4 matrices inverted in C++) against the oracle
    a2 = dict(args, num_chars=130)
    pre_o = O.glow_tts_inference(sd, x, xl, dict(a2, num_flow_blocks_dec=0, inference_noise_scale=0.0))
    t_o = int(pre_o["y_lengths"].max())
    nz_o = torch.randn(B, 80, t_o, generator=g)
    want = O.glow_tts_inference(sd, x, xl, a2, noise=nz_o)
    nat = NativeGlowTTS(m, sd)
    td, _ = nat.encode(xg, xlg, durations=want["durations"].to(gpu))          # ceil() cliff: the oracle's integer durations
    assert td == t_o
    got = nat.decode(td, nz_o.to(gpu))
    assert got["model_outputs"].shape == want["model_outputs"].shape
    assert _errs(got["model_outputs"], want["model_outputs"])[1] < 1e-5
    assert _errs(got["durations_log"], want["durations_log"])[1] < 1e-5
    assert torch.equal(got["alignments"].cpu(), want["alignments"])
    with pytest.raises(_lib.TtsAmdError):
        NativeGlowTTS(m, {k: v for k, v in sd.items() if "flows.5." not in k})
    nat.close()


def test_cxx_host_runs_vits_from_a_flat_weight_file_and_reproduces_the_reference_golden(gpu, tmp_path):
    """A host that is NOT Python (tests/native/vits_host.cpp, linked against libtts_amd.so and the HIP runtime only): loads the
    weights from a flat file, runs create / load / finalize / encode / decode through the model-level C ABI and reproduces the
    waveform of the committed fixture generated from the REAL reference modules (tests/golden/vits_small_sdp.npz) within
    1e-4 RMS / 1e-5 relative."""
    import ctypes
    import os
    import shutil
    import struct
    import subprocess

    import numpy as np

    from tests.golden import cases
    from tts_amd import native

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    gold = np.load(os.path.join(root, "tests", "golden", "vits_small_sdp.npz"))
    args = dict(cases.VITS_SMALL, use_sdp=True)
    sd = W.make_vits_state(args, seed=1234)
    x = torch.randint(0, 100, (3, 37), generator=torch.Generator().manual_seed(0))
    xl = torch.tensor([37, 30, 21])
    t_dec = gold["z_p"].shape[2]
    torch.manual_seed(7)                        # the reference draws randn(B,2,T) then randn_like(m_p) (tests/test_vits_gpu.py)
    noise_dp = torch.randn(3, 2, 37)
    noise_z = torch.randn_like(torch.empty(3, t_dec, 192).transpose(1, 2)).contiguous()
    m = Vits({"model_args": args})
    m.load_state_dict(sd)                       # (CPU object: only its arguments are read)
    cfg = native.vits_config(m, sd, precision="h2")
    wpath, cpath, exe = str(tmp_path / "weights.bin"), str(tmp_path / "case.bin"), str(tmp_path / "vits_host")
    native.save_flat_weights(sd, wpath)
    wav = torch.from_numpy(gold["model_outputs"]).float().contiguous()
    with open(cpath, "wb") as f:
        f.write(b"TTSAMDC1" + struct.pack("<I", ctypes.sizeof(cfg)) + bytes(cfg) + struct.pack("<ii", 3, 37))
        f.write(x.numpy().astype(np.int64).tobytes() + xl.numpy().astype(np.int64).tobytes() + noise_dp.numpy().tobytes())
        f.write(torch.from_numpy(gold["durations"]).float().reshape(3, 37).contiguous().numpy().tobytes())
        f.write(struct.pack("<i", t_dec) + noise_z.numpy().tobytes() + wav.numpy().tobytes() + struct.pack("<ff", 1e-4, 1e-5))
    libdir = os.path.join(root, "tts_amd")
    r = subprocess.run([hipcc, "-std=c++17", "-O2", os.path.join(root, "tests", "native", "vits_host.cpp"), "-I", os.path.join(root, "include"),
                        "-L", libdir, "-ltts_amd", "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, wpath, cpath], capture_output=True, text=True, timeout=240)
    print(r.stdout)
    assert r.returncode == 0 and "vits_host: ok" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
