"""TEST INFRASTRUCTURE ONLY (build container only) — the reference's real leaf modules, built via
oracle/ref_shim.py and loaded with the seeded state_dicts of oracle/weights.py, plus the ~40-line
re-statements of the glue that lives in un-importable classes (`Vits.inference` vits.py:1112-1173,
`GlowTTS.inference` glow_tts.py:341-374) composed over those REAL modules.  Used to pin
oracle/tts_oracle.py and to generate tests/golden/*.npz.  Needs /root/reference.
"""
import torch

from . import ref_shim
from .tts_oracle import GLOW_DEFAULTS, VITS_DEFAULTS


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def hifigan(sd, cfg, in_channels, prefix="", pre_wn=True, post_wn=True, post_bias=True, cond_channels=0):
    hg = ref_shim.ref("TTS.vocoder.models.hifigan_generator")
    m = hg.HifiganGenerator(in_channels, 1, cfg["resblock_type"], cfg["resblock_dilation_sizes"],
                            cfg["resblock_kernel_sizes"], cfg["upsample_kernel_sizes"],
                            cfg["upsample_initial_channel"], cfg["upsample_factors"],
                            inference_padding=cfg.get("inference_padding", 5), cond_channels=cond_channels,
                            conv_pre_weight_norm=pre_wn,
                            conv_post_weight_norm=post_wn, conv_post_bias=post_bias)
    m.load_state_dict(_sub(sd, prefix), strict=True)
    return m.eval()


class RefVits:
    """Real reference sub-modules wired as in Vits.__init__ (vits.py:653-718)."""

    def __init__(self, sd, args=None):
        a = dict(VITS_DEFAULTS)
        a.update(args or {})
        self.a = a
        nw = ref_shim.ref("TTS.tts.layers.vits.networks")
        sdp = ref_shim.ref("TTS.tts.layers.vits.stochastic_duration_predictor")
        dp = ref_shim.ref("TTS.tts.layers.glow_tts.duration_predictor")
        h = a["hidden_channels"]
        spk = int(a.get("embedded_speaker_dim", 0) or 0)
        lng = int(a.get("embedded_language_dim", 4)) if a.get("use_language_embedding", False) else 0
        self.emb_l = sd.get("emb_l.weight")
        self.text_encoder = nw.TextEncoder(a["num_chars"], h, h, a["hidden_channels_ffn_text_encoder"],
                                           a["num_heads_text_encoder"], a["num_layers_text_encoder"],
                                           a["kernel_size_text_encoder"], 0.1, language_emb_dim=lng)
        self.flow = nw.ResidualCouplingBlocks(h, h, kernel_size=a["kernel_size_flow"],
                                              dilation_rate=a["dilation_rate_flow"], num_layers=a["num_layers_flow"],
                                              cond_channels=spk)
        if a["use_sdp"]:
            self.duration_predictor = sdp.StochasticDurationPredictor(h, 192, 3, 0.5, 4, cond_channels=spk,
                                                                      language_emb_dim=lng)
        else:
            self.duration_predictor = dp.DurationPredictor(h, 256, 3, 0.5, cond_channels=spk, language_emb_dim=lng)
        self.emb_g = sd.get("emb_g.weight")
        self.posterior_encoder = None
        if any(k.startswith("posterior_encoder.") for k in sd):
            self.posterior_encoder = nw.PosteriorEncoder(a.get("out_channels", 513), h, h,
                                                         kernel_size=a.get("kernel_size_posterior_encoder", 5),
                                                         dilation_rate=a.get("dilation_rate_posterior_encoder", 1),
                                                         num_layers=a.get("num_layers_posterior_encoder", 16),
                                                         cond_channels=spk)
            self.posterior_encoder.load_state_dict(_sub(sd, "posterior_encoder."), strict=True)
            self.posterior_encoder.eval()
        self.text_encoder.load_state_dict(_sub(sd, "text_encoder."), strict=True)
        self.flow.load_state_dict(_sub(sd, "flow."), strict=True)
        self.duration_predictor.load_state_dict(_sub(sd, "duration_predictor."), strict=True)
        for m in (self.text_encoder, self.flow, self.duration_predictor):
            m.eval()
        self.waveform_decoder = None
        if any(k.startswith("waveform_decoder.") for k in sd):
            cfg = dict(resblock_type=a["resblock_type_decoder"],
                       resblock_dilation_sizes=a["resblock_dilation_sizes_decoder"],
                       resblock_kernel_sizes=a["resblock_kernel_sizes_decoder"],
                       upsample_kernel_sizes=a["upsample_kernel_sizes_decoder"],
                       upsample_initial_channel=a["upsample_initial_channel_decoder"],
                       upsample_factors=a["upsample_rates_decoder"], inference_padding=0)
            self.waveform_decoder = hifigan(sd, cfg, h, "waveform_decoder.", False, False, False, cond_channels=spk)

    @torch.no_grad()
    def inference(self, x, x_lengths, seed, speaker_ids=None, d_vectors=None, language_ids=None):
        """vits.py:1112-1173 over the real modules; RNG draws happen inside the reference modules
        (stochastic_duration_predictor.py:287, vits.py:1155) under torch.manual_seed(seed)."""
        helpers = ref_shim.ref("TTS.tts.utils.helpers")
        a = self.a
        torch.manual_seed(seed)
        g = None                                                     # vits.py:873-886,1116-1117
        if d_vectors is not None:
            g = torch.nn.functional.normalize(d_vectors).unsqueeze(-1)
        elif speaker_ids is not None:
            g = torch.nn.functional.embedding(speaker_ids, self.emb_g).unsqueeze(-1)
        lang_emb = None                                              # vits.py:1119-1122
        if language_ids is not None:
            lang_emb = torch.nn.functional.embedding(language_ids, self.emb_l).unsqueeze(-1)
        x, m_p, logs_p, x_mask = self.text_encoder(x, x_lengths, lang_emb=lang_emb)
        if a["use_sdp"]:
            logw = self.duration_predictor(x, x_mask, g=g, reverse=True, noise_scale=a["inference_noise_scale_dp"],
                                           lang_emb=lang_emb)
        else:
            logw = self.duration_predictor(x, x_mask, g=g, lang_emb=lang_emb)
        w = torch.exp(logw) * x_mask * a["length_scale"]
        w_ceil = torch.ceil(w)
        y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
        y_mask = helpers.sequence_mask(y_lengths, None).to(x_mask.dtype).unsqueeze(1)
        attn_mask = x_mask * y_mask.transpose(1, 2)
        attn = helpers.generate_path(w_ceil.squeeze(1), attn_mask.squeeze(1).transpose(1, 2))
        m_p = torch.matmul(attn.transpose(1, 2), m_p.transpose(1, 2)).transpose(1, 2)
        logs_p = torch.matmul(attn.transpose(1, 2), logs_p.transpose(1, 2)).transpose(1, 2)
        z_p = m_p + torch.randn_like(m_p) * torch.exp(logs_p) * a["inference_noise_scale"]
        z = self.flow(z_p, y_mask, g=g, reverse=True)
        o = self.waveform_decoder((z * y_mask)[:, :, : a["max_inference_len"]], g=g)
        return {"model_outputs": o, "alignments": attn, "durations": w_ceil, "z": z, "z_p": z_p, "m_p": m_p,
                "logs_p": logs_p, "y_mask": y_mask, "logw": logw, "x": x}


    @torch.no_grad()
    def voice_conversion(self, y, y_lengths, g_src, g_tgt, seed):
        """vits.py:1220-1228 over the real modules (the randn_like draw happens inside PosteriorEncoder)."""
        torch.manual_seed(seed)
        z, _, _, y_mask = self.posterior_encoder(y, y_lengths, g=g_src)
        z_p = self.flow(z, y_mask, g=g_src)
        z_hat = self.flow(z_p, y_mask, g=g_tgt, reverse=True)
        o_hat = self.waveform_decoder(z_hat * y_mask, g=g_tgt)
        return {"model_outputs": o_hat, "z": z, "z_p": z_p, "z_hat": z_hat}


class RefGlow:
    """Real reference Encoder/Decoder wired as in GlowTTS.__init__ (glow_tts.py:80-105)."""

    def __init__(self, sd, args=None):
        a = dict(GLOW_DEFAULTS)
        a.update(args or {})
        self.a = a
        enc = ref_shim.ref("TTS.tts.layers.glow_tts.encoder")
        dec = ref_shim.ref("TTS.tts.layers.glow_tts.decoder")
        ep = dict(a["encoder_params"])
        ep.setdefault("dropout_p", 0.1)
        cin = int(a.get("c_in_channels", 0) or 0)
        self.emb_g = sd.get("emb_g.weight")
        self.encoder = enc.Encoder(a["num_chars"], out_channels=a["out_channels"],
                                   hidden_channels=a["hidden_channels_enc"], hidden_channels_dp=a["hidden_channels_dp"],
                                   encoder_type="rel_pos_transformer", encoder_params=ep, mean_only=a["mean_only"],
                                   use_prenet=a["use_encoder_prenet"], dropout_p_dp=0.1, c_in_channels=cin)
        self.decoder = dec.Decoder(a["out_channels"], a["hidden_channels_dec"], a["kernel_size_dec"],
                                   a["dilation_rate"], a["num_flow_blocks_dec"], a["num_block_layers"], dropout_p=0.05,
                                   num_splits=a["num_splits"], num_squeeze=a["num_squeeze"],
                                   sigmoid_scale=a["sigmoid_scale"], c_in_channels=cin)
        self.encoder.load_state_dict(_sub(sd, "encoder."), strict=True)
        self.decoder.load_state_dict(_sub(sd, "decoder."), strict=True)
        self.encoder.eval()
        self.decoder.eval()
        self.decoder.store_inverse()  # GlowTTS.load_checkpoint(eval=True), glow_tts.py:522-530

    @torch.no_grad()
    def inference(self, x, x_lengths, seed, speaker_ids=None, d_vectors=None, language_ids=None):
        helpers = ref_shim.ref("TTS.tts.utils.helpers")
        a = self.a
        torch.manual_seed(seed)
        g = None                                                     # glow_tts.py:179-191
        if speaker_ids is not None:
            g = torch.nn.functional.normalize(torch.nn.functional.embedding(speaker_ids, self.emb_g)).unsqueeze(-1)
        elif d_vectors is not None:
            g = torch.nn.functional.normalize(d_vectors).unsqueeze(-1)
        o_mean, o_log_scale, o_dur_log, x_mask = self.encoder(x, x_lengths, g=g)
        w = (torch.exp(o_dur_log) - 1) * x_mask * a["length_scale"]
        w_ceil = torch.clamp_min(torch.ceil(w), 1)
        y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
        y_mask = torch.unsqueeze(helpers.sequence_mask(y_lengths, None), 1).to(x_mask.dtype)
        attn_mask = torch.unsqueeze(x_mask, -1) * torch.unsqueeze(y_mask, 2)
        attn = helpers.generate_path(w_ceil.squeeze(1), attn_mask.squeeze(1)).unsqueeze(1)
        y_mean = torch.matmul(attn.squeeze(1).transpose(1, 2), o_mean.transpose(1, 2)).transpose(1, 2)
        y_log_scale = torch.matmul(attn.squeeze(1).transpose(1, 2), o_log_scale.transpose(1, 2)).transpose(1, 2)
        z = (y_mean + torch.exp(y_log_scale) * torch.randn_like(y_mean) * a["inference_noise_scale"]) * y_mask
        y, _ = self.decoder(z, y_mask, g=g, reverse=True)
        return {"model_outputs": y.transpose(1, 2), "durations": w_ceil, "durations_log": o_dur_log.transpose(1, 2),
                "y_mean": y_mean.transpose(1, 2)}
