"""XTTS `HifiDecoder` (vocoder half of BASELINE config 5) on the HIP kernels — drop-in for the inference surface of
`TTS.tts.layers.xtts.hifigan_decoder.HifiDecoder` (hifigan_decoder.py:615-735): GPT latents [B,T,1024] -> linear
interpolation x(1024/256) -> x(24000/22050) -> HiFiGAN generator conditioned on the speaker d-vector `g` at the input
(`cond_layer`) AND after every upsampling layer (`conds[i]`).  The ResNet speaker encoder that produces `g` from
reference audio and the GPT-2 acoustic model are outside this build's scope: `g` is an input."""
import torch

from . import _lib, ops
from .hifigan import HifiganGenerator


class HifiDecoder:
    def __init__(self, input_sample_rate=22050, output_sample_rate=24000, output_hop_length=256,
                 ar_mel_length_compression=1024, decoder_input_dim=1024, resblock_type_decoder="1",
                 resblock_dilation_sizes_decoder=((1, 3, 5), (1, 3, 5), (1, 3, 5)), resblock_kernel_sizes_decoder=(3, 7, 11),
                 upsample_rates_decoder=(8, 8, 2, 2), upsample_initial_channel_decoder=512,
                 upsample_kernel_sizes_decoder=(16, 16, 4, 4), d_vector_dim=512,
                 cond_d_vector_in_each_upsampling_layer=True, speaker_encoder_audio_config=None):
        self.input_sample_rate, self.output_sample_rate = input_sample_rate, output_sample_rate
        self.output_hop_length, self.ar_mel_length_compression = output_hop_length, ar_mel_length_compression
        self.waveform_decoder = HifiganGenerator(
            decoder_input_dim, 1, resblock_type_decoder, resblock_dilation_sizes_decoder, resblock_kernel_sizes_decoder,
            upsample_kernel_sizes_decoder, upsample_initial_channel_decoder, upsample_rates_decoder, inference_padding=0,
            cond_channels=d_vector_dim, conv_pre_weight_norm=False, conv_post_weight_norm=False, conv_post_bias=False,
            cond_in_each_up_layer=cond_d_vector_in_each_upsampling_layer)
        self.speaker_encoder = None   # out of scope: pass `g`

    @property
    def device(self):
        return self.waveform_decoder.device

    def parameters(self):
        return self.waveform_decoder.parameters()

    def eval(self):
        return self

    def cuda(self, device=None):
        self.waveform_decoder.cuda(device)
        return self

    def to(self, device):
        self.waveform_decoder.to(device)
        return self

    def load_state_dict(self, sd, strict=True):
        self.waveform_decoder.load_state_dict(sd, prefix="waveform_decoder.")

    def load_checkpoint(self, checkpoint_path, eval=False):  # noqa: A002
        state = torch.load(checkpoint_path, map_location="cpu", weights_only=False)["model"]
        self.load_state_dict({k: v for k, v in state.items() if "waveform_decoder." in k})   # :722-728

    @torch.no_grad()
    def forward(self, latents, g=None):
        """latents [B, T, C] (GPT hidden states), g [B, d_vector_dim, 1] -> waveform [B, 1, T_wav]."""
        _lib.require_gpu(latents, "latents")
        z = latents.float().transpose(1, 2).contiguous()                     # [B, C, T]
        z = ops.linear_interp(z, self.ar_mel_length_compression / self.output_hop_length)
        if self.output_sample_rate != self.input_sample_rate:
            z = ops.linear_interp(z, self.output_sample_rate / self.input_sample_rate)
        return self.waveform_decoder.forward(z, g=g)

    inference = forward
    __call__ = forward
