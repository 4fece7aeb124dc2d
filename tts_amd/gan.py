"""GAN vocoder wrapper — drop-in for the inference surface of `TTS.vocoder.models.gan.GAN`
(gan.py:22-66 ctor/inference, :229-252 load_checkpoint, :371-374 init_from_config) with the generator built like
`setup_generator` (TTS/vocoder/models/__init__.py:34-94): `HifiganGenerator(in_channels=audio.num_mels,
out_channels=1, **generator_model_params)`.  The discriminator is training-only and never built here
(the reference drops it at eval-load, gan.py:248)."""
import torch

from . import _lib
from .hifigan import HifiganGenerator
from .vits import _get

HIFIGAN_GENERATOR_DEFAULTS = dict(  # TTS/vocoder/configs/hifigan_config.py:95-104
    upsample_factors=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
    resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], resblock_type="1")


class GAN:
    def __init__(self, config, ap=None):
        self.config = config
        self.ap = ap
        gm = str(_get(config, "generator_model", "hifigan_generator")).lower()
        if gm != "hifigan_generator":
            raise _lib.TtsAmdError("tts_amd.GAN: generator_model %r has no HIP path (hifigan_generator only)" % gm)
        params = dict(HIFIGAN_GENERATOR_DEFAULTS)
        gp = _get(config, "generator_model_params", None) or {}
        params.update(gp if isinstance(gp, dict) else vars(gp))
        audio = _get(config, "audio", None)
        num_mels = _get(audio, "num_mels", 80)
        self.model_g = HifiganGenerator(in_channels=num_mels, out_channels=1, **params)
        self.model_d = None
        self.y_hat_g = None

    @staticmethod
    def init_from_config(config, verbose=True):
        return GAN(config, ap=_get(config, "_ap", None))

    def parameters(self):
        return self.model_g.parameters()

    def eval(self):
        return self

    def cuda(self, device=None):
        self.model_g.cuda(device)
        return self

    def to(self, device):
        self.model_g.to(device)
        return self

    def load_state_dict(self, sd, strict=True):
        """Accepts the full GAN state_dict (`model_g.*`, `model_d.*`) or a bare generator state_dict."""
        if any(k.startswith("model_g.") for k in sd):
            self.model_g.load_state_dict(sd, prefix="model_g.")
        else:
            self.model_g.load_state_dict(sd)

    def load_checkpoint(self, config, checkpoint_path, eval=False, cache=False):  # noqa: A002
        """gan.py:229-252: old checkpoints hold only the generator; eval drops D and strips weight-norm
        (folding happens when the weights are packed)."""
        state = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        self.load_state_dict(state["model"])

    @torch.no_grad()
    def inference(self, x):
        """gan.py:58-66: `model_g.inference(x)` — mel [B,C,T] -> wav [B,1,(T+2*pad)*hop] (replicate pad, no crop)."""
        return self.model_g.inference(x)

    def forward(self, x):
        return self.model_g.forward(x)

    __call__ = forward
