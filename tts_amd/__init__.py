"""tts_amd — MI355X-native (gfx950) inference hot path of coqui-ai/TTS: VITS / Glow-TTS + HiFiGAN.

Python host code (mirroring the reference's plug-in surface) over hand-written HIP kernels
reached through the C ABI of include/tts_amd.h.  No CPU or PyTorch-op fallback exists.
"""
__version__ = "0.1.0"
