"""TEST INFRASTRUCTURE ONLY — re-export of the seeded synthetic-checkpoint factory (tts_amd/synthetic.py) under
the name the tests use."""
from tts_amd.synthetic import (HIFIGAN_V1, HIFIGAN_V2, _F, _dds, _flows, _transformer, _wn,  # noqa: F401
                               make_glow_state, make_hifi_decoder_state, make_hifigan_state, make_vits_state)
