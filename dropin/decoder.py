"""Alias so `_target_: decoder.GreedyDecoder` (configuration/config.yaml:15) resolves to the MI355X decoder."""
from wav2letter_pytorch_amd.decoder import Decoder, GreedyDecoder  # noqa: F401
from wav2letter_pytorch_amd.beam_search import (PrefixBeamSearchLMDecoder, get_time_per_word,  # noqa: F401,E402
                                                prefix_beam_search)
