"""Alias so `_target_: decoder.GreedyDecoder` (configuration/config.yaml:15) resolves to the MI355X decoder."""
from wav2letter_pytorch_amd.decoder import Decoder, GreedyDecoder  # noqa: F401
