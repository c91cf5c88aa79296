"""Alias of base_asr_models.ConvCTCASR."""
from wav2letter_pytorch_amd.base_asr_models import ConvCTCASR  # noqa: F401
