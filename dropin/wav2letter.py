"""Alias so an unmodified train.py (`from wav2letter import Wav2Letter`, train.py:12) gets the MI355X path."""
from wav2letter_pytorch_amd.wav2letter import Conv1dBlock, Wav2Letter  # noqa: F401
