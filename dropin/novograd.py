"""Alias of the reference module name ``novograd`` (optimizer `_target_: novograd.Novograd`)."""
from wav2letter_pytorch_amd.novograd import Novograd  # noqa: F401
