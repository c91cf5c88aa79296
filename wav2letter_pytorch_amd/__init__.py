"""MI355X-native Wav2Letter / Jasper CTC training path (drop-in for the model side of
assafmu/wav2letter_pytorch).  Importing this package loads libw2l_hip.so; there is no CPU
fallback."""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)
from .base_asr_models import ConvCTCASR  # noqa: F401
from .ctc_loss import CTCLoss  # noqa: F401
from .decoder import Decoder, GreedyDecoder  # noqa: F401
from .wav2letter import Conv1dBlock, Wav2Letter  # noqa: F401
from .jasper import Jasper, JasperBlock, MaskedConv1d  # noqa: F401

__all__ = ['ConvCTCASR', 'CTCLoss', 'Decoder', 'GreedyDecoder', 'Conv1dBlock', 'Wav2Letter', 'Jasper', 'JasperBlock', 'MaskedConv1d']
