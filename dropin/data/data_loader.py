from wav2letter_pytorch_amd.data.data_loader import *  # noqa: F401,F403
from wav2letter_pytorch_amd.data.data_loader import (BatchAudioDataLoader, SpectrogramDataset, SpectrogramExtractor,  # noqa: F401
                                                     _collator, load_audio)
