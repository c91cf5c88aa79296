from wav2letter_pytorch_amd.data.label_sets import *  # noqa: F401,F403
from wav2letter_pytorch_amd.data.label_sets import labels_map  # noqa: F401
