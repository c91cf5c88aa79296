from wav2letter_pytorch_amd.data import label_sets  # noqa: F401
from . import data_loader  # noqa: F401
