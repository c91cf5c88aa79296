"""Alias so an unmodified train.py (`from jasper import Jasper`, train.py:13) gets the MI355X path."""
from wav2letter_pytorch_amd.jasper import (GroupShuffle, Jasper, JasperBlock, MaskedConv1d, compute_new_kernel_size,  # noqa: F401
                                          get_same_padding, init_weights, jasper_activations)
