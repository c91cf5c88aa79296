from wav2letter_pytorch_amd.data.augmentations import Identity, SpecAugment, SpecCutout  # noqa: F401
