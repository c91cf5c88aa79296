"""Data side of the training flow (reference: data/).  ``label_sets`` is plain data and imported eagerly; ``data_loader``
and ``augmentations`` call into libw2l_hip.so and are imported on first access, so that reading the default config tree
(defaults.py needs only the label tables) does not map the HIP library into a launch parent (launch.py)."""
import importlib

from . import label_sets  # noqa: F401

_LAZY = ('data_loader', 'augmentations', 'mel')


def __getattr__(name):
    if name in _LAZY:
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')
