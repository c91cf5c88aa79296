from . import label_sets  # noqa: F401
from . import data_loader  # noqa: F401
from . import augmentations  # noqa: F401
