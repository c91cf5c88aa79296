from . import label_sets  # noqa: F401
