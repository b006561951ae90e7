from .ep import EfficientProbing  # noqa: F401
