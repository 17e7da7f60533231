from .base import CompressionModel  # noqa: F401
