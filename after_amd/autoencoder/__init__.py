from .model import AutoEncoder  # noqa: F401
