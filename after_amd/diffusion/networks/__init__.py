from .transformerv2 import DenoiserV2  # noqa: F401
