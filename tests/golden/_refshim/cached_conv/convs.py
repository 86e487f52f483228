from . import get_padding  # noqa: F401  (base.gin imports `from cached_conv import convs`)
