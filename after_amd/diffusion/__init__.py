from .networks import DenoiserV2  # noqa: F401
from .model import RectifiedFlow  # noqa: F401
