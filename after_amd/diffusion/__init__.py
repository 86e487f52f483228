from .networks import DenoiserV2, Encoder1D, ECAPATDNN  # noqa: F401
from .model import RectifiedFlow  # noqa: F401
