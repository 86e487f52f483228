from .networks import DenoiserV2, Encoder1D, ECAPATDNN, UNET1D  # noqa: F401
from .model import RectifiedFlow  # noqa: F401
