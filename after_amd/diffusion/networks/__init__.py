from .transformerv2 import DenoiserV2  # noqa: F401
from .encoder import Encoder1D  # noqa: F401
from .ecapa_encoder import ECAPATDNN  # noqa: F401
from .unet1d import UNET1D  # noqa: F401
