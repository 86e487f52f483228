"""after_amd: MI355X-native implementation of AFTER's latent-diffusion sampling
path (denoiser x N Euler steps + the RAVE-style autoencoder around it), behind
AFTER's own Python model/sampler API.  Compute runs in hand-written HIP kernels
(after_amd/csrc) reached through the C ABI of include/after_hip.h."""
from . import configs  # noqa: F401
from .diffusion import DenoiserV2, RectifiedFlow, Encoder1D, ECAPATDNN, UNET1D  # noqa: F401
from .autoencoder import AutoEncoder  # noqa: F401

from .streaming import Streamer, MidiStreamer  # noqa: F401

__all__ = ["configs", "DenoiserV2", "RectifiedFlow", "AutoEncoder", "Encoder1D", "ECAPATDNN",
           "Streamer", "MidiStreamer", "UNET1D"]
