"""Host-side mirror of the reference's sampler API
(after/diffusion/model.py:17-51 `Base`, :570-573 / :721-785 `RectifiedFlow`).

Only inference is built: `sample`, `model_forward`, the encoder attributes and
`load_state_dict`.  Training (`fit`, `cfgdrop`, ...) is out of scope."""
import torch
from torch import nn

from .. import _lib
from .networks.transformerv2 import DenoiserV2
from .networks.unet1d import UNET1D


class Base(nn.Module):
    """model.py:17-55: holds the networks; same constructor arguments."""

    def __init__(self,
                 net,
                 sr,
                 encoder=None,
                 encoder_time=None,
                 post_encoder=None,
                 classifier=None,
                 emb_model=None,
                 time_transform=None,
                 drop_value=-4.,
                 drop_rate=0.2,
                 device="cpu",
                 **kwargs):
        super().__init__()
        if not isinstance(net, (DenoiserV2, UNET1D)):
            raise TypeError("after_amd.RectifiedFlow needs an after_amd.DenoiserV2 (or UNET1D) as `net` "
                            f"(got {type(net).__name__}); there is no generic torch fallback")
        self.net = net
        self.encoder = encoder
        self.encoder_time = encoder_time
        self.post_encoder = post_encoder
        self.classifier = classifier
        self.time_transform = time_transform
        self.sr = sr
        self.drop_value = drop_value
        self.drop_rate = drop_rate
        self.extra_modules = nn.ModuleDict({})
        self.to(device)
        self.emb_model = emb_model

    @property
    def device(self):
        return next(self.parameters()).device

    def fit(self, *a, **k):
        raise NotImplementedError("training is out of scope of after_amd (inference path only)")


class RectifiedFlow(Base):
    """model.py:570-785.  `sample` runs the whole Euler loop inside libafter_hip
    (after_sample); `model_forward` is one CFG-combined velocity evaluation."""

    cfg_mode = _lib.CFG_API  # model.py:730-759; exports use CFG_EXPORT / CFG_MIDI

    def model_forward(self,
                      x: torch.Tensor,
                      time: torch.Tensor,
                      cond: torch.Tensor,
                      time_cond: torch.Tensor,
                      guidance_timbre: float,
                      guidance_structure: float,
                      cache_index: int = 0) -> torch.Tensor:
        """model.py:721-761.  Both networks evaluate the 3x CFG batch, the guidance combination (and,
        in `sample`, the Euler update) inside libafter_hip."""
        return self.net.cfg_forward(x, time, cond, time_cond, guidance_timbre, guidance_structure,
                                    self.drop_value, self.cfg_mode, cache_index)

    @torch.no_grad()
    def sample(self, x0, cond, time_cond, nb_steps, guidance_timbre=1., guidance_structure=1.):
        """model.py:763-785."""
        x0 = x0.to(self.device)
        return self.net.cfg_sample(x0, cond, time_cond, nb_steps, guidance_timbre, guidance_structure,
                                   self.drop_value, self.cfg_mode)
