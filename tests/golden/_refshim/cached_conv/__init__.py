"""Stand-in for cached-conv>=2.5.0 in its NON-cached (offline) mode, used ONLY
by tests/golden/make_golden.py.  Written from the published behaviour of
cached_conv (padding arithmetic + thin wrappers around torch convs); the
streaming caches are not modelled.  `_PADDING_MODE` emulates the scoped gin
binding `encoder_time/convs.get_padding.mode = 'causal'` (base.gin:55)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

_PADDING_MODE = "centered"
USE_BUFFER_CONV = False


def use_cached_conv(state: bool):
    global USE_BUFFER_CONV
    USE_BUFFER_CONV = bool(state)
    if state:
        raise RuntimeError("cached_conv shim: streaming mode not modelled")


def set_padding_mode(mode: str):
    global _PADDING_MODE
    assert mode in ("centered", "causal")
    _PADDING_MODE = mode


def get_padding(kernel_size, stride=1, dilation=1, mode=None):
    mode = _PADDING_MODE if mode is None else mode
    if kernel_size == 1:
        return (0, 0)
    p = (kernel_size - 1) * dilation + 1
    if mode == "centered":
        return ((p - 1) // 2, p // 2)
    elif mode == "causal":
        return (p // 2 + (p - 1) // 2, 0)
    raise ValueError(mode)


class Conv1d(nn.Conv1d):

    def __init__(self, *args, **kwargs):
        self._pad = kwargs.get("padding", (0, 0))
        self.cumulative_delay = 0
        kwargs.pop("cumulative_delay", None)
        kwargs["padding"] = 0
        super().__init__(*args, **kwargs)
        if isinstance(self._pad, int):
            self._pad = (self._pad, self._pad)

    def forward(self, x):
        x = F.pad(x, self._pad)
        return F.conv1d(x, self.weight, self.bias, self.stride, 0,
                        self.dilation, self.groups)


class ConvTranspose1d(nn.ConvTranspose1d):

    def __init__(self, *args, **kwargs):
        kwargs.pop("cumulative_delay", None)
        super().__init__(*args, **kwargs)
        self.cumulative_delay = 0


class CachedSequential(nn.Sequential):

    def __init__(self, *args, **kwargs):
        super().__init__(*args)
        self.cumulative_delay = 0


class AlignBranches(nn.Module):

    def __init__(self, *branches, delays=None, cumulative_delay=0, stride=1):
        super().__init__()
        self.branches = nn.ModuleList(branches)
        self.cumulative_delay = 0

    def forward(self, x):
        return [b(x) for b in self.branches]
