"""Host-side mirror of the reference's ECAPATDNN
(after/diffusion/networks/ecapa_encoder.py:458-666), used as `encoder`.

Parameter containers under the reference's state-dict keys; compute in libafter_hip
(after_ecapa_forward)."""
import ctypes

import torch
from torch import nn

from ... import _lib


class _ConvReflect(nn.Module):

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, k)


class _TDNN(nn.Module):

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = _ConvReflect(cin, cout, k)
        self.norm = nn.BatchNorm1d(cout)


class _Res2Net(nn.Module):

    def __init__(self, ch, scale, k):
        super().__init__()
        self.blocks = nn.ModuleList([_TDNN(ch // scale, ch // scale, k) for _ in range(scale - 1)])


class _SE(nn.Module):

    def __init__(self, cin, se, cout):
        super().__init__()
        self.conv1 = _ConvReflect(cin, se, 1)
        self.conv2 = _ConvReflect(se, cout, 1)


class _SERes2Net(nn.Module):

    def __init__(self, cin, cout, scale, se, k):
        super().__init__()
        self.tdnn1 = _TDNN(cin, cout, 1)
        self.res2net_block = _Res2Net(cout, scale, k)
        self.tdnn2 = _TDNN(cout, cout, 1)
        self.se_block = _SE(cout, se, cout)
        self.shortcut = _ConvReflect(cin, cout, 1) if cin != cout else nn.Identity()


class _ASP(nn.Module):

    def __init__(self, ch, att):
        super().__init__()
        self.tdnn = _TDNN(ch * 3, att, 1)
        self.conv = _ConvReflect(att, ch, 1)


class ECAPATDNN(nn.Module):
    """Drop-in for the reference ECAPATDNN in the shipped settings (pooling, global
    context, groups 1, regularisation 'ac' or 'none')."""

    def __init__(self, in_size, out_dim, channels, kernel_sizes, dilations, groups, res2net_scale,
                 se_channels, attention_channels, global_context, pooling, use_tanh,
                 spherical_normalisation, regularisation="none"):
        super().__init__()
        if not (global_context and pooling) or spherical_normalisation or \
                regularisation not in ("ac", "none") or any(g != 1 for g in groups):
            raise NotImplementedError("after_amd builds ECAPATDNN as the shipped configs use it")
        self.in_size, self.out_dim = in_size, out_dim
        self.channels, self.kernel_sizes, self.dilations = list(channels), list(kernel_sizes), list(
            dilations)
        self.res2net_scale, self.se_channels = res2net_scale, se_channels
        self.attention_channels, self.use_tanh = attention_channels, use_tanh
        self.blocks = nn.ModuleList([_TDNN(in_size, channels[0], kernel_sizes[0])])
        for i in range(1, len(channels) - 1):
            self.blocks.append(_SERes2Net(channels[i - 1], channels[i], res2net_scale, se_channels,
                                          kernel_sizes[i]))
        self.mfa = _TDNN(channels[-1], channels[-1], kernel_sizes[-1])
        self.asp = _ASP(channels[-1], attention_channels)
        self.asp_bn = nn.BatchNorm1d(channels[-1] * 2)
        self.fc = _ConvReflect(channels[-1] * 2, out_dim, 1)
        self.requires_grad_(False)
        self.eval()
        self._handle = None
        self._cap = (0, 0)

    def _apply(self, fn, *a, **k):
        self._release()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._release()
        return super().load_state_dict(*a, **k)

    def refresh(self):
        """Call after mutating parameters in place: the HIP handle owns a re-laid-out copy."""
        self._release()

    def _release(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            _lib.lib().after_ecapa_destroy(h)
        self._handle = None
        self._cap = (0, 0)

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _weight_names(self):
        def TD(p):
            return [p + "conv.conv.weight", p + "conv.conv.bias", p + "norm.weight", p + "norm.bias",
                    p + "norm.running_mean", p + "norm.running_var"]

        def CV(p):
            return [p + "conv.weight", p + "conv.bias"]

        n = len(self.channels)
        names = TD("blocks.0.")
        for i in range(1, n - 1):
            b = f"blocks.{i}."
            names += TD(b + "tdnn1.")
            for j in range(self.res2net_scale - 1):
                names += TD(f"{b}res2net_block.blocks.{j}.")
            names += TD(b + "tdnn2.") + CV(b + "se_block.conv1.") + CV(b + "se_block.conv2.")
            if self.channels[i - 1] != self.channels[i]:
                names += CV(b + "shortcut.")
        names += TD("mfa.") + TD("asp.tdnn.") + CV("asp.conv.")
        names += ["asp_bn.weight", "asp_bn.bias", "asp_bn.running_mean", "asp_bn.running_var"]
        names += CV("fc.")
        return names

    def _ensure(self, B, T):
        cb, ct = self._cap
        if self._handle is not None and B <= cb and T <= ct:
            return self._handle
        L = _lib.lib()
        self._release()
        cap = (max(B, cb), max(T, ct))
        sd = self.state_dict()
        ws = [_lib.require_gpu_tensor(sd[n], n) for n in self._weight_names()]
        arr = (ctypes.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
        cfg = _lib.EcapaCfg()
        cfg.in_size, cfg.out_dim, cfg.n_blocks = self.in_size, self.out_dim, len(self.channels)
        for i in range(len(self.channels)):
            cfg.channels[i] = self.channels[i]
            cfg.kernel_sizes[i] = self.kernel_sizes[i]
            cfg.dilations[i] = self.dilations[i]
        cfg.res2net_scale = self.res2net_scale
        cfg.se_channels = self.se_channels
        cfg.attention_channels = self.attention_channels
        cfg.use_tanh = int(self.use_tanh)
        out = ctypes.c_void_p()
        dev = next(w for w in ws if w is not None).device
        with torch.cuda.device(dev):
            torch.cuda.synchronize(dev)
            rc = L.after_ecapa_create(ctypes.byref(cfg), arr, len(ws), cap[0], cap[1], ctypes.byref(out))
        _lib.check(rc, "after_ecapa_create")
        self._handle = out
        self._cap = cap
        return out

    @torch.no_grad()
    def forward(self, X, return_full=False):
        """ecapa_encoder.py:567-624."""
        X = _lib.require_gpu_tensor(X, "X")
        B, C, T = X.shape
        if C != self.in_size:
            raise ValueError(f"expected {self.in_size} channels, got {C}")
        h = self._ensure(B, T)
        out = torch.empty(B, self.out_dim, device=X.device)
        with torch.cuda.device(X.device):
            _lib.check(_lib.lib().after_ecapa_forward(h, _lib.ptr(X), _lib.ptr(out), B, T,
                                                      _lib.current_stream(X.device)),
                       "after_ecapa_forward")
        if return_full:
            return out, out, torch.zeros((), device=X.device)
        return out

    def forward_stream(self, X):
        """ecapa_encoder.py:626-666."""
        return self.forward(X)
