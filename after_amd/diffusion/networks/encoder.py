"""Host-side mirror of the reference's Encoder1D
(after/diffusion/networks/encoder.py:116-322), used as `encoder_time`.

Parameter containers under the reference's state-dict keys (each BatchNorm is
registered twice, as gn1/gn2 and inside the branch Sequential -- encoder.py:51-58 --
and the mirror keeps that aliasing); compute in libafter_hip
(after_encoder1d_forward)."""
import ctypes

import torch
from torch import nn

from ... import _lib
from ...autoencoder.model import _Branches, _WNConv


class _V2ConvBlock(nn.Module):

    def __init__(self, ch, k):
        super().__init__()
        self.gn1 = nn.BatchNorm1d(ch)
        self.gn2 = nn.BatchNorm1d(ch)
        main = nn.Sequential(self.gn1, nn.SiLU(), _WNConv(ch, ch, k), self.gn2, nn.SiLU(),
                             nn.Dropout(0.15), _WNConv(ch, ch, k))
        self.net = _Branches(main, nn.Identity())


class _V2EncoderBlock(nn.Module):

    def __init__(self, cin, cout, k, ratio):
        super().__init__()
        pool = _WNConv(cin, cout, 1 if ratio == 1 else 2 * ratio)
        self.net = nn.Sequential(_V2ConvBlock(cin, k), pool)


class Encoder1D(nn.Module):
    """Drop-in for the reference Encoder1D in the shipped settings (no conditioning,
    no VQ / VAE / upscaling; `ac` regularisation is the identity on z, :261-263)."""

    def __init__(self, in_size=1, channels=(64, 128, 128, 256, 256), ratios=(2, 2, 2, 2, 2),
                 kernel_size=5, cond=None, use_tanh=True, average_out=False, upscale_out=False,
                 vector_quantizer=None, spherical_normalization=False, vae_regularisation=False,
                 ac_regularisation=False, wassertstein_regularisation=False,
                 padding_mode="centered"):
        super().__init__()
        if cond or average_out or upscale_out or vector_quantizer is not None or \
                spherical_normalization or vae_regularisation or wassertstein_regularisation:
            raise NotImplementedError("after_amd builds Encoder1D as the shipped configs use it")
        self.in_size = in_size
        self.channels = list(channels)
        self.ratios = [1] + list(ratios)
        self.kernel_size = kernel_size
        self.use_tanh = use_tanh
        self.padding_mode = padding_mode
        self.out_channels = self.channels[-1]
        self.total_ratio = 1
        for r in self.ratios:
            self.total_ratio *= r
        n = len(self.channels)
        mods = [_V2EncoderBlock(in_size, self.channels[0], kernel_size, self.ratios[0])]
        for i in range(1, n):
            mods.append(_V2EncoderBlock(self.channels[i - 1], self.channels[i], kernel_size,
                                        self.ratios[i]))
        mods.append(_V2ConvBlock(self.channels[-1], kernel_size))
        self.net = nn.Sequential(*mods)
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
            _lib.lib().after_encoder1d_destroy(h)
        self._handle = None
        self._cap = (0, 0)

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _weight_names(self):
        def BN(p):
            return [p + "weight", p + "bias", p + "running_mean", p + "running_var"]

        def WN(p):
            return [p + "weight_g", p + "weight_v", p + "bias"]

        def V2(p):
            b = p + "net.branches.0."
            return BN(b + "0.") + WN(b + "2.") + BN(b + "3.") + WN(b + "6.")

        n = len(self.channels)
        names = []
        for i in range(n):
            names += V2(f"net.{i}.net.0.") + WN(f"net.{i}.net.1.")
        names += V2(f"net.{n}.")
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
        cfg = _lib.Encoder1dCfg()
        cfg.in_size = self.in_size
        cfg.n_blocks = len(self.channels)
        for i, c in enumerate(self.channels):
            cfg.channels[i] = c
        for i, r in enumerate(self.ratios[:len(self.channels)]):
            cfg.ratios[i] = r
        cfg.kernel_size = self.kernel_size
        cfg.causal = int(self.padding_mode == "causal")
        cfg.use_tanh = int(self.use_tanh)
        out = ctypes.c_void_p()
        dev = next(w for w in ws if w is not None).device
        with torch.cuda.device(dev):
            torch.cuda.synchronize(dev)
            rc = L.after_encoder1d_create(ctypes.byref(cfg), arr, len(ws), cap[0], cap[1], ctypes.byref(out))
        _lib.check(rc, "after_encoder1d_create")
        self._handle = out
        self._cap = cap
        if getattr(self, "_streaming", False):  # a re-created handle starts a fresh stream
            _lib.check(L.after_encoder1d_enable_streaming(out, 1), "after_encoder1d_enable_streaming")
        return out

    @torch.no_grad()
    def forward(self, x, return_full: bool = False):
        """encoder.py:273-298."""
        x = _lib.require_gpu_tensor(x, "x")
        B, C, T = x.shape
        if C != self.in_size:
            raise ValueError(f"expected {self.in_size} channels, got {C}")
        if T % self.total_ratio:
            raise ValueError(f"T={T} is not a multiple of the total ratio {self.total_ratio}")
        h = self._ensure(B, T)
        out = torch.empty(B, self.channels[-1], T // self.total_ratio, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().after_encoder1d_forward(h, _lib.ptr(x), _lib.ptr(out), B, T,
                                                          _lib.current_stream(x.device)),
                       "after_encoder1d_forward")
        if return_full:
            return out, out, torch.zeros((), device=x.device)
        return out

    def forward_stream(self, x):
        """encoder.py:300-322.  Offline semantics unless `enable_streaming` was called: then the
        causal convs keep their left context between calls (cc.use_cached_conv(True),
        export.py:17,438-441) and consecutive chunks continue one stream."""
        return self.forward(x)

    def enable_streaming(self, batch: int, chunk_frames: int, enable: bool = True):
        h = self._ensure(batch, chunk_frames)
        _lib.check(_lib.lib().after_encoder1d_enable_streaming(h, int(enable)),
                   "after_encoder1d_enable_streaming")
        self._streaming = bool(enable)

    def reset_state(self):
        if self._handle is None:
            raise RuntimeError("enable_streaming first")
        dev = next(self.parameters()).device
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().after_encoder1d_reset_state(self._handle, _lib.current_stream(dev)),
                       "after_encoder1d_reset_state")
