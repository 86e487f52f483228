"""Host-side mirror of the reference's AutoEncoder
(after/autoencoder/networks/SimpleNetsStream.py:831-954) and of the exported
encode / decode / forward surface (after_scripts/export_autoencoder.py:235-265).

The module tree only carries parameters under the reference's state-dict keys
(weight-normed convs as weight_g / weight_v, CachedGroupNorm
with its `pad` buffer, SnakeBeta alpha / beta, the four PQMF buffers); compute runs
in libafter_hip (after_ae_encode / after_ae_decode)."""
import ctypes
import math
from typing import Sequence

import torch
from torch import nn

from .. import _lib
from . import pqmf as pqmf_design


class _Snake(nn.Module):

    def __init__(self, dim):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(dim))
        self.beta = nn.Parameter(torch.ones(dim))


class _CachedGroupNorm(nn.Module):

    def __init__(self, groups, channels):
        super().__init__()
        self.gn = nn.GroupNorm(groups, channels)
        self.register_buffer("pad", torch.zeros(4, channels, 1))


class _WNConv(nn.Module):
    """Parameters of torch.nn.utils.weight_norm(nn.Conv1d / nn.ConvTranspose1d)."""

    def __init__(self, cin, cout, k, transposed=False):
        super().__init__()
        ref = nn.ConvTranspose1d(cin, cout, k) if transposed else nn.Conv1d(cin, cout, k)
        v = ref.weight.detach()
        self.bias = nn.Parameter(ref.bias.detach().clone())
        self.weight_g = nn.Parameter(v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1).clone())
        self.weight_v = nn.Parameter(v.clone())


class _ConvBlock(nn.Module):

    def __init__(self, cin, cout, k, use_norm):
        super().__init__()
        gn = _CachedGroupNorm(min(cin, 8), cin) if use_norm else nn.Identity()
        self.net = nn.Sequential(gn, _Snake(cin), _WNConv(cin, cout, k))


class _Branches(nn.Module):

    def __init__(self, *mods):
        super().__init__()
        self.branches = nn.ModuleList(mods)


class _ResBlock(nn.Module):

    def __init__(self, cin, cout, k, use_norm):
        super().__init__()
        main = nn.Sequential(_ConvBlock(cin, cout, k, use_norm), _ConvBlock(cout, cout, 1, use_norm))
        short = _WNConv(cin, cout, 1) if cin != cout else nn.Identity()
        self.net = _Branches(main, short)


class _NoRes(nn.Module):

    def __init__(self, cin, cout, k, use_norm):
        super().__init__()
        self.net = nn.Sequential(_ConvBlock(cin, cout, k, use_norm), _ConvBlock(cout, cout, 1, use_norm))


class _Down(nn.Module):

    def __init__(self, cin, cout, f, nd, k, use_norm):
        super().__init__()
        mods = [_ResBlock(cin, cin, k, use_norm) for _ in range(nd)]
        mods += [_Snake(cin), _WNConv(cin, cout, 2 * f)]
        self.net = nn.Sequential(*mods)


class _Up(nn.Module):

    def __init__(self, cin, cout, f, nd, k, use_norm):
        super().__init__()
        mods = [_Snake(cin), _WNConv(cin, cout, 2 * f, transposed=True)]
        mods += [_ResBlock(cout, cout, k, use_norm) for _ in range(nd)]
        self.net = nn.Sequential(*mods)


class _Encoder(nn.Module):

    def __init__(self, cin, channels, mult, factors, nd, k, zc, use_norm):
        super().__init__()
        mods = [_ResBlock(cin, channels * mult[0], k, use_norm)]
        for i, f in enumerate(factors):
            mods.append(_Down(channels * mult[i], channels * mult[i + 1], f, nd, k, use_norm))
        mods += [_Snake(channels * mult[-1]), _WNConv(channels * mult[-1], zc, 3)]
        self.net = nn.Sequential(*mods)


class _Decoder(nn.Module):

    def __init__(self, cout, channels, mult, factors, nd, k, zc, use_norm, use_loudness, use_noise=False):
        super().__init__()
        mods = [_WNConv(zc, channels * mult[0], k)]
        for i, f in enumerate(factors):
            mods.append(_Up(channels * mult[i], channels * mult[i + 1], f, nd, k, use_norm))
        self.net = nn.Sequential(*mods)
        branches = [_NoRes(channels * mult[-1], cout * 2 if use_loudness else cout, k, use_norm)]
        if use_noise:  # SimpleNetsStream.py:622-627: NoiseGenerator(in_size, data_size=out_channels, ratios=[2,2,2], noise_bands=5)
            self.noise_module = _NoiseGenerator(channels * mult[-1], cout)
            branches.append(self.noise_module)
        self.synth = _Branches(*branches)


class _NoiseGenerator(nn.Module):
    """Parameters of SimpleNetsStream.py:499-535 (hidden 128, ratios [2, 2, 2], 5 noise bands): plain Conv1d k = 3 at
    `net.0 / .2 / .4` (LeakyReLU(0.2) in between) and the `target_size` buffer."""

    def __init__(self, in_size, data_size, hidden=128, noise_bands=5):
        super().__init__()
        ch = [in_size, hidden, hidden, data_size * noise_bands]
        self.net = nn.Sequential(nn.Conv1d(ch[0], ch[1], 3), nn.Identity(), nn.Conv1d(ch[1], ch[2], 3), nn.Identity(),
                                 nn.Conv1d(ch[2], ch[3], 3))
        self.register_buffer("target_size", torch.tensor(8).long())


class _PQMF(nn.Module):

    def __init__(self, attenuation, n_band):
        super().__init__()
        h, hk, fw, iw = pqmf_design.design(attenuation, n_band)
        self.register_buffer("hk", torch.from_numpy(hk))
        self.register_buffer("h", torch.from_numpy(h))
        self.forward_conv = nn.Module()
        self.forward_conv.register_parameter("weight", nn.Parameter(torch.from_numpy(fw)))
        self.inverse_conv = nn.Module()
        self.inverse_conv.register_parameter("weight", nn.Parameter(torch.from_numpy(iw)))


class ReluBottleneck(nn.Module):
    """SimpleNetsStream.py:742-760: identity on z at inference (`apply_noise=False`), plus the
    regulariser mean(ELU(|z| - scale)) + 1 (core.py:189-198) that `encode` returns beside z."""

    def __init__(self, scale: float = 3, sigma: float = 0.):
        super().__init__()
        self.scale = float(scale)
        self.sigma = float(sigma)


class TanhBottleneck(nn.Module):
    """SimpleNetsStream.py:719-740: z = scale * tanh(z) (+ sigma * randn: `AutoEncoder.encode` calls the
    bottleneck with its default apply_noise=True; sigma defaults to 0), regulariser tensor(0.)."""

    def __init__(self, scale: float = 3, sigma: float = 0.):
        super().__init__()
        self.scale = float(scale)
        self.sigma = float(sigma)


class VAEBottleneck(nn.Module):
    """SimpleNetsStream.py:763-785: the encoder emits 2 x z_channels (mean | scale, :864-867);
    z = randn * (softplus(scale) + 1e-2) + mean, regulariser = the KL term; `encode(return_mean=True)` also
    returns the mean."""


def _resolve_bottleneck(b):
    """The bottleneck shapes what `encode` returns, and it has no parameters -- a checkpoint cannot reveal which one
    a codec was trained with -- so the binding (baseAE.gin:41-43,48) must be named: None / "relu" ->
    ReluBottleneck, "tanh" -> TanhBottleneck, "vae" -> VAEBottleneck, or an instance."""
    if b is None or (isinstance(b, str) and b.lower() in ("relu", "relubottleneck")):
        return ReluBottleneck()
    if isinstance(b, str) and b.lower() in ("tanh", "tanhbottleneck"):
        return TanhBottleneck()
    if isinstance(b, str) and b.lower() in ("vae", "vaebottleneck"):
        return VAEBottleneck()
    if isinstance(b, (ReluBottleneck, TanhBottleneck, VAEBottleneck)):
        return b
    name = b if isinstance(b, str) else type(b).__name__
    raise NotImplementedError(f"bottleneck {name!r}: SimpleNetsStream.py defines Relu / Tanh / VAE bottlenecks")


def _resolve_activation(a):
    """The codec's `activation` class (SimpleNetsStream.py:161,169: built as `activation(dim=channels)`) -> "beta" for
    SnakeBeta (core.py:227-268; imported AS `Snake` by SimpleNetsStream.py:15, the default everywhere) or "alpha" for
    the one-parameter core.Snake (core.py:201-209: x + sin^2(alpha x) / (alpha + 1e-9), alpha of shape [dim, 1]).
    A bare "Snake" means what it means inside SimpleNetsStream -- SnakeBeta; the one-parameter class is named by its
    module ("core.Snake", "after.autoencoder.core.Snake") or passed as the class itself."""
    if a is None:
        return "beta"
    if isinstance(a, str):
        parts = a.lstrip("@").split(".")
        name, mod = parts[-1], parts[:-1]
    else:
        name = getattr(a, "__name__", type(a).__name__)
        mod = str(getattr(a, "__module__", "")).split(".")
    if name == "SnakeBeta" or (name == "Snake" and (not mod or mod[-1] != "core")):
        return "beta"
    if name == "Snake":
        return "alpha"
    raise NotImplementedError(f"activation {name!r}: after_amd builds the reference's two snake activations "
                              "(core.py:201-209 Snake, :227-268 SnakeBeta); no other class in the reference takes "
                              "the `dim=` argument ConvBlock1d passes")


class AutoEncoder(nn.Module):
    """Drop-in for the reference AutoEncoder on MI355X: same constructor arguments, same
    encode/decode/forward signatures.  `bottleneck`: None / "relu" / a ReluBottleneck
    (identity on z at inference + the regulariser, SimpleNetsStream.py:742-760)."""

    def __init__(self,
                 in_channels: int,
                 channels: int,
                 z_channels: int,
                 multipliers: Sequence[int],
                 factors: Sequence[int],
                 dilations: Sequence[int],
                 kernel_size: int,
                 resnet_groups: int = 8,
                 bottleneck=None,
                 activation=None,
                 use_norm: bool = True,
                 decoder_ratio: float = 1,
                 pqmf_bands: int = 0,
                 use_loudness: bool = False,
                 use_noise: bool = False,
                 padding_mode: str = "centered"):
        super().__init__()
        # SimpleNetsStream.py:853-859: pqmf_bands > 1 -> CachedPQMF, the codec's input channels are the bands; otherwise
        # DummyIdentity -- the codec runs on the (mono) audio itself (baseAE.gin:15: "Set to 1 if no pqmf")
        if pqmf_bands > 1 and in_channels != pqmf_bands:
            raise NotImplementedError("a PQMF codec has in_channels = pqmf_bands")
        if pqmf_bands <= 1 and in_channels != 1:
            raise NotImplementedError("without PQMF (pqmf_bands <= 1) after_amd builds the mono codec: in_channels = 1")
        if use_noise and padding_mode != "centered":
            raise NotImplementedError("use_noise=True (NoiseGenerator) is built for centred padding (whole-clip decoding)")
        if resnet_groups != 8:
            raise NotImplementedError("resnet_groups must be 8 (every shipped config)")
        self.snake = _resolve_activation(activation)
        self.bottleneck = _resolve_bottleneck(bottleneck)
        self.cfg = dict(in_channels=in_channels, channels=channels, z_channels=z_channels,
                        multipliers=list(multipliers), factors=list(factors),
                        dilations=list(dilations), kernel_size=kernel_size, use_norm=use_norm,
                        decoder_ratio=decoder_ratio, pqmf_bands=pqmf_bands,
                        use_loudness=use_loudness, use_noise=use_noise, padding_mode=padding_mode)
        self.pqmf_bands = pqmf_bands
        self._bands = max(1, pqmf_bands)  # what the C side calls pqmf_bands: 1 = the identity bank
        self.z_channels = z_channels
        self.ratio = self._bands * math.prod(factors)
        nd = len(dilations)
        self.dec_multipliers = [int(m * decoder_ratio) for m in list(multipliers)[::-1]]
        # (no parameters without PQMF, like the reference's DummyIdentity: the state dicts stay interchangeable)
        self.pqmf = _PQMF(100, pqmf_bands) if pqmf_bands > 1 else nn.Identity()
        # SimpleNetsStream.py:864-867: a VAE codec's encoder emits mean and scale
        self.encoder_out_channels = 2 * z_channels if isinstance(self.bottleneck, VAEBottleneck) else z_channels
        self.encoder = _Encoder(in_channels, channels, list(multipliers), list(factors), nd,
                                kernel_size, self.encoder_out_channels, use_norm)
        self.decoder = _Decoder(in_channels, channels, self.dec_multipliers, list(factors)[::-1], nd,
                                kernel_size, z_channels, use_norm, use_loudness, use_noise)
        if self.snake == "alpha":  # core.Snake: one parameter of shape [dim, 1] where SnakeBeta has two of [dim]
            for m in self.modules():
                if isinstance(m, _Snake):
                    del m.beta
                    m.alpha = nn.Parameter(torch.ones(m.alpha.shape[0], 1))
        self.requires_grad_(False)
        self._handle = None
        self._cap = (0, 0)

    def cfg_kwargs(self):
        """Constructor arguments of an identical codec (streaming twin, after_amd.streaming)."""
        return dict(self.cfg, bottleneck=self.bottleneck, activation="core.Snake" if self.snake == "alpha" else None)

    # ------------------------------------------------------------ handle management
    def _apply(self, fn, *a, **k):
        self._release()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._release()
        return super().load_state_dict(*a, **k)

    def refresh(self):
        self._release()

    def _release(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            _lib.lib().after_ae_destroy(h)
        self._handle = None
        self._cap = (0, 0)

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _weight_names(self):
        """State-dict keys in the order documented in include/after_hip.h."""
        c = self.cfg
        nd, n = len(c["dilations"]), len(c["factors"])

        # the C side takes (alpha, beta) per activation and computes 1 / (beta + 1e-9): core.Snake is beta = alpha
        bk = "beta" if self.snake == "beta" else "alpha"

        def CB(p):
            return [p + "net.0.gn.weight", p + "net.0.gn.bias", p + "net.1.alpha", p + "net.1." + bk,
                    p + "net.2.weight_g", p + "net.2.weight_v", p + "net.2.bias"]

        def WN(p):
            return [p + "weight_g", p + "weight_v", p + "bias"]

        def SN(p):
            return [p + "alpha", p + bk]

        names = ["pqmf.forward_conv.weight", "pqmf.inverse_conv.weight"]  # (identity bank: replaced by the unit tap in _ensure)
        e = "encoder.net."
        names += CB(e + "0.net.branches.0.0.") + CB(e + "0.net.branches.0.1.")
        if c["in_channels"] != c["channels"] * c["multipliers"][0]:
            names += WN(e + "0.net.branches.1.")
        for i in range(n):
            s = f"{e}{i + 1}.net."
            for j in range(nd):
                names += CB(f"{s}{j}.net.branches.0.0.") + CB(f"{s}{j}.net.branches.0.1.")
            names += SN(f"{s}{nd}.") + WN(f"{s}{nd + 1}.")
        names += SN(f"{e}{n + 1}.") + WN(f"{e}{n + 2}.")
        d = "decoder.net."
        names += WN(d + "0.")
        for i in range(n):
            s = f"{d}{i + 1}.net."
            names += SN(s + "0.") + WN(s + "1.")
            for j in range(nd):
                names += CB(f"{s}{j + 2}.net.branches.0.0.") + CB(f"{s}{j + 2}.net.branches.0.1.")
        names += CB("decoder.synth.branches.0.net.0.") + CB("decoder.synth.branches.0.net.1.")
        if c["use_noise"]:  # (the module is registered twice, as in the reference: decoder.noise_module and decoder.synth.branches.1)
            for i in (0, 2, 4):
                names += [f"decoder.noise_module.net.{i}.weight", f"decoder.noise_module.net.{i}.bias"]
        return names

    def _ensure(self, batch: int, samples: int):
        cb, cs = self._cap
        if self._handle is not None and batch <= cb and samples <= cs:
            return self._handle
        L = _lib.lib()
        self._release()
        cap = (max(batch, cb), max(samples, cs))
        sd = self.state_dict()
        ws = []
        if self._bands == 1:
            self._unit_tap = torch.ones(1, device=next(self.parameters()).device)
        for name in self._weight_names():
            if self._bands == 1 and name.startswith("pqmf."):
                ws.append(self._unit_tap)
            elif name.endswith("gn.weight") or name.endswith("gn.bias"):
                ws.append(_lib.require_gpu_tensor(sd[name], name) if name in sd else None)
            else:
                ws.append(_lib.require_gpu_tensor(sd[name], name))
        arr = (ctypes.c_void_p * len(ws))(*[w.data_ptr() if w is not None else None for w in ws])
        c = self.cfg
        cfg = _lib.AECfg()
        cfg.pqmf_bands = self._bands
        cfg.channels = c["channels"]
        cfg.z_channels = c["z_channels"]
        cfg.encoder_out_channels = self.encoder_out_channels
        cfg.n_stages = len(c["factors"])
        cfg.n_dilations = len(c["dilations"])
        cfg.kernel_size = c["kernel_size"]
        cfg.use_norm = int(c["use_norm"])
        cfg.use_loudness = int(c["use_loudness"])
        cfg.use_noise = int(c["use_noise"])
        cfg.causal = int(c["padding_mode"] == "causal")
        for i, m in enumerate(c["multipliers"]):
            cfg.multipliers[i] = m
        for i, m in enumerate(self.dec_multipliers):
            cfg.dec_multipliers[i] = m
        for i, f in enumerate(c["factors"]):
            cfg.factors[i] = f
        for i, dd in enumerate(c["dilations"]):
            cfg.dilations[i] = dd
        out = ctypes.c_void_p()
        dev = next(w for w in ws if w is not None).device
        with torch.cuda.device(dev):
            torch.cuda.synchronize(dev)
            rc = L.after_ae_create(ctypes.byref(cfg), arr, len(ws), cap[0], cap[1], ctypes.byref(out))
        _lib.check(rc, "after_ae_create")
        self._handle = out
        self._cap = cap
        if getattr(self, "_streaming", False):  # a re-created handle starts a fresh stream
            _lib.check(L.after_ae_enable_streaming(out, 1), "after_ae_enable_streaming")
            if getattr(self, "_lane_rows", 0) and 2 * self._lane_rows <= cap[0]:
                _lib.check(L.after_ae_set_stream_lanes(out, self._lane_rows), "after_ae_set_stream_lanes")
        if getattr(self, "_dec_gn_window", None):
            _lib.check(L.after_ae_set_decoder_gn_window(out, int(self._dec_gn_window)),
                       "after_ae_set_decoder_gn_window")
        if getattr(self, "_enc_stream_window", None) is not None:
            _lib.check(L.after_ae_enable_encoder_streaming(out, 1, int(self._enc_stream_window)),
                       "after_ae_enable_encoder_streaming")
        return out

    def reserve(self, batch: int, samples: int):
        self._ensure(batch, samples)

    # ------------------------------------------------------------ streaming (nn~ / real time)
    def enable_streaming(self, batch: int, chunk_samples: int, enable: bool = True):
        """`cc.use_cached_conv(True)` twin of the causal codec (export_autoencoder.py:293-303):
        encode / decode become stateful over consecutive chunks of `batch` streams.  Only the
        causal, GroupNorm-free configuration streams (baseAE.gin:32-33,49)."""
        h = self._ensure(batch, chunk_samples)
        _lib.check(_lib.lib().after_ae_enable_streaming(h, int(enable)), "after_ae_enable_streaming")
        self._streaming = bool(enable)

    def enable_encoder_streaming(self, batch: int, chunk_samples: int, gn_window_samples: int = 131072,
                                 enable: bool = True) -> int:
        """The streaming twin of a NON-causal codec's encoder, as export_autoencoder.py:305-312 builds it
        (`model.encoder` under cc.use_cached_conv(True) with CachedGroupNorm.stream = True; PQMF and
        decoder stay offline): `encode` becomes stateful over consecutive chunks of `batch` streams, every
        conv reading cached past context, shortcuts and strided convs delay-compensated, GroupNorm
        statistics taken over the previous `gn_window_samples` + the chunk (the reference's window is the
        length of the module's first call: 131072 in the export script).  Returns the lag of the latents
        behind the offline encoder, in latent frames."""
        h = self._ensure(batch, max(chunk_samples, self.ratio))
        L = _lib.lib()
        _lib.check(L.after_ae_enable_encoder_streaming(h, int(enable), int(gn_window_samples)),
                   "after_ae_enable_encoder_streaming")
        self._enc_stream_window = int(gn_window_samples) if enable else None
        return int(L.after_ae_encoder_delay(h))

    def set_decoder_gn_window(self, batch: int, chunk_samples: int, window_latent_frames: int = 64):
        """CachedGroupNorm(stream=True) on the decoder, as the same export binds it (SimpleNetsStream.py:
        95-147): GroupNorm statistics over the previous `window_latent_frames` + the call (the reference's
        window is the length of the first decode after construction: 64 frames in the export script).
        0 = plain GroupNorm.  `reset_state` starts a new stream."""
        h = self._ensure(batch, max(chunk_samples, self.ratio))
        _lib.check(_lib.lib().after_ae_set_decoder_gn_window(h, int(window_latent_frames)),
                   "after_ae_set_decoder_gn_window")
        self._dec_gn_window = int(window_latent_frames) or None

    @property
    def encoder_delay(self) -> int:
        return int(_lib.lib().after_ae_encoder_delay(self._handle)) if self._handle is not None else 0

    def reset_state(self):
        """Start of a new stream: zero every conv / PQMF context."""
        if self._handle is None:
            raise RuntimeError("enable_streaming first")
        dev = next(self.parameters()).device
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().after_ae_reset_state(self._handle, _lib.current_stream(dev)),
                       "after_ae_reset_state")

    def set_stream_lanes(self, lane_rows: int):
        """Two independent groups of `lane_rows` streams in this streaming encoder (see `encode(row0=...)`); call on a
        freshly enabled / reset stream."""
        if self._handle is None:
            raise RuntimeError("enable_streaming first")
        _lib.check(_lib.lib().after_ae_set_stream_lanes(self._handle, int(lane_rows)), "after_ae_set_stream_lanes")
        self._lane_rows = int(lane_rows)

    # ------------------------------------------------------------ reference surface
    @torch.no_grad()
    def encode(self, x, with_multi: bool = False, return_mean: bool = False, row0=None):
        """SimpleNetsStream.py:918-941 -> (z, regloss); export: z only.  row0 (streaming codecs with lanes): the first
        context row of this batch -- 0 or `lane_rows` for one lane, 0 with a batch of 2 x lane_rows for both."""
        vae = isinstance(self.bottleneck, VAEBottleneck)
        if return_mean and not vae:  # :932-934 is the VAEBottleneck protocol
            raise TypeError("return_mean=True needs a VAEBottleneck (the other bottlenecks take no such argument)")
        x = _lib.require_gpu_tensor(x, "x")
        if x.dim() != 3 or x.shape[1] != 1:
            raise ValueError(f"encode expects [B, 1, L], got {tuple(x.shape)}")
        B, _, L = x.shape
        if L % self.ratio:
            raise ValueError(f"length {L} is not a multiple of the codec ratio {self.ratio}")
        h = self._ensure(B, L)
        T = L // self.ratio
        z = torch.empty(B, self.encoder_out_channels, T, device=x.device, dtype=torch.float32)
        L_ = _lib.lib()
        mean = None
        with torch.cuda.device(x.device):
            st = _lib.current_stream(x.device)
            if row0 is None:
                _lib.check(L_.after_ae_encode(h, _lib.ptr(x), _lib.ptr(z), B, L, st), "after_ae_encode")
            else:
                _lib.check(L_.after_ae_encode_rows(h, _lib.ptr(x), _lib.ptr(z), B, L, int(row0), st), "after_ae_encode_rows")
            reg = torch.zeros((), device=x.device, dtype=torch.float32)
            if isinstance(self.bottleneck, ReluBottleneck):
                # z unchanged, reg = mean(ELU(|z| - scale)) + 1 (:753-760)
                _lib.check(L_.after_latent_reg(_lib.ptr(z), z.numel(), self.bottleneck.scale, _lib.ptr(reg), st),
                           "after_latent_reg")
            elif isinstance(self.bottleneck, TanhBottleneck):
                _lib.check(L_.after_bottleneck_tanh(_lib.ptr(z), z.numel(), self.bottleneck.scale, st),
                           "after_bottleneck_tanh")
                if self.bottleneck.sigma > 0:  # :733-735 (apply_noise defaults to True)
                    z.add_(self.bottleneck.sigma * torch.randn_like(z))
            else:  # VAE: z = randn * std + mean, reg = KL (:770-785)
                zraw, Z = z, self.z_channels
                z = torch.empty(B, Z, T, device=x.device, dtype=torch.float32)
                mean = torch.empty_like(z)
                noise = torch.randn(B, Z, T, device=x.device, dtype=torch.float32)
                _lib.check(L_.after_bottleneck_vae(_lib.ptr(zraw), _lib.ptr(noise), _lib.ptr(z), _lib.ptr(mean),
                                                   _lib.ptr(reg), B, Z, T, st), "after_bottleneck_vae")
        if return_mean:
            return z, reg, mean
        if with_multi:
            return z, self.pqmf_forward(x), reg
        return z, reg

    @torch.no_grad()
    def decode(self, z, with_multi: bool = False, noise_u=None):
        """SimpleNetsStream.py:943-954; with_multi=True also returns x_multiband, the decoder
        output before the PQMF synthesis bank.  use_noise codecs: `noise_u` = the uniform [0, 1) draws of the
        NoiseGenerator ([B, T_band / 8, bands, 8]: the reference's torch.rand_like(ir), :545); drawn here when None."""
        z = _lib.require_gpu_tensor(z, "z")
        if z.dim() != 3 or z.shape[1] != self.z_channels:
            raise ValueError(f"decode expects [B, {self.z_channels}, T], got {tuple(z.shape)}")
        B, _, T = z.shape
        h = self._ensure(B, T * self.ratio)
        x = torch.empty(B, 1, T * self.ratio, device=z.device, dtype=torch.float32)
        if self.cfg["use_noise"]:
            shape = (B, T * self.ratio // self._bands // 8, self._bands, 8)
            if noise_u is None:
                noise_u = torch.rand(shape, device=z.device, dtype=torch.float32)
            noise_u = _lib.require_gpu_tensor(noise_u, "noise_u")
            if tuple(noise_u.shape) != shape:
                raise ValueError(f"noise_u must be {shape}, got {tuple(noise_u.shape)}")
            _lib.check(_lib.lib().after_ae_set_noise(h, _lib.ptr(noise_u)), "after_ae_set_noise")
        with torch.cuda.device(z.device):
            if with_multi:
                mb = torch.empty(B, self._bands, T * self.ratio // self._bands, device=z.device,
                                 dtype=torch.float32)
                _lib.check(_lib.lib().after_ae_decode_multi(h, _lib.ptr(z), _lib.ptr(x), _lib.ptr(mb), B, T,
                                                            _lib.current_stream(z.device)),
                           "after_ae_decode_multi")
                return x, mb
            _lib.check(_lib.lib().after_ae_decode(h, _lib.ptr(z), _lib.ptr(x), B, T,
                                                  _lib.current_stream(z.device)), "after_ae_decode")
        return x

    def forward(self, x):
        """export_autoencoder.py:235-249: decode(encode(x))."""
        return self.decode(self.encode(x)[0])

    @torch.no_grad()
    def pqmf_forward(self, x):
        x = _lib.require_gpu_tensor(x, "x")
        B, _, L = x.shape
        h = self._ensure(B, max(L, self.ratio))
        mb = torch.empty(B, self._bands, L // self._bands, device=x.device)
        _lib.check(_lib.lib().after_ae_pqmf_forward(h, _lib.ptr(x), _lib.ptr(mb), B, L,
                                                    _lib.current_stream(x.device)),
                   "after_ae_pqmf_forward")
        return mb

    @torch.no_grad()
    def pqmf_inverse(self, mb):
        mb = _lib.require_gpu_tensor(mb, "mb")
        B, M, Tm = mb.shape
        h = self._ensure(B, max(Tm * M, self.ratio))
        x = torch.empty(B, 1, Tm * M, device=mb.device)
        _lib.check(_lib.lib().after_ae_pqmf_inverse(h, _lib.ptr(mb), _lib.ptr(x), B, Tm,
                                                    _lib.current_stream(mb.device)),
                   "after_ae_pqmf_inverse")
        return x
