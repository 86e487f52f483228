"""UNET1D: drop-in container for the reference's Conv1d / GroupNorm / SiLU / FiLM denoiser
(after/diffusion/networks/unet1d.py:254-429).  Same constructor arguments and state-dict keys;
`forward` runs in libafter_hip (after_unet1d_*).  Built for time_cond_channels > 0, cond_channels > 0, with or
without the self-attention layers of n_attn_layers > 0 (blocks.py:201-243); no shipped gin config selects
this network -- it is SURVEY 8(f)-4."""
import ctypes

import torch
import torch.nn as nn

from ... import _lib


def _groups(c):
    return min(16, c // 4)


class _ConvBlock(nn.Module):  # ConvBlock1D, unet1d.py:29-118

    def __init__(self, in_c, out_c, skip_c, tc_c, time_c, cond_c, k):
        super().__init__()
        cat = in_c + skip_c + tc_c
        self.conv1 = nn.Conv1d(cat, out_c, k, padding="same")
        self.gn1 = nn.GroupNorm(_groups(cat), cat)
        self.conv2 = nn.Conv1d(out_c, out_c, k, padding="same")
        self.gn2 = nn.GroupNorm(_groups(out_c), out_c)
        self.act = nn.SiLU()
        self.time_mlp = nn.Sequential(nn.Linear(time_c, 128), nn.SiLU(), nn.Linear(128, 2 * out_c))
        self.cond_mlp = nn.Sequential(nn.Linear(cond_c, 128), nn.SiLU(), nn.Linear(128, 2 * out_c))
        self.to_out = nn.Conv1d(in_c, out_c, 1, padding="same") if skip_c else nn.Identity()


class _SelfAttn(nn.Module):  # SelfAttention1d, blocks.py:201-243 (parameters only: the arithmetic is after_unet1d_forward's)

    def __init__(self, c, n_head):
        super().__init__()
        if n_head < 1 or c % n_head:
            raise ValueError(f"SelfAttention1d({c}, {n_head}): heads must divide the channels")
        self.n_head = n_head
        self.norm = nn.GroupNorm(1, c)
        self.qkv_proj = nn.Conv1d(c, 3 * c, 1)
        self.out_proj = nn.Conv1d(c, c, 1)


class _EncBlock(nn.Module):  # EncoderBlock1D, :121-165

    def __init__(self, in_c, out_c, tc_c, time_c, cond_c, k, ratio, use_self_attn=False):
        super().__init__()
        self.conv = _ConvBlock(in_c, in_c, 0, tc_c, time_c, cond_c, k)
        self.self_attn = _SelfAttn(in_c, 4) if use_self_attn else nn.Identity()
        self.pool = nn.Conv1d(in_c, out_c, k, padding="same") if ratio == 1 else \
            nn.Conv1d(in_c, out_c, k, stride=ratio, padding=k // 2)


class _MidBlock(nn.Module):  # MiddleBlock1D, :168-197

    def __init__(self, in_c, tc_c, time_c, cond_c, k, use_self_attn=False):
        super().__init__()
        self.conv = _ConvBlock(in_c, in_c, 0, tc_c, time_c, cond_c, k)
        self.self_attn = _SelfAttn(in_c, in_c // 32) if use_self_attn else nn.Identity()


class _DecBlock(nn.Module):  # DecoderBlock1D, :200-251

    def __init__(self, in_c, out_c, tc_c, time_c, cond_c, k, ratio, skip_size=None, use_self_attn=False):
        super().__init__()
        if ratio == 1:
            self.up = nn.Identity() if in_c == out_c else nn.Conv1d(in_c, out_c, 3, padding="same")
        else:
            self.up = nn.Sequential(nn.Upsample(mode="nearest", scale_factor=ratio),
                                    nn.Conv1d(in_c, out_c, 3, padding="same"))
        self.conv = _ConvBlock(out_c, out_c, skip_size if skip_size is not None else out_c, tc_c, time_c,
                               cond_c, k)
        self.self_attn = _SelfAttn(out_c, 4) if use_self_attn else nn.Identity()


class UNET1D(nn.Module):

    def __init__(self, in_size=128, out_size=None, channels=(128, 128, 256, 256), ratios=(2, 2, 2, 2, 2),
                 kernel_size=5, time_channels=64, time_cond_in_channels=1, time_cond_channels=64,
                 cond_channels=32, n_attn_layers=0, use_res_last=False):
        super().__init__()
        if not time_cond_channels or not cond_channels or not time_channels:
            raise NotImplementedError("after_amd builds UNET1D with time_cond_channels > 0, cond_channels > 0, "
                                      "time_channels > 0")
        if not 0 <= n_attn_layers <= len(channels):
            raise ValueError(f"n_attn_layers = {n_attn_layers} with {len(channels)} levels")
        if kernel_size % 2 == 0:
            raise NotImplementedError("odd kernel_size only (padding='same')")
        channels = list(channels)
        self.channels = channels
        self.in_size = in_size
        self.out_size = in_size if out_size is None else out_size
        self.kernel_size = kernel_size
        self.time_channels, self.time_cond_in_channels = time_channels, time_cond_in_channels
        self.time_cond_channels, self.cond_channels = time_cond_channels, cond_channels
        self.use_res_last = bool(use_res_last)
        self.n_attn_layers = na = int(n_attn_layers)
        n = len(channels)
        R = [1] + list(ratios)
        self.ratios = R[:n]
        k, tcc, tc, cc = kernel_size, time_cond_channels, time_channels, cond_channels
        mods = [nn.Sequential(nn.Conv1d(time_cond_in_channels, tcc, k, padding="same"), nn.SiLU())]
        for i in range(n):
            mods.append(nn.Sequential(
                nn.Conv1d(tcc, tcc, k, stride=R[i], padding="same" if R[i] == 1 else k // 2), nn.SiLU()))
        self.cond_emb_time = nn.ModuleList(mods)
        self.up_layers = nn.ModuleList()
        self.down_layers = nn.ModuleList([_EncBlock(in_size, channels[0], tcc, tc, cc, k, R[0])])
        for i in range(1, n):
            self.down_layers.append(_EncBlock(channels[i - 1], channels[i], tcc, tc, cc, k, R[i],
                                              use_self_attn=i >= n - na))  # unet1d.py:339
            self.up_layers.append(_DecBlock(channels[n - i], channels[n - i - 1], tcc, tc, cc, k, R[n - i],
                                            use_self_attn=i <= na))  # :350
        self.up_layers.append(_DecBlock(channels[0], self.out_size, tcc, tc, cc, k, R[0], skip_size=in_size))
        self.middle_block = _MidBlock(channels[-1], tcc, tc, cc, k, use_self_attn=na > 0)  # :372
        self.total_ratio = 1
        for r in self.ratios:
            self.total_ratio *= r
        self.requires_grad_(False)
        self.eval()
        self._handle = None
        self._cap = (0, 0)

    @property
    def name(self):
        return "unet"

    # ------------------------------------------------------------ handle management
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
            _lib.lib().after_unet1d_destroy(h)
        self._handle = None
        self._cap = (0, 0)

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _weight_names(self):
        def CB(p, skip):
            names = [f"{p}.{m}.{t}" for m in ("conv1", "gn1", "conv2", "gn2") for t in ("weight", "bias")]
            names += [f"{p}.{m}.{i}.{t}" for m in ("time_mlp", "cond_mlp") for i in (0, 2) for t in ("weight", "bias")]
            return names + ([f"{p}.to_out.weight", f"{p}.to_out.bias"] if skip else [])

        def SA(p, layer):
            if isinstance(layer.self_attn, nn.Identity):
                return []
            return [f"{p}.self_attn.{m}.{t}" for m in ("norm", "qkv_proj", "out_proj") for t in ("weight", "bias")]

        n = len(self.channels)
        names = [f"cond_emb_time.{i}.0.{t}" for i in range(n + 1) for t in ("weight", "bias")]
        for i in range(n):
            names += CB(f"down_layers.{i}.conv", False) + SA(f"down_layers.{i}", self.down_layers[i])
            names += [f"down_layers.{i}.pool.weight", f"down_layers.{i}.pool.bias"]
        names += CB("middle_block.conv", False) + SA("middle_block", self.middle_block)
        for j, layer in enumerate(self.up_layers):
            if isinstance(layer.up, nn.Identity):
                names += [None, None]
            elif isinstance(layer.up, nn.Sequential):
                names += [f"up_layers.{j}.up.1.weight", f"up_layers.{j}.up.1.bias"]
            else:
                names += [f"up_layers.{j}.up.weight", f"up_layers.{j}.up.bias"]
            names += CB(f"up_layers.{j}.conv", True) + SA(f"up_layers.{j}", layer)
        return names

    def _ensure(self, B, T):
        cb, ct = self._cap
        if self._handle is not None and B <= cb and T <= ct:
            return self._handle
        L = _lib.lib()
        self._release()
        cap = (max(B, cb), max(T, ct))
        sd = self.state_dict()
        ws = [None if nm is None else _lib.require_gpu_tensor(sd[nm], nm) for nm in self._weight_names()]
        arr = (ctypes.c_void_p * len(ws))(*[None if w is None else w.data_ptr() for w in ws])
        cfg = _lib.Unet1dCfg()
        cfg.in_size, cfg.out_size, cfg.n_blocks = self.in_size, self.out_size, len(self.channels)
        for i, c in enumerate(self.channels):
            cfg.channels[i] = c
        for i, r in enumerate(self.ratios):
            cfg.ratios[i] = r
        cfg.kernel_size = self.kernel_size
        cfg.time_channels, cfg.time_cond_in_channels = self.time_channels, self.time_cond_in_channels
        cfg.time_cond_channels, cfg.cond_channels = self.time_cond_channels, self.cond_channels
        cfg.use_res_last = int(self.use_res_last)
        cfg.n_attn_layers = self.n_attn_layers
        out = ctypes.c_void_p()
        dev = next(w for w in ws if w is not None).device
        with torch.cuda.device(dev):
            torch.cuda.synchronize(dev)
            rc = L.after_unet1d_create(ctypes.byref(cfg), arr, len(ws), cap[0], cap[1], ctypes.byref(out))
        _lib.check(rc, "after_unet1d_create")
        self._handle = out
        self._cap = cap
        return out

    @torch.no_grad()
    def forward(self, x, time=None, time_cond=None, cond=None, time_emb=None, cache_index: int = 0):
        """unet1d.py:374-414."""
        if time_emb is not None:
            raise NotImplementedError("precomputed time_emb is not supported")
        x = _lib.require_gpu_tensor(x, "x")
        B, C, T = x.shape
        if C != self.in_size:
            raise ValueError(f"expected {self.in_size} channels, got {C}")
        if T % self.total_ratio:
            raise ValueError(f"T={T} is not a multiple of the total ratio {self.total_ratio}")
        time = _lib.require_gpu_tensor(time.reshape(-1).contiguous(), "time")
        cond = _lib.require_gpu_tensor(cond, "cond")
        time_cond = _lib.require_gpu_tensor(time_cond, "time_cond")
        if time.numel() != B or tuple(cond.shape) != (B, self.cond_channels) or \
                tuple(time_cond.shape) != (B, self.time_cond_in_channels, T):
            raise ValueError("time [B], cond [B, cond_channels], time_cond [B, time_cond_in_channels, T] expected")
        h = self._ensure(B, T)
        out = torch.empty(B, self.out_size, T, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().after_unet1d_forward(h, _lib.ptr(x), _lib.ptr(time), _lib.ptr(cond),
                                                       _lib.ptr(time_cond), _lib.ptr(out), B, T,
                                                       _lib.current_stream(x.device)), "after_unet1d_forward")
        return out

    # ------------------------------------------------------------ sampler back end (RectifiedFlow)
    def _check_cfg_inputs(self, x, cond, time_cond):
        x = _lib.require_gpu_tensor(x, "x")
        B, C, T = x.shape
        cond = _lib.require_gpu_tensor(cond.to(x.device), "cond")
        time_cond = _lib.require_gpu_tensor(time_cond.to(x.device), "time_cond")
        if C != self.in_size or self.out_size != self.in_size or T % self.total_ratio or \
                tuple(cond.shape) != (B, self.cond_channels) or \
                tuple(time_cond.shape) != (B, self.time_cond_in_channels, T):
            raise ValueError(f"bad shapes x{tuple(x.shape)} cond{tuple(cond.shape)} time_cond{tuple(time_cond.shape)}")
        return x, cond, time_cond, B, T

    @torch.no_grad()
    def cfg_forward(self, x, time, cond, time_cond, guidance_timbre, guidance_structure, drop_value,
                    cfg_mode=_lib.CFG_API, cache_index=0):
        """RectifiedFlow.model_forward (model.py:721-761) on the device (after_unet1d_model_forward)."""
        x, cond, time_cond, B, T = self._check_cfg_inputs(x, cond, time_cond)
        time = _lib.require_gpu_tensor(time.reshape(-1).to(x.device), "time")
        if time.numel() != B:
            raise ValueError(f"time has {time.numel()} entries for batch {B}")
        h = self._ensure(3 * B, T)
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().after_unet1d_model_forward(
                h, _lib.ptr(x), _lib.ptr(time), _lib.ptr(cond), _lib.ptr(time_cond), _lib.ptr(out), B, T,
                float(guidance_timbre), float(guidance_structure), float(drop_value), int(cfg_mode),
                _lib.current_stream(x.device)), "after_unet1d_model_forward")
        return out

    @torch.no_grad()
    def cfg_sample(self, x0, cond, time_cond, nb_steps, guidance_timbre, guidance_structure, drop_value,
                   cfg_mode=_lib.CFG_API, out=None):
        """RectifiedFlow.sample (model.py:763-785): the whole Euler loop inside after_unet1d_sample."""
        x0, cond, time_cond, B, T = self._check_cfg_inputs(x0, cond, time_cond)
        h = self._ensure(3 * B, T)
        if out is None:
            out = torch.empty_like(x0)
        with torch.cuda.device(x0.device):
            _lib.check(_lib.lib().after_unet1d_sample(
                h, _lib.ptr(x0), _lib.ptr(cond), _lib.ptr(time_cond), _lib.ptr(out), B, T, int(nb_steps),
                float(guidance_timbre), float(guidance_structure), float(drop_value), int(cfg_mode),
                _lib.current_stream(x0.device)), "after_unet1d_sample")
        return out
