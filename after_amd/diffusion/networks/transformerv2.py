"""Host-side mirror of the reference's DenoiserV2
(after/diffusion/networks/transformerv2.py:460-543).

The module tree below exists to hold parameters under the reference's own
state-dict keys so that `load_state_dict` of an AFTER
checkpoint works unchanged; no torch op of these containers is ever executed.
`forward` hands device pointers to the HIP implementation
(`after_denoiser_forward`, include/after_hip.h)."""
import ctypes
from typing import Optional

import torch
from torch import nn

from ... import _lib


class _RotaryFreqs(nn.Module):
    """Parameter container for RotaryEmbedding(32).freqs (rotary_embedding.py:68,80).
    The values are fixed by the architecture; the HIP path rebuilds its cos/sin
    tables from the same formula, so the tensor is carried for key compatibility."""

    def __init__(self, dim: int = 32, theta: float = 10000.0):
        super().__init__()
        freqs = 1.0 / (theta**(torch.arange(0, dim, 2)[:dim // 2].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)


class _MHA(nn.Module):

    def __init__(self, rotary):
        super().__init__()
        self.rotary_emb = rotary


class _SelfAttention(nn.Module):

    def __init__(self, embed_dim, rotary):
        super().__init__()
        self.qkv_linear = nn.Linear(embed_dim, 3 * embed_dim, bias=False)
        self.mha = _MHA(rotary)
        self.rotary_emb = rotary


class _MLP(nn.Module):

    def __init__(self, embed_dim, mult, dropout):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(embed_dim, mult * embed_dim), nn.GELU(),
                                 nn.Linear(mult * embed_dim, embed_dim), nn.Dropout(dropout))


class _DecoderBlock(nn.Module):
    """transformerv2.py:299-335 (parameter layout only)."""

    def __init__(self, embed_dim, cond_dim, tcond_dim, mult, dropout, rotary):
        super().__init__()
        self.self_attention = _SelfAttention(embed_dim, rotary)
        self.mlp = _MLP(embed_dim, mult, dropout)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm3 = nn.LayerNorm(embed_dim)
        self.linear = nn.Linear(cond_dim, 2 * embed_dim)
        self.tcond_linear = nn.Linear(tcond_dim, 2 * embed_dim)


class _TransBlock(nn.Module):
    """transformerv2.py:365-431 (parameter layout only)."""

    def __init__(self, n_channels, seq_len, mult, embed_dim, tcond_dim, dropout, n_layers):
        super().__init__()
        self.patchify_and_embed = nn.Sequential(nn.Identity(), nn.Linear(n_channels, embed_dim),
                                                nn.GELU())
        self.patchify_and_embed_tcond = nn.Sequential(nn.Identity(),
                                                      nn.Linear(tcond_dim, tcond_dim), nn.GELU())
        self.rotary_emb = _RotaryFreqs(32)
        self.register_buffer("precomputed_pos_enc", torch.arange(0, seq_len).long())
        self.decoder_blocks = nn.ModuleList([
            _DecoderBlock(embed_dim, embed_dim, tcond_dim, mult, dropout, self.rotary_emb)
            for _ in range(n_layers)
        ])
        self.out_proj = nn.Sequential(nn.Linear(embed_dim, n_channels), nn.Identity())


class DenoiserV2(nn.Module):
    """Drop-in for the reference DenoiserV2 on MI355X (same constructor arguments,
    same forward signature, same state-dict keys)."""

    def __init__(self,
                 n_channels: int,
                 seq_len: int = 32,
                 embed_dim: int = 256,
                 cond_dim: int = 64,
                 tcond_dim: int = 0,
                 noise_embed_dims: int = 128,
                 n_layers: int = 6,
                 mlp_multiplier: int = 2,
                 dropout: float = 0.1,
                 causal: bool = False,
                 pos_emb_type="learnable",
                 local_attention_size: Optional[int] = None,
                 attention_chunk_size: int = 4):
        super().__init__()
        if pos_emb_type != "rotary":
            raise NotImplementedError(
                "after_amd builds the shipped configuration only: pos_emb_type='rotary' "
                f"(base.gin:76), got {pos_emb_type!r}")
        if cond_dim <= 0 or tcond_dim <= 0:
            raise NotImplementedError("cond_dim and tcond_dim must be > 0 (all shipped configs)")
        self.noise_embed_dims = noise_embed_dims
        self.embed_dim = embed_dim
        self.n_channels = n_channels
        self.cond_dim = cond_dim
        self.tcond_dim = tcond_dim
        self.n_layers = n_layers
        self.mlp_multiplier = mlp_multiplier
        self.causal = bool(causal)
        self.local_attention_size = local_attention_size
        self.attention_chunk_size = attention_chunk_size
        self.embedding = nn.Sequential(nn.Linear(cond_dim + noise_embed_dims, embed_dim), nn.GELU(),
                                       nn.Linear(embed_dim, embed_dim))
        self.denoiser_trans_block = _TransBlock(n_channels, seq_len, mlp_multiplier, embed_dim,
                                                tcond_dim, dropout, n_layers)
        self.requires_grad_(False)
        self._handle = None
        self._cap = (0, 0, 0)
        self._profile = False
        self._streaming = False
        self._stream_args = None  # (cache, steps, rows, frames) of the last enable_streaming_cache

    @property
    def name(self):
        return "transformer"

    # ------------------------------------------------------------ handle management
    def _apply(self, fn, *a, **k):
        self._release()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._release()
        return super().load_state_dict(*a, **k)

    def refresh(self):
        """Call after mutating parameters in place: the HIP handle owns a re-laid-out
        copy of the weights."""
        self._release()

    def _release(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            _lib.lib().after_denoiser_destroy(h)
        self._handle = None
        self._cap = (0, 0, 0)
        self._streaming = False

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _weights(self):
        sd = self.state_dict()
        P = "denoiser_trans_block."
        names = [
            "embedding.0.weight", "embedding.0.bias", "embedding.2.weight", "embedding.2.bias",
            P + "patchify_and_embed.1.weight", P + "patchify_and_embed.1.bias",
            P + "patchify_and_embed_tcond.1.weight", P + "patchify_and_embed_tcond.1.bias",
            P + "out_proj.0.weight", P + "out_proj.0.bias"
        ]
        for l in range(self.n_layers):
            B = f"{P}decoder_blocks.{l}."
            names += [
                B + "self_attention.qkv_linear.weight", B + "mlp.mlp.0.weight",
                B + "mlp.mlp.0.bias", B + "mlp.mlp.2.weight", B + "mlp.mlp.2.bias",
                B + "norm1.weight", B + "norm1.bias", B + "norm3.weight", B + "norm3.bias",
                B + "linear.weight", B + "linear.bias", B + "tcond_linear.weight",
                B + "tcond_linear.bias"
            ]
        return [_lib.require_gpu_tensor(sd[n], n) for n in names]

    def _ensure(self, rows: int, T: int, steps: int = 1):
        cr, ct, cs = self._cap
        if self._handle is not None and rows <= cr and T <= ct and steps <= cs:
            return self._handle
        if self._handle is not None and getattr(self, "_streaming", False):
            raise _lib.AFTERHipError(
                f"request (rows={rows}, T={T}) exceeds the capacity fixed when the streaming "
                f"caches were enabled {self._cap}; call enable_streaming_cache with larger limits")
        L = _lib.lib()
        self._release()
        cap = (max(rows, cr), max(T, ct), max(steps, cs, 1))
        if self._stream_args is not None and not getattr(self, "_enabling", False):
            _, st_steps, st_rows, st_frames = self._stream_args  # a rebuilt handle must hold the caches again
            cap = (max(cap[0], st_rows), max(cap[1], st_frames), max(cap[2], st_steps))
        ws = self._weights()
        arr = (ctypes.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
        cfg = _lib.DenoiserCfg(
            n_channels=self.n_channels, embed_dim=self.embed_dim, cond_dim=self.cond_dim,
            tcond_dim=self.tcond_dim, noise_embed_dims=self.noise_embed_dims,
            n_layers=self.n_layers, mlp_multiplier=self.mlp_multiplier, causal=int(self.causal),
            local_attention_size=-1 if self.local_attention_size is None else int(
                self.local_attention_size), attention_chunk_size=self.attention_chunk_size)
        out = ctypes.c_void_p()
        dev = next(w for w in ws if w is not None).device
        with torch.cuda.device(dev):
            torch.cuda.synchronize(dev)
            rc = L.after_denoiser_create(ctypes.byref(cfg), arr, len(ws), cap[0], cap[1], cap[2], ctypes.byref(out))
        _lib.check(rc, "after_denoiser_create")
        self._handle = out
        self._cap = cap
        if self._profile:  # measurement hooks survive a rebuilt handle (.to(), load_state_dict, refresh, growth)
            _lib.check(L.after_denoiser_profile_min_flops(out, float(getattr(self, "_profile_min_flops", 0.0))),
                       "after_denoiser_profile_min_flops")
            _lib.check(L.after_denoiser_profile_kernel(out, int(getattr(self, "_profile_kernel", 0))),
                       "after_denoiser_profile_kernel")
            _lib.check(L.after_denoiser_profile(out, 1), "after_denoiser_profile")
        if getattr(self, "_gemm_path", None) is not None:
            _lib.check(L.after_denoiser_set_gemm_path(out, *self._gemm_path), "after_denoiser_set_gemm_path")
        if getattr(self, "_stream_persist", None) is not None:
            _lib.check(L.after_denoiser_set_stream_persist(out, int(self._stream_persist)), "after_denoiser_set_stream_persist")
        if getattr(self, "_sample_persist", None) is not None:
            _lib.check(L.after_denoiser_set_sample_persist(out, int(self._sample_persist)), "after_denoiser_set_sample_persist")
        if getattr(self, "_persist_check", None) is not None:
            _lib.check(L.after_denoiser_set_persist_check(out, int(self._persist_check)), "after_denoiser_set_persist_check")
        if self._stream_args is not None and not getattr(self, "_enabling", False):
            # the handle was rebuilt (.to(), load_state_dict, refresh): a Streamer still expects
            # its K/V caches -- re-create them (zeroed = a new stream), as AutoEncoder / Encoder1D do
            cache, steps_, rows_, _ = self._stream_args
            _lib.check(L.after_denoiser_enable_cache(out, cache, steps_, rows_), "after_denoiser_enable_cache")
            self._streaming = True
        return out

    def reserve(self, rows: int, T: int, steps: int = 1):
        """Provision workspaces up front (rows = 3 x clips for a CFG sample)."""
        self._ensure(rows, T, steps)

    # ------------------------------------------------------------ streaming caches
    def enable_streaming_cache(self, max_cache_size: Optional[int] = None,
                               max_diffusion_steps: int = 16, max_batch_size: int = 4,
                               max_frames: int = 64):
        """Per-layer, per-diffusion-step K/V ring caches (transformerv2.py:143-155).  The
        reference enables them with the gin binding `MHAttention.max_cache_size =
        LOCAL_ATTENTION_SIZE` (after_scripts/export.py:77-79) and sizes them with
        `max_diffusion_steps` / `max_batch_size` (network rows: 3 x clips under CFG)."""
        cache = self.local_attention_size if max_cache_size is None else max_cache_size
        if cache is None or cache < 0 or not self.causal:
            raise ValueError("streaming K/V caches need causal attention with a finite local_attention_size "
                             "(export.py:73-79 binds max_cache_size = LOCAL_ATTENTION_SIZE)")
        # re-enabling with larger limits (a second Streamer on the same net) must be able to grow
        # the handle: the capacity lock of _ensure only guards forwards of an enabled stream
        self._streaming = False
        self._enabling = True
        try:
            self._ensure(max_batch_size, max_frames, int(max_diffusion_steps))
        finally:
            self._enabling = False
        self._cap = (max(self._cap[0], max_batch_size), max(self._cap[1], max_frames), self._cap[2])
        _lib.check(_lib.lib().after_denoiser_enable_cache(self._handle, int(cache),
                                                          int(max_diffusion_steps),
                                                          int(max_batch_size)),
                   "after_denoiser_enable_cache")
        self._streaming = True
        self._stream_args = (int(cache), int(max_diffusion_steps), int(max_batch_size), int(max_frames))

    def reset_cache(self):
        if self._handle is None:
            raise _lib.AFTERHipError("reset_cache before enable_streaming_cache")
        dev = self._device()  # the HANDLE's device and its current stream, not the process's current device
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().after_denoiser_reset_cache(self._handle, _lib.current_stream(dev)),
                       "after_denoiser_reset_cache")

    # ------------------------------------------------------------ reference surface
    def roll_cache(self, size: int, cache_index: int):
        """transformerv2.py:514-515."""
        if self._handle is None:
            raise _lib.AFTERHipError("roll_cache before any forward")
        dev = self._device()
        with torch.cuda.device(dev):
            _lib.check(
                _lib.lib().after_denoiser_roll_cache(self._handle, int(size), int(cache_index),
                                                     _lib.current_stream(dev)),
                "after_denoiser_roll_cache")

    @torch.no_grad()
    def forward(self,
                x,
                time: torch.Tensor,
                cond: Optional[torch.Tensor] = None,
                time_cond: Optional[torch.Tensor] = None,
                cache_index: int = 0) -> torch.Tensor:
        """transformerv2.py:517-543."""
        if cond is None or time_cond is None:
            raise ValueError("cond and time_cond are required (cond_dim, tcond_dim > 0)")
        x = _lib.require_gpu_tensor(x, "x")
        b, C, T = x.shape
        if C != self.n_channels:
            raise ValueError(f"x has {C} channels, the denoiser expects {self.n_channels}")
        if len(time.shape) > 1:  # :525-528
            time = time[..., 0]
        time = _lib.require_gpu_tensor(time.reshape(-1).to(x.device), "time")
        if time.numel() != b:
            raise ValueError(f"time has {time.numel()} entries for batch {b}")
        cond = _lib.require_gpu_tensor(cond, "cond")
        time_cond = _lib.require_gpu_tensor(time_cond, "time_cond")
        if tuple(cond.shape) != (b, self.cond_dim) or tuple(time_cond.shape) != (b, self.tcond_dim, T):
            raise ValueError(f"bad conditioning shapes {tuple(cond.shape)} / {tuple(time_cond.shape)}")
        h = self._ensure(b, T, 1)
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(
                _lib.lib().after_denoiser_forward(h, _lib.ptr(x), _lib.ptr(time), _lib.ptr(cond),
                                                  _lib.ptr(time_cond), _lib.ptr(out), b, T,
                                                  int(cache_index), _lib.current_stream(x.device)),
                "after_denoiser_forward")
        return out

    # ------------------------------------------------------------ sampler back end
    def _check_cfg_inputs(self, x, cond, time_cond):
        x = _lib.require_gpu_tensor(x, "x")
        B, C, T = x.shape
        cond = _lib.require_gpu_tensor(cond.to(x.device), "cond")
        time_cond = _lib.require_gpu_tensor(time_cond.to(x.device), "time_cond")
        if C != self.n_channels or tuple(cond.shape) != (B, self.cond_dim) or \
                tuple(time_cond.shape) != (B, self.tcond_dim, T):
            raise ValueError(f"bad shapes x{tuple(x.shape)} cond{tuple(cond.shape)} "
                             f"time_cond{tuple(time_cond.shape)}")
        return x, cond, time_cond, B, T

    @torch.no_grad()
    def cfg_forward(self, x, time, cond, time_cond, guidance_timbre, guidance_structure,
                    drop_value, cfg_mode=_lib.CFG_API, cache_index=0):
        x, cond, time_cond, B, T = self._check_cfg_inputs(x, cond, time_cond)
        if len(time.shape) > 1:
            time = time[..., 0]
        time = _lib.require_gpu_tensor(time.reshape(-1).to(x.device), "time")
        if time.numel() != B:
            raise ValueError(f"time has {time.numel()} entries for batch {B}")
        h = self._ensure(3 * B, T, 1)
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(
                _lib.lib().after_model_forward(h, _lib.ptr(x), _lib.ptr(time), _lib.ptr(cond),
                                               _lib.ptr(time_cond), _lib.ptr(out), B, T,
                                               float(guidance_timbre), float(guidance_structure),
                                               float(drop_value), int(cfg_mode), int(cache_index),
                                               _lib.current_stream(x.device)),
                "after_model_forward")
        return out

    @torch.no_grad()
    def cfg_sample(self, x0, cond, time_cond, nb_steps, guidance_timbre, guidance_structure,
                   drop_value, cfg_mode=_lib.CFG_API, out=None):
        """RectifiedFlow.sample (model.py:763-785) as one call of after_sample on x0's current stream.

        Host synchronisation: when the call is served by a persistent kernel (one base / tiny clip, or >= 5 base clips) and the
        handle is in its default checked mode (`set_persist_check(None)`), after_sample waits on the host for that launch's failure
        words (one event wait behind the kernel) so that a refused / failed launch is served by the launch path within this call:
        the call returns with the sampler finished on the device.  `set_persist_check(False)` restores fully asynchronous
        enqueue (a failure then surfaces at the next call or in `check()`); calls served by launches never synchronise."""
        x0, cond, time_cond, B, T = self._check_cfg_inputs(x0, cond, time_cond)
        h = self._ensure(3 * B, T, int(nb_steps))
        if out is None:
            out = torch.empty_like(x0)
        with torch.cuda.device(x0.device):
            _lib.check(
                _lib.lib().after_sample(h, _lib.ptr(x0), _lib.ptr(cond), _lib.ptr(time_cond),
                                        _lib.ptr(out), B, T, int(nb_steps),
                                        float(guidance_timbre), float(guidance_structure),
                                        float(drop_value), int(cfg_mode),
                                        _lib.current_stream(x0.device)), "after_sample")
        return out

    # ------------------------------------------------------------ measurement hooks
    def set_gemm_path(self, mode: int, min_rows: int = 0):
        """Arithmetic path of the qkv / MLP Linears (include/after_hip.h: after_denoiser_set_gemm_path):
        0 fp32 MFMA; 1 bf16-split kernel for >= min_rows token rows (the default); 2 bf16-split always; 3 the OPT-IN bf16
        tolerance tier (latents within 5e-2 abs / 1e-2 rel-L2 of the fp32 reference): path selection
        as 1, and the persistent offline samplers issue ONE bf16 MFMA per product block -- the operands' top bf16 planes, fp32
        accumulate; everything else (LayerNorm, attention, RoPE, GELU, the sampler tail) stays fp32.  Never the default."""
        self._gemm_path = (int(mode), int(min_rows))
        if self._handle is not None:
            _lib.check(_lib.lib().after_denoiser_set_gemm_path(self._handle, int(mode), int(min_rows)),
                       "after_denoiser_set_gemm_path")

    def set_stream_persist(self, enable: bool):
        """Streaming sampler (cfg_sample on a handle with K/V caches): one persistent launch per Euler step where the
        geometry allows it (include/after_hip.h: after_denoiser_set_stream_persist), or the launch-per-kernel path."""
        self._stream_persist = bool(enable)
        if self._handle is not None:
            _lib.check(_lib.lib().after_denoiser_set_stream_persist(self._handle, int(bool(enable))),
                       "after_denoiser_set_stream_persist")

    def set_sample_persist(self, enable: bool):
        """Offline cfg_sample of ONE clip as one persistent launch (include/after_hip.h: after_denoiser_set_sample_persist)
        where the geometry allows it, or the launch-per-kernel path."""
        self._sample_persist = bool(enable)
        if self._handle is not None:
            _lib.check(_lib.lib().after_denoiser_set_sample_persist(self._handle, int(bool(enable))),
                       "after_denoiser_set_sample_persist")

    def set_persist_check(self, enable):
        """Persistent samplers, when a launch's failure words are looked at (include/after_hip.h: after_denoiser_set_persist_check).
        None (the default): the stateless offline samplers synchronise their stream and look at once -- a refused or failed launch
        is served by launches within the same cfg_sample call, which therefore never returns an untouched tensor -- and the
        streaming sampler defers to the next call / check().  True: every persistent call synchronises and reports its own
        failure (streaming: AFTERHipError from that very call).  False: every persistent call defers (no host synchronisation in
        cfg_sample; a failure surfaces in the next call or in check())."""
        self._persist_check = -1 if enable is None else int(bool(enable))
        if self._handle is not None:
            _lib.check(_lib.lib().after_denoiser_set_persist_check(self._handle, self._persist_check), "after_denoiser_set_persist_check")

    def _device(self):
        return next(self.parameters()).device

    def check(self):
        """Waits for the last persistent launch's failure words and raises if a persistent sampler failed since the last look
        (include/after_hip.h: after_denoiser_check): the way to validate the LAST chunk of a stream."""
        if self._handle is not None:
            dev = self._device()
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().after_denoiser_check(self._handle, _lib.current_stream(dev)), "after_denoiser_check")

    def sample_persist(self) -> bool:
        """True when the last cfg_sample of this handle ran as the persistent offline kernel."""
        if self._handle is None:
            return False
        a = ctypes.c_int()
        _lib.check(_lib.lib().after_denoiser_sample_persist(self._handle, ctypes.byref(a)), "after_denoiser_sample_persist")
        return bool(a.value)

    def sample_path(self) -> int:
        """How the last cfg_sample of this handle ran: 0 by launches, 1 the one-clip persistent kernel (time segments over the
        XCDs), 2 the batch persistent kernel (one clip per XCD)."""
        if self._handle is None:
            return 0
        a = ctypes.c_int()
        _lib.check(_lib.lib().after_denoiser_sample_persist(self._handle, ctypes.byref(a)), "after_denoiser_sample_persist")
        return int(a.value)

    def sample_launches(self) -> int:
        """Persistent launches the last cfg_sample took (0: per-op launches).  The one-clip kernel takes two clips per launch."""
        if self._handle is None:
            return 0
        a = ctypes.c_int()
        _lib.check(_lib.lib().after_denoiser_sample_launches(self._handle, ctypes.byref(a)), "after_denoiser_sample_launches")
        return int(a.value)

    def sample_arith(self) -> int:
        """The arithmetic of the qkv / MLP Linears in the last cfg_sample (include/after_hip.h: after_denoiser_sample_arith): 0 the fp32
        MFMA chain, 1 three bf16 planes x six products, 2 two fp16 pieces x three products (the persistent samplers' default),
        3 the opt-in bf16 tolerance tier."""
        if self._handle is None:
            return 0
        a = ctypes.c_int()
        _lib.check(_lib.lib().after_denoiser_sample_arith(self._handle, ctypes.byref(a)), "after_denoiser_sample_arith")
        return int(a.value)

    def stream_persist(self) -> bool:
        """True when the last streaming cfg_sample of this handle ran (and the next one of the same shape will run) as
        persistent launches."""
        if self._handle is None:
            return False
        a = ctypes.c_int()
        _lib.check(_lib.lib().after_denoiser_stream_persist(self._handle, ctypes.byref(a)), "after_denoiser_stream_persist")
        return bool(a.value)

    def gemm_path(self):
        """(mode, min_rows) in effect (the handle's, i.e. including the AFTER_GEMM_X6 environment default)."""
        if self._handle is None:
            self._ensure(1, 1, 1)
        m, r = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().after_denoiser_gemm_path(self._handle, ctypes.byref(m), ctypes.byref(r)),
                   "after_denoiser_gemm_path")
        return m.value, r.value

    def profile(self, enable: bool = True, min_flops: float = 0.0, kernel: int = 0):
        """Bracket the GEMM launches with HIP events: only those of >= min_flops, and of kernel class
        `kernel` (0 both, 1 the bf16-split gemm_x6 kernel, 2 the fp32 MFMA kernel)."""
        self._profile = bool(enable)
        self._profile_min_flops = float(min_flops)
        self._profile_kernel = int(kernel)
        if self._handle is not None:
            _lib.check(_lib.lib().after_denoiser_profile_kernel(self._handle, int(kernel)),
                       "after_denoiser_profile_kernel")
            _lib.check(_lib.lib().after_denoiser_profile_min_flops(self._handle, float(min_flops)),
                       "after_denoiser_profile_min_flops")
            _lib.check(_lib.lib().after_denoiser_profile(self._handle, int(enable)),
                       "after_denoiser_profile")

    def set_step_trace(self, enable: bool):
        """Persistent samplers: per-phase wall-clock stamps of every workgroup (include/after_hip.h: after_denoiser_set_step_trace)."""
        if self._handle is None:
            raise _lib.AFTERHipError("set_step_trace: no handle yet (reserve() or one forward first)")
        _lib.check(_lib.lib().after_denoiser_set_step_trace(self._handle, int(bool(enable))), "after_denoiser_set_step_trace")

    def step_trace(self):
        """numpy [workgroups, 128] uint64: the stamps of the last persistent Euler step (after_denoiser_step_trace)."""
        import numpy as np
        if self._handle is None:
            raise _lib.AFTERHipError("step_trace: no handle yet")
        n = torch.cuda.get_device_properties(self._device()).multi_processor_count
        buf = np.zeros((n, 128), dtype=np.uint64)
        _lib.check(_lib.lib().after_denoiser_step_trace(self._handle, buf.ctypes.data_as(ctypes.c_void_p), n), "after_denoiser_step_trace")
        return buf

    def gemm_time(self, with_bytes: bool = False):
        """(total_ms, launches, flops[, algorithmic bytes]) of the bracketed GEMM launches since the
        last call; synchronise the stream first."""
        ms, n, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
        _lib.check(
            _lib.lib().after_denoiser_gemm_time2(self._handle, ctypes.byref(ms), ctypes.byref(n),
                                                 ctypes.byref(fl), ctypes.byref(by)), "after_denoiser_gemm_time2")
        return (ms.value, n.value, fl.value, by.value) if with_bytes else (ms.value, n.value, fl.value)
