"""Oracle: DenoiserV2 forward (reference after/diffusion/networks/transformerv2.py).

Test infrastructure -- see oracle/__init__.py.  Functional restatement on a
reference-format state dict `sd`; `cfg` is a dict with the DenoiserV2
constructor arguments.
"""
import math
from typing import Optional

import torch
import torch.nn.functional as F

P = "denoiser_trans_block."


def positional_embedding(t: torch.Tensor, num_channels: int = 64,
                         max_positions: float = 10_000.0, factor: float = 100.0):
    """transformerv2.py:31-43 (PositionalEmbedding.forward, endpoint=False)."""
    x = t.reshape(-1) * factor
    half = num_channels // 2
    freqs = torch.arange(half, dtype=torch.float32) / half
    freqs = (1.0 / max_positions)**freqs
    x = torch.outer(x, freqs.to(x.dtype))
    return torch.cat([x.cos(), x.sin()], dim=1)


def band_bounds(j: int, chunk: int, window: Optional[int], k_len: int):
    """Allowed key range [lo, hi) of absolute key row j.

    transformerv2.py:62-96 (combined_sliding_chunkwise_mask) and :46-59
    (chunk_wise_causal_mask when window is None / negative): row j of chunk
    [i, e) sees [i, e) plus [max(0, j-window+1), i)."""
    i = (j // chunk) * chunk
    e = min(i + chunk, k_len)
    if window is None or window < 0:
        lo = 0
    else:
        lo = min(i, max(0, j - window + 1))
    return lo, e


def rope_tables(n_pos: int, rot_dim: int = 32, theta: float = 10000.0,
                dtype=torch.float32):
    """rotary_embedding.py:68 (freqs), :350-355 (angles, each repeated twice
    interleaved).  Returns cos, sin of shape [n_pos, rot_dim]."""
    freqs = 1.0 / (theta**(torch.arange(0, rot_dim, 2)[:rot_dim // 2].float() / rot_dim))
    ang = torch.outer(torch.arange(n_pos, dtype=torch.float32), freqs)
    ang = ang.repeat_interleave(2, dim=1)
    return ang.cos().to(dtype), ang.sin().to(dtype)


def _rotate_half(x):
    """rotary_embedding.py:132-140: adjacent pairs (x0, x1) -> (-x1, x0)."""
    x = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)


def _apply_rope(t, cos, sin):
    """rotary_embedding.py:143-173 with start_index 0, scale 1."""
    rot = cos.shape[-1]
    mid = t[..., :rot]
    mid = mid * cos + _rotate_half(mid) * sin
    return torch.cat([mid, t[..., rot:]], dim=-1)


class DenoiserCache:
    """Per-layer, per-diffusion-step K/V ring caches of the streaming path
    (transformerv2.py:143-204).  k/v: [n_layers][max_batch, max_steps, H, C, Dh],
    zero-initialised like the reference buffers."""

    def __init__(self, n_layers, max_batch, max_steps, heads, cache, dh,
                 dtype=torch.float32):
        z = lambda: torch.zeros(max_batch, max_steps, heads, cache, dh, dtype=dtype)
        self.k = [z() for _ in range(n_layers)]
        self.v = [z() for _ in range(n_layers)]
        self.last_k = [None] * n_layers
        self.last_v = [None] * n_layers
        self.cache = cache

    def roll(self, roll_size: int, cache_index: int):
        """transformerv2.py:171-188 (MHAttention.roll_cache)."""
        for l in range(len(self.k)):
            lk, lv = self.last_k[l], self.last_v[l]
            b = lk.shape[0]
            kc = torch.cat([self.k[l][:b, cache_index], lk[:, :, :roll_size]], dim=2)
            vc = torch.cat([self.v[l][:b, cache_index], lv[:, :, :roll_size]], dim=2)
            if kc.shape[2] > self.cache:
                kc = kc[:, :, -self.cache:]
                vc = vc[:, :, -self.cache:]
            self.k[l][:b, cache_index] = kc
            self.v[l][:b, cache_index] = vc


def banded_attention_loop(q, k, v, chunk: int, window: Optional[int], causal: bool = True,
                          rot_dim: int = 32):
    """Per-query restatement (kept as the cross-check of banded_attention)."""
    b, h, qn, dh = q.shape
    kn = k.shape[2]
    off = kn - qn
    cos, sin = rope_tables(kn, rot_dim, dtype=q.dtype)
    qr = _apply_rope(q, cos[off:], sin[off:])
    kr = _apply_rope(k, cos, sin)
    out = torch.empty_like(q)
    scale = 1.0 / math.sqrt(dh)
    for j in range(qn):
        if causal:
            lo, hi = band_bounds(j + off, chunk, window, kn)
        else:
            lo, hi = 0, kn
        s = torch.einsum("bhd,bhkd->bhk", qr[:, :, j], kr[:, :, lo:hi]) * scale
        p = torch.softmax(s, dim=-1)
        out[:, :, j] = torch.einsum("bhk,bhkd->bhd", p, v[:, :, lo:hi])
    return out


def banded_attention(q, k, v, chunk: int, window: Optional[int], causal: bool = True,
                     rot_dim: int = 32):
    """transformerv2.py:190-236 (MHAttention.forward) on [b, H, n, Dh] tensors,
    restated WITHOUT the dense mask: each query gathers only its allowed keys
    (band_bounds), RoPE positions follow rotary_embedding.py:215-236 (queries
    offset by k_len - q_len).  Vectorised over queries: key slot m of query j is
    key lo(j) + m, slots past hi(j) are masked."""
    b, h, qn, dh = q.shape
    kn = k.shape[2]
    off = kn - qn
    if not causal or window is None or window < 0:
        return banded_attention_loop(q, k, v, chunk, window, causal, rot_dim)
    cos, sin = rope_tables(kn, rot_dim, dtype=q.dtype)
    qr = _apply_rope(q, cos[off:], sin[off:])
    kr = _apply_rope(k, cos, sin)
    bounds = [band_bounds(j + off, chunk, window, kn) for j in range(qn)]
    lo = torch.tensor([x[0] for x in bounds])
    hi = torch.tensor([x[1] for x in bounds])
    nk = int((hi - lo).max())
    idx = lo[:, None] + torch.arange(nk)[None, :]           # [qn, nk]
    valid = idx < hi[:, None]
    idx = idx.clamp(max=kn - 1)
    kg = kr[:, :, idx]                                       # [b, h, qn, nk, dh]
    vg = v[:, :, idx]
    s = torch.einsum("bhqd,bhqkd->bhqk", qr, kg) * (1.0 / math.sqrt(dh))
    s = s.masked_fill(~valid, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.einsum("bhqk,bhqkd->bhqd", p, vg)


def _ln(x, w=None, b=None, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1], ), w, b, eps)


def decoder_block(sd, pre, x, features, tcond, cfg, layer, cache: Optional[DenoiserCache],
                  cache_index: int):
    """transformerv2.py:340-362 (DecoderBlock.forward)."""
    E = cfg["embed_dim"]
    H = E // 64
    if cfg["tcond_dim"] > 0:
        x = _ln(x)
        ab = F.linear(tcond, sd[pre + "tcond_linear.weight"], sd[pre + "tcond_linear.bias"])
        alpha, beta = ab.chunk(2, dim=-1)
        x = x * (1 + alpha) + beta
    h = _ln(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    qkv = F.linear(h, sd[pre + "self_attention.qkv_linear.weight"])
    q, k, v = qkv.chunk(3, dim=2)
    sp = lambda t: t.reshape(t.shape[0], t.shape[1], H, 64).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    if cache is not None:
        b = k.shape[0]
        cache.last_k[layer], cache.last_v[layer] = k, v
        k = torch.cat([cache.k[layer][:b, cache_index], k], dim=2)
        v = torch.cat([cache.v[layer][:b, cache_index], v], dim=2)
    a = banded_attention(q, k, v, cfg["attention_chunk_size"], cfg["local_attention_size"],
                         cfg["causal"])
    a = a.permute(0, 2, 1, 3).reshape(x.shape)
    x = a + x
    if cfg["cond_dim"] > 0:
        x = _ln(x)
        ab = F.linear(features, sd[pre + "linear.weight"], sd[pre + "linear.bias"])
        alpha, beta = ab.chunk(2, dim=-1)
        x = x * (1 + alpha.unsqueeze(1)) + beta.unsqueeze(1)
    h = _ln(x, sd[pre + "norm3.weight"], sd[pre + "norm3.bias"])
    h = F.gelu(F.linear(h, sd[pre + "mlp.mlp.0.weight"], sd[pre + "mlp.mlp.0.bias"]))
    h = F.linear(h, sd[pre + "mlp.mlp.2.weight"], sd[pre + "mlp.mlp.2.bias"])
    return h + x


def denoiser_forward(sd, cfg, x, time, cond=None, time_cond=None,
                     cache: Optional[DenoiserCache] = None, cache_index: int = 0):
    """transformerv2.py:517-543 (DenoiserV2.forward) + :437-457
    (DenoiserTransBlock.forward).  x [b, C, T]; time [b] | [b,1,1] | [b,1,T];
    cond [b, ZT]; time_cond [b, ZS, T]."""
    if time.dim() > 1:
        time = time[..., 0]
    time = time.reshape(-1)
    noise = positional_embedding(time, cfg["noise_embed_dims"]).to(x.dtype)
    if cfg["cond_dim"] > 0:
        emb_in = torch.cat([noise, cond], dim=-1) if cond is not None else noise
        f = F.gelu(F.linear(emb_in, sd["embedding.0.weight"], sd["embedding.0.bias"]))
        features = F.linear(f, sd["embedding.2.weight"], sd["embedding.2.bias"])
    else:
        features = None
    h = F.gelu(F.linear(x.transpose(1, 2), sd[P + "patchify_and_embed.1.weight"],
                        sd[P + "patchify_and_embed.1.bias"]))
    if cfg["tcond_dim"] > 0 and time_cond is not None:
        tc = F.gelu(F.linear(time_cond.transpose(1, 2),
                             sd[P + "patchify_and_embed_tcond.1.weight"],
                             sd[P + "patchify_and_embed_tcond.1.bias"]))
    else:
        tc = None
    for l in range(cfg["n_layers"]):
        h = decoder_block(sd, f"{P}decoder_blocks.{l}.", h, features, tc, cfg, l, cache,
                          cache_index)
    out = F.linear(h, sd[P + "out_proj.0.weight"], sd[P + "out_proj.0.bias"])
    return out.transpose(1, 2)
