"""Oracle: UNET1D (reference after/diffusion/networks/unet1d.py), functional on the reference's
state-dict keys.  Test infrastructure -- see oracle/__init__.py.  time_cond_channels > 0, cond_channels > 0; the
self-attention layers of n_attn_layers > 0 are recognised by their state-dict keys."""
import torch
import torch.nn.functional as F


def spe(t, dim, max_positions=10000, scale=32):
    """unet1d.py:7-24."""
    x = t.reshape(-1) * scale
    freqs = torch.arange(0, dim // 2, dtype=torch.float32)
    w = (1 / max_positions)**(2 * freqs / dim)
    x = x[:, None] * w[None, :]
    return torch.cat([torch.sin(x), torch.cos(x)], dim=-1)


def _gn(sd, pre, x):
    c = x.shape[1]
    return F.group_norm(x, min(16, c // 4), sd[pre + "weight"], sd[pre + "bias"], 1e-5)


def _mlp(sd, pre, v):
    h = F.silu(F.linear(v, sd[pre + "0.weight"], sd[pre + "0.bias"]))
    return F.linear(h, sd[pre + "2.weight"], sd[pre + "2.bias"])


def conv_block(sd, pre, x, temb, cond, skip=None, tcond=None, res=True):
    """ConvBlock1D.forward, unet1d.py:83-118."""
    r = x
    if skip is not None:
        x = torch.cat([x, skip], 1)
    if tcond is not None:
        x = torch.cat([x, tcond], 1)
    k = sd[pre + "conv1.weight"].shape[-1]
    x = F.silu(_gn(sd, pre + "gn1.", x))
    x = F.conv1d(x, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"], padding=k // 2)
    tm, ta = _mlp(sd, pre + "time_mlp.", temb).chunk(2, 1)
    x = x * tm[:, :, None] + ta[:, :, None]
    cm, ca = _mlp(sd, pre + "cond_mlp.", cond).chunk(2, 1)
    x = x * cm[:, :, None] + ca[:, :, None]
    x = F.silu(_gn(sd, pre + "gn2.", x))
    x = F.conv1d(x, sd[pre + "conv2.weight"], sd[pre + "conv2.bias"], padding=k // 2)
    if res:
        if pre + "to_out.weight" in sd:
            r = F.conv1d(r, sd[pre + "to_out.weight"], sd[pre + "to_out.bias"])
        return x + r
    return x


def self_attention(sd, pre, x, n_head):
    """SelfAttention1d.forward (blocks.py:201-243; dropout is the identity at inference): GroupNorm(1, C) -> 1 x 1 qkv ->
    full softmax attention per head, q and k each scaled by d ** -0.25 -> 1 x 1 out_proj, + input.  Identity when the block
    has no such layer (`pre`self_attn.* absent from the state dict)."""
    if pre + "self_attn.norm.weight" not in sd:
        return x
    p = pre + "self_attn."
    n, c, s = x.shape
    qkv = F.conv1d(F.group_norm(x, 1, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5), sd[p + "qkv_proj.weight"],
                   sd[p + "qkv_proj.bias"])
    qkv = qkv.view(n, n_head * 3, c // n_head, s).transpose(2, 3)
    q, k, v = qkv.chunk(3, dim=1)
    scale = k.shape[3]**-0.25
    att = ((q * scale) @ (k.transpose(2, 3) * scale)).softmax(3)
    y = (att @ v).transpose(2, 3).contiguous().view(n, c, s)
    return x + F.conv1d(y, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def unet1d_forward(sd, cfg, x, time, cond, time_cond):
    """UNET1D.forward (time_cond_channels > 0 branch), unet1d.py:374-414."""
    n = len(cfg["channels"])
    R = ([1] + list(cfg["ratios"]))[:n]
    k = cfg["kernel_size"]
    temb = spe(time, cfg["time_channels"])
    skips, tcs = [], []
    tc = time_cond
    for i in range(n):
        stride = 1 if i == 0 else R[i - 1]
        tc = F.silu(F.conv1d(tc, sd[f"cond_emb_time.{i}.0.weight"], sd[f"cond_emb_time.{i}.0.bias"],
                             stride=stride, padding=k // 2))
        skip = conv_block(sd, f"down_layers.{i}.conv.", x, temb, cond, tcond=tc)
        skip = self_attention(sd, f"down_layers.{i}.", skip, 4)  # EncoderBlock1D: SelfAttention1d(in_c, 4), unet1d.py:145-146
        x = F.conv1d(skip, sd[f"down_layers.{i}.pool.weight"], sd[f"down_layers.{i}.pool.bias"],
                     stride=R[i], padding=k // 2)
        skips.append(skip)
        tcs.append(tc)
    tc = F.silu(F.conv1d(tc, sd[f"cond_emb_time.{n}.0.weight"], sd[f"cond_emb_time.{n}.0.bias"],
                         stride=R[n - 1], padding=k // 2))
    x = conv_block(sd, "middle_block.conv.", x, temb, cond, tcond=tc)
    x = self_attention(sd, "middle_block.", x, max(1, x.shape[1] // 32))  # MiddleBlock1D: (in_c, in_c // 32), :191-192
    for j in range(n):
        ratio = R[n - 1 - j]
        pre = f"up_layers.{j}."
        if ratio != 1:
            x = F.interpolate(x, scale_factor=ratio, mode="nearest")
            x = F.conv1d(x, sd[pre + "up.1.weight"], sd[pre + "up.1.bias"], padding=1)
        elif pre + "up.weight" in sd:
            x = F.conv1d(x, sd[pre + "up.weight"], sd[pre + "up.bias"], padding=1)
        last = j == n - 1
        x = conv_block(sd, pre + "conv.", x, temb, cond, skip=skips.pop(-1), tcond=tcs.pop(-1),
                       res=(not last) or bool(cfg.get("use_res_last", False)))
        x = self_attention(sd, pre, x, 4)  # DecoderBlock1D: SelfAttention1d(out_c, 4), :242-243
    return x
