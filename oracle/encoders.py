"""Oracle: the once-per-clip conditioning encoders.

  encoder1d_forward -- reference after/diffusion/networks/encoder.py (Encoder1D,
                       used as `encoder_time`, causal padding via base.gin:55)
  ecapa_forward     -- reference after/diffusion/networks/ecapa_encoder.py
                       (ECAPATDNN, used as `encoder`)

Test infrastructure -- see oracle/__init__.py."""
import math

import torch
import torch.nn.functional as F

from .autoencoder import fold_weight_norm, get_padding


def _bn(sd, pre, x, eps=1e-5):
    """nn.BatchNorm1d in eval mode (running statistics)."""
    shape = (1, -1, 1) if x.dim() == 3 else (1, -1)
    rm, rv = sd[pre + "running_mean"].reshape(shape), sd[pre + "running_var"].reshape(shape)
    w, b = sd[pre + "weight"].reshape(shape), sd[pre + "bias"].reshape(shape)
    return (x - rm) / torch.sqrt(rv + eps) * w + b


def _wnconv(sd, pre, x, pad=(0, 0), stride=1):
    w = fold_weight_norm(sd[pre + "weight_g"], sd[pre + "weight_v"])
    return F.conv1d(F.pad(x, pad), w, sd[pre + "bias"], stride=stride)


def _v2_conv_block(sd, pre, x, k, mode):
    """encoder.py:25-71 (V2ConvBlock1D): BN->SiLU->conv->BN->SiLU->(dropout)->conv, + x."""
    b = pre + "net.branches.0."
    pad = get_padding(k, mode=mode)
    y = _wnconv(sd, b + "2.", F.silu(_bn(sd, b + "0.", x)), pad)
    y = _wnconv(sd, b + "6.", F.silu(_bn(sd, b + "3.", y)), pad)
    return y + x


def encoder1d_forward(sd, x, cfg):
    """encoder.py:273-298 (Encoder1D.forward) for the configs' settings
    (use_tanh/average_out/upscale_out/spherical/vae all False, `ac`
    regularisation leaves z unchanged :261-263).  cfg: channels, ratios,
    kernel_size, padding_mode."""
    k = cfg["kernel_size"]
    mode = cfg["padding_mode"]
    ratios = [1] + list(cfg["ratios"])
    n = len(cfg["channels"])
    for i in range(n):
        x = _v2_conv_block(sd, f"net.{i}.net.0.", x, k, mode)
        r = ratios[i]
        pad = get_padding(2 * r, r, mode=mode) if r != 1 else (0, 0)
        x = _wnconv(sd, f"net.{i}.net.1.", x, pad, stride=r)
    x = _v2_conv_block(sd, f"net.{n}.", x, k, mode)
    if cfg.get("average_out", False):
        x = x.mean(-1)
    if cfg.get("use_tanh", False):
        x = torch.tanh(x)
    return x


def _conv_reflect(sd, pre, x, dilation=1):
    """ecapa_encoder.py:12-82 (Conv1dSamePaddingReflect, stride 1)."""
    w, b = sd[pre + "conv.weight"], sd[pre + "conv.bias"]
    k = w.shape[-1]
    L = x.shape[-1]
    L_out = math.floor((L - dilation * (k - 1) - 1) / 1) + 1
    p = (L - L_out) // 2
    if p > 0:
        x = F.pad(x, (p, p), mode="reflect")
    return F.conv1d(x, w, b, dilation=dilation)


def _tdnn(sd, pre, x, dilation=1):
    """ecapa_encoder.py:85-139 (TDNNBlock): BN(ReLU(conv(x)))."""
    return _bn(sd, pre + "norm.", F.relu(_conv_reflect(sd, pre + "conv.", x, dilation)))


def _se_res2net(sd, pre, x, scale, dilation):
    """ecapa_encoder.py SERes2NetBlock / Res2NetBlock / SEBlock."""
    if (pre + "shortcut.conv.weight") in sd:
        residual = _conv_reflect(sd, pre + "shortcut.", x)
    else:
        residual = x
    x = _tdnn(sd, pre + "tdnn1.", x)
    chunks = torch.chunk(x, scale, dim=1)
    ys = [chunks[0]]
    y = None
    for i in range(scale - 1):
        xi = chunks[i + 1]
        y = _tdnn(sd, f"{pre}res2net_block.blocks.{i}.", xi if i == 0 else xi + y, dilation)
        ys.append(y)
    x = torch.cat(ys, dim=1)
    x = _tdnn(sd, pre + "tdnn2.", x)
    s = x.mean(dim=2, keepdim=True)
    s = F.relu(_conv_reflect(sd, pre + "se_block.conv1.", s))
    s = torch.sigmoid(_conv_reflect(sd, pre + "se_block.conv2.", s))
    return s * x + residual


def _stats(x, m, eps=1e-12):
    mean = (m * x).sum(dim=2)
    std = torch.sqrt((m * (x - mean.unsqueeze(2)).pow(2)).sum(dim=2).clamp(eps))
    return mean, std


def ecapa_forward(sd, x, cfg):
    """ecapa_encoder.py:567-624 (ECAPATDNN.forward, pooling, global context,
    regularisation 'ac' = identity on Z).  cfg: channels, kernel_sizes,
    dilations, res2net_scale."""
    ch = cfg["channels"]
    dil = cfg["dilations"]
    feats = []
    z = _tdnn(sd, "blocks.0.", x, dil[0])
    feats.append(z)
    for i in range(1, len(ch) - 1):
        z = _se_res2net(sd, f"blocks.{i}.", z, cfg["res2net_scale"], dil[i])
        feats.append(z)
    z = torch.cat(feats[1:], dim=1)
    z = _tdnn(sd, "mfa.", z, dil[-1])
    L = z.shape[-1]
    mean, std = _stats(z, torch.tensor(1.0 / L, dtype=z.dtype))
    attn = torch.cat([z, mean.unsqueeze(2).repeat(1, 1, L), std.unsqueeze(2).repeat(1, 1, L)],
                     dim=1)
    attn = _conv_reflect(sd, "asp.conv.", torch.tanh(_tdnn(sd, "asp.tdnn.", attn)))
    attn = F.softmax(attn, dim=2)
    mean, std = _stats(z, attn)
    st = torch.cat((mean, std), dim=1).unsqueeze(2)
    st = _bn(sd, "asp_bn.", st)
    out = _conv_reflect(sd, "fc.", st).squeeze(2)
    if cfg.get("use_tanh", False):
        out = torch.tanh(out)
    return out
