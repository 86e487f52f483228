"""Oracle: AutoEncoder.encode / .decode (reference
after/autoencoder/networks/SimpleNetsStream.py, pqmf.py, core.py).

Test infrastructure -- see oracle/__init__.py.  Functional restatement on the
reference-format state dict; `cfg` holds the AutoEncoder
constructor arguments of baseAE.gin plus `padding_mode` ("centered"|"causal",
the cached_conv.get_padding gin switch, baseAE.gin:32-33)."""
import torch
import torch.nn.functional as F


def get_padding(kernel_size, stride=1, dilation=1, mode="centered"):
    """cached_conv.get_padding (third-party, absent here; restated from its
    published behaviour -- see oracle/__init__.py caveat)."""
    if kernel_size == 1:
        return (0, 0)
    p = (kernel_size - 1) * dilation + 1
    if mode == "centered":
        return ((p - 1) // 2, p // 2)
    return (p // 2 + (p - 1) // 2, 0)


def fold_weight_norm(g, v):
    """torch.nn.utils.weight_norm(dim=0): w = g * v / ||v|| with the norm over
    every dim but 0 (SimpleNetsStream.py:84-92)."""
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def _wn(sd, pre):
    return fold_weight_norm(sd[pre + "weight_g"], sd[pre + "weight_v"]), sd.get(pre + "bias")


def snake_beta(x, alpha, beta):
    """core.py:217-260 (SnakeBeta, linear-scale): x + sin^2(alpha x)/(beta+1e-9)."""
    a = alpha.reshape(1, -1, 1)
    b = beta.reshape(1, -1, 1)
    return x + (1.0 / (b + 0.000000001)) * torch.sin(x * a).pow(2)


def snake(sd, pre, x):
    """The codec's activation by the state dict's own keys: SnakeBeta (alpha, beta of [dim]) or the one-parameter
    core.py:201-209 Snake (alpha of [dim, 1]): x + (alpha + 1e-9).reciprocal() * sin(alpha x)^2."""
    alpha = sd[pre + "alpha"]
    return snake_beta(x, alpha, sd.get(pre + "beta", alpha))


def reverse_half(x):
    """pqmf.py:16-20: negate odd bands at even time indices."""
    m = torch.ones_like(x)
    m[..., 1::2, ::2] = -1
    return x * m


def pqmf_forward(sd, x, mode="centered"):
    """pqmf.py:286-290 (CachedPQMF.forward): Conv1d(1->M, k=513, stride M,
    pad get_padding(513) = (256,256) centred / (512,0) under the causal gin
    switch, pqmf.py:263-270) then reverse_half."""
    w = sd.get("pqmf.forward_conv.weight")
    if w is None or w.shape[0] == 1:  # pqmf_bands <= 1: DummyIdentity, SimpleNetsStream.py:853-859, 925-928
        return x
    M = w.shape[0]
    x = F.pad(x, get_padding(w.shape[-1], mode=mode))
    return reverse_half(F.conv1d(x, w, stride=M))


def pqmf_inverse(sd, x, mode="centered"):
    """pqmf.py:292-301 (CachedPQMF.inverse); conv pad get_padding(33) (pqmf.py:272-280)."""
    w = sd.get("pqmf.inverse_conv.weight")
    if w is None or w.shape[0] == 1:  # (SimpleNetsStream.py:946-949)
        return x
    m = w.shape[0]
    x = reverse_half(x)
    x = F.conv1d(F.pad(x, get_padding(w.shape[-1], mode=mode)), w) * m
    x = x.flip(1)
    x = x.permute(0, 2, 1)
    x = x.reshape(x.shape[0], x.shape[1], -1, m).permute(0, 2, 1, 3)
    return x.reshape(x.shape[0], x.shape[1], -1)


def conv_block(sd, pre, x, cfg, kernel_size=3, dilation=1, norm=None):
    """SimpleNetsStream.py:150-194 (ConvBlock1d): GroupNorm(min(C,8)) ->
    SnakeBeta -> weight-normed Conv1d with get_padding(k, dilation).
    norm: optional stateful stand-in for the GroupNorm, norm(prefix, x) (CachedGroupNorm.stream,
    oracle/cached.py)."""
    c = x.shape[1]
    if cfg["use_norm"] and norm is not None:
        x = norm(pre, x)
    elif cfg["use_norm"]:
        x = F.group_norm(x, min(c, 8), sd[pre + "net.0.gn.weight"], sd[pre + "net.0.gn.bias"],
                         1e-5)
    x = snake(sd, pre + "net.1.", x)
    w, b = _wn(sd, pre + "net.2.")
    x = F.pad(x, get_padding(kernel_size, dilation=dilation, mode=cfg["padding_mode"]))
    return F.conv1d(x, w, b, dilation=dilation)


def resnet_block(sd, pre, x, cfg, dilation=1, use_res=True, norm=None):
    """SimpleNetsStream.py:197-254 (ResnetBlock1d) / :257-298 (NoRes)."""
    k = cfg["kernel_size"]
    if use_res:
        y = conv_block(sd, pre + "net.branches.0.0.", x, cfg, k, dilation, norm)
        y = conv_block(sd, pre + "net.branches.0.1.", y, cfg, 1, 1, norm)
        if (pre + "net.branches.1.weight_v") in sd:
            w, b = _wn(sd, pre + "net.branches.1.")
            res = F.conv1d(x, w, b)
        else:
            res = x
        return y + res
    y = conv_block(sd, pre + "net.0.", x, cfg, k, dilation, norm)
    return conv_block(sd, pre + "net.1.", y, cfg, 1, 1, norm)


def encoder_forward(sd, x, cfg):
    """SimpleNetsStream.py:400-459 (Encoder1d) with DownsampleBlock1d :301-341."""
    pre = "encoder.net."
    x = resnet_block(sd, pre + "0.", x, cfg)
    n = len(cfg["factors"])
    for i in range(n):
        bp = f"{pre}{i + 1}.net."
        for j, d in enumerate(cfg["dilations"]):
            x = resnet_block(sd, f"{bp}{j}.", x, cfg, d)
        nb = len(cfg["dilations"])
        x = snake(sd, f"{bp}{nb}.", x)
        f = cfg["factors"][i]
        w, b = _wn(sd, f"{bp}{nb + 1}.")
        x = F.pad(x, get_padding(2 * f, f, mode=cfg["padding_mode"]))
        x = F.conv1d(x, w, b, stride=f)
    x = snake(sd, f"{pre}{n + 1}.", x)
    w, b = _wn(sd, f"{pre}{n + 2}.")
    x = F.pad(x, get_padding(3, mode=cfg["padding_mode"]))
    return F.conv1d(x, w, b)


def mod_sigmoid(x):
    """core.py:7-8."""
    return 2 * torch.sigmoid(x)**2.3 + 1e-7


def noise_generator(sd, pre, x, noise_u):
    """SimpleNetsStream.py:499-550 (NoiseGenerator, ratios [2, 2, 2]) with :462-495 (amp_to_impulse_response,
    fft_convolve).  noise_u: the uniform [0, 1) draws the reference takes from torch.rand_like(ir) (:545),
    [B, T / 8, data, 8] -- an input here, so that the branch is deterministic."""
    h = x
    for i in (0, 2, 4):
        h = F.conv1d(F.pad(h, get_padding(3, 2)), sd[f"{pre}net.{i}.weight"], sd[f"{pre}net.{i}.bias"], stride=2)
        if i != 4:
            h = F.leaky_relu(h, 0.2)
    amp = mod_sigmoid(h - 5).permute(0, 2, 1)
    data = noise_u.shape[2]
    amp = amp.reshape(amp.shape[0], amp.shape[1], data, -1)
    # amp_to_impulse_response(amp, target_size = 8)
    a = torch.fft.irfft(torch.view_as_complex(torch.stack([amp, torch.zeros_like(amp)], -1)))
    n = a.shape[-1]
    a = torch.roll(a, n // 2, -1) * torch.hann_window(n, dtype=a.dtype)
    a = F.pad(a, (0, 8 - n))
    ir = torch.roll(a, -n // 2, -1)
    noise = noise_u.to(ir.dtype) * 2 - 1
    # fft_convolve(noise, ir)
    sig = F.pad(noise, (0, noise.shape[-1]))
    ker = F.pad(ir, (ir.shape[-1], 0))
    out = torch.fft.irfft(torch.fft.rfft(sig) * torch.fft.rfft(ker))
    out = out[..., out.shape[-1] // 2:]
    out = out.permute(0, 2, 1, 3)
    return out.reshape(out.shape[0], out.shape[1], -1)


def decoder_forward(sd, z, cfg, norm=None, noise_u=None):
    """SimpleNetsStream.py:552-651 (Decoder1d) with UpsampleBlock1d :344-384."""
    pre = "decoder.net."
    w, b = _wn(sd, pre + "0.")
    x = F.conv1d(F.pad(z, get_padding(cfg["kernel_size"], mode=cfg["padding_mode"])), w, b)
    factors = cfg["factors"][::-1]
    for i, f in enumerate(factors):
        bp = f"{pre}{i + 1}.net."
        x = snake(sd, bp + "0.", x)
        w, b = _wn(sd, bp + "1.")
        if cfg.get("stream_convT"):
            # cached_conv.CachedConvTranspose1d (third-party, absent here): conv_transpose1d with
            # padding 0, overlap-add of the last `stride` samples into the next chunk, bias after.
            # On a whole stream that is the padding-0 output truncated to f*T samples.
            x = F.conv_transpose1d(x, w, None, stride=f, padding=0)[..., :f * x.shape[-1]] \
                + b[None, :, None]
        else:
            x = F.conv_transpose1d(x, w, b, stride=f, padding=f // 2)
        for j, d in enumerate(cfg["dilations"]):
            x = resnet_block(sd, f"{bp}{j + 2}.", x, cfg, d, norm=norm)
    noise = None
    if cfg.get("use_noise"):  # SimpleNetsStream.py:635-650: both branches read the last stage's output
        if noise_u is None:
            raise ValueError("use_noise: pass the uniform draws (noise_u)")
        # the module is registered twice (decoder.noise_module, decoder.synth.branches.1: same tensors in a real
        # checkpoint); load_state_dict leaves the later key's values in it
        noise = noise_generator(sd, "decoder.synth.branches.1.", x, noise_u)
    x = resnet_block(sd, "decoder.synth.branches.0.", x, cfg, 1, use_res=False, norm=norm)
    if cfg["use_loudness"]:
        x, amp = x.split(x.shape[1] // 2, 1)
        x = x * torch.sigmoid(amp)
    if noise is not None:
        x = x + noise
    return x


def ae_encode(sd, x, cfg):
    """SimpleNetsStream.py:918-941; ReluBottleneck is the identity on z at
    inference (:753-760, apply_noise=False)."""
    return encoder_forward(sd, pqmf_forward(sd, x, cfg["padding_mode"]), cfg)


def ae_decode(sd, z, cfg, noise_u=None):
    """SimpleNetsStream.py:943-954."""
    return pqmf_inverse(sd, decoder_forward(sd, z, cfg, noise_u=noise_u), cfg["padding_mode"])


def tanh_bottleneck(z, scale=3.0):
    """TanhBottleneck.forward without its noise term (SimpleNetsStream.py:719-740): scale * tanh(z)."""
    return scale * torch.tanh(z)


def vae_bottleneck(zraw):
    """VAEBottleneck.forward (SimpleNetsStream.py:763-785) up to the sampling: (mean, std, kl) of the encoder
    output [B, 2Z, T]; the reference draws z = randn * std + mean."""
    mean, scale = zraw.chunk(2, 1)
    std = torch.nn.functional.softplus(scale) + 1e-2
    var = std * std
    kl = (mean * mean + var - torch.log(var) - 1).sum(1).mean()
    return mean, std, kl


def ae_encode_raw(sd, x, cfg):
    """The encoder's output before the bottleneck (for VAE codecs: 2 x z_channels)."""
    return encoder_forward(sd, pqmf_forward(sd, x, cfg["padding_mode"]), cfg)
