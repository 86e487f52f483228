"""Oracle: the streaming NON-CAUSAL codec encoder of after_scripts/export_autoencoder.py:305-312
(`cc.use_cached_conv(True)` + `CachedGroupNorm.stream = True` around an Encoder1d built with
centred padding), chunk by chunk.  Test infrastructure -- see oracle/__init__.py.

PARITY PARTLY PINNED.  `CachedGroupNorm` is AFTER's own code (SimpleNetsStream.py:95-147) and the
restatement below is pinned against the reference class (tests/golden/cached_gn.npz).  The cached
convolutions live in the third-party `cached_conv` package (acids-ircam/cached_conv, >= 2.5.0 in
the reference's requirements), absent from /root/reference: CachedPadding1d, CachedConv1d,
AlignBranches and CachedSequential are restated from the package's published algorithm --

  CachedPadding1d(n, crop)   x <- cat(pad, x); pad <- last n samples; crop drops the last n (a delay)
  CachedConv1d               padding (l, r) becomes a cache of l + r past samples (no padding);
                             stride_delay = (s - (r + cd) % s) % s extra samples of input delay;
                             cumulative_delay = (r + stride_delay + cd) // s
  AlignBranches              every branch's INPUT is delayed by (max delay - its own delay)
  CachedSequential           cumulative_delay of its last delayed member

-- and anchored on the identity the package exists to provide, which tests/test_streaming_cpu.py
checks on this restatement: with plain (non-streaming) GroupNorm or none, the chunked encoder
output is the offline centred-padding output delayed by `delay` latent frames, for any chunking.
The reference's own call sites fix where the delays come from (SimpleNetsStream.py:162-184,
:236-249, :323-338, :441-456: ConvBlock1d's convs are built with cumulative_delay 0, the
residual branch is aligned by block1's delay, Downsample1d receives the running delay)."""
import torch
import torch.nn.functional as F

from .autoencoder import _wn, decoder_forward, get_padding, pqmf_forward, pqmf_inverse, snake_beta


class CachedPad:
    """cached_conv.CachedPadding1d"""

    def __init__(self, n, crop=False):
        self.n, self.crop, self.pad = int(n), crop, None

    def __call__(self, x):
        if not self.n:
            return x
        if self.pad is None:
            self.pad = torch.zeros(x.shape[0], x.shape[1], self.n, dtype=x.dtype)
        x = torch.cat((self.pad, x), -1)
        self.pad = x[..., -self.n:].clone()
        return x[..., :-self.n] if self.crop else x


class CachedConv:
    """cached_conv.CachedConv1d around a weight-normed Conv1d"""

    def __init__(self, w, b, k, stride=1, dilation=1, cd=0, pad_stride_arg=1):
        pl, pr = get_padding(k, pad_stride_arg, dilation, "centered")
        self.w, self.b, self.s, self.d = w, b, stride, dilation
        sd = (stride - ((pr + cd) % stride)) % stride
        self.cumulative_delay = (pr + sd + cd) // stride
        self.cache = CachedPad(pl + pr)
        self.downsampling_delay = CachedPad(sd, crop=True)

    def __call__(self, x):
        x = self.cache(self.downsampling_delay(x))
        return F.conv1d(x, self.w, self.b, stride=self.s, dilation=self.d)


class StreamGroupNorm:
    """SimpleNetsStream.py:95-147 CachedGroupNorm: stream=True normalises cat(pad, x) and keeps the
    last `padding_size` raw samples; padding_size is the length of the first call ("automatic")."""

    def __init__(self, groups, weight, bias, stream=True):
        self.g, self.w, self.b, self.stream = groups, weight, bias, stream
        self.pad, self.P = None, 0

    def __call__(self, x):
        if not self.stream:
            return F.group_norm(x, self.g, self.w, self.b, 1e-5)
        t = x.shape[-1]
        if self.pad is None:
            self.P = t
            self.pad = torch.zeros(x.shape[0], x.shape[1], self.P, dtype=x.dtype)
        x = torch.cat((self.pad, x), -1)
        self.pad = x[..., -self.P:].clone()
        return F.group_norm(x, self.g, self.w, self.b, 1e-5)[..., -t:]


class _ConvBlock:
    """ConvBlock1d (:150-194); its conv is always built with cumulative_delay 0 inside a ResnetBlock1d"""

    def __init__(self, sd, pre, cin, cfg, k, dilation, gn_stream):
        self.norm = None
        if cfg["use_norm"]:
            self.norm = StreamGroupNorm(min(cin, 8), sd[pre + "net.0.gn.weight"], sd[pre + "net.0.gn.bias"],
                                        gn_stream)
        self.alpha = sd[pre + "net.1.alpha"]
        self.beta = sd.get(pre + "net.1.beta", self.alpha)  # (core.Snake: one parameter)
        w, b = _wn(sd, pre + "net.2.")
        self.conv = CachedConv(w, b, k, 1, dilation, 0)
        self.cumulative_delay = self.conv.cumulative_delay

    def __call__(self, x):
        if self.norm is not None:
            x = self.norm(x)
        return self.conv(snake_beta(x, self.alpha, self.beta))


class _ResBlock:
    """ResnetBlock1d (:197-254) under AlignBranches(net, to_out, delays=[additional_delay, 0])"""

    def __init__(self, sd, pre, cin, cfg, dilation, cd, gn_stream):
        k = cfg["kernel_size"]
        self.b1 = _ConvBlock(sd, pre + "net.branches.0.0.", cin, cfg, k, dilation, gn_stream)
        cout = self.b1.conv.w.shape[0]
        self.b2 = _ConvBlock(sd, pre + "net.branches.0.1.", cout, cfg, 1, 1, gn_stream)
        self.to_out = _wn(sd, pre + "net.branches.1.") if (pre + "net.branches.1.weight_v") in sd else None
        add = self.b1.cumulative_delay
        self.res_delay = CachedPad(add, crop=True)  # max(delays) - 0
        self.cumulative_delay = add + cd
        self.cout = cout

    def __call__(self, x):
        y = self.b2(self.b1(x))
        r = self.res_delay(x)
        if self.to_out is not None:
            r = F.conv1d(r, self.to_out[0], self.to_out[1])
        return y + r


class NonCausalStreamEncoder:
    """AE_notcausal.encode of export_stream.ts (:305-312, :127-133): the (offline, per chunk) PQMF
    analysis, the cached Encoder1d and the bottleneck (identity at inference).  `delay`: latent
    frames by which the stream lags the offline encoder."""

    def __init__(self, sd, cfg, gn_stream=True):
        assert cfg["padding_mode"] == "centered"
        self.sd, self.cfg = sd, cfg
        pre = "encoder.net."
        M = sd["pqmf.forward_conv.weight"].shape[0] if "pqmf.forward_conv.weight" in sd else 1
        cin = max(M, 1) * cfg.get("in_channels", 1) if M > 1 else cfg.get("in_channels", 1)
        self.stem = _ResBlock(sd, pre + "0.", cin, cfg, 1, 0, gn_stream)
        cd, c = self.stem.cumulative_delay, self.stem.cout
        self.stages = []
        n = len(cfg["factors"])
        for i in range(n):
            bp = f"{pre}{i + 1}.net."
            blocks = []
            for j, d in enumerate(cfg["dilations"]):
                rb = _ResBlock(sd, f"{bp}{j}.", c, cfg, d, cd, gn_stream)
                cd = rb.cumulative_delay
                blocks.append(rb)
            nb = len(cfg["dilations"])
            f = cfg["factors"][i]
            w, b = _wn(sd, f"{bp}{nb + 1}.")
            down = CachedConv(w, b, 2 * f, f, 1, cd, pad_stride_arg=f)  # Downsample1d :32-48
            cd = down.cumulative_delay
            self.stages.append((blocks, (sd[f"{bp}{nb}.alpha"], sd.get(f"{bp}{nb}.beta", sd[f"{bp}{nb}.alpha"])), down))
            c = w.shape[0]
        self.tail_act = (sd[f"{pre}{n + 1}.alpha"], sd.get(f"{pre}{n + 1}.beta", sd[f"{pre}{n + 1}.alpha"]))
        w, b = _wn(sd, f"{pre}{n + 2}.")
        self.tail = CachedConv(w, b, 3, 1, 1, cd)
        self.delay = self.tail.cumulative_delay

    def encoder(self, mb):
        x = self.stem(mb)
        for blocks, (a, be), down in self.stages:
            for rb in blocks:
                x = rb(x)
            x = down(snake_beta(x, a, be))
        return self.tail(snake_beta(x, *self.tail_act))

    def encode(self, audio_chunk):
        return self.encoder(pqmf_forward(self.sd, audio_chunk, "centered"))


class StreamNormDecoder:
    """The decoder twin of the same export (export_autoencoder.py:305-312): an OFFLINE Decoder1d
    (zero-padded convs on every call) whose GroupNorms are CachedGroupNorm(stream=True) -- the gin
    binding precedes both constructions.  Stateful over consecutive decode calls."""

    def __init__(self, sd, cfg):
        self.sd, self.cfg, self.norms = sd, cfg, {}

    def _norm(self, pre, x):
        if pre not in self.norms:
            self.norms[pre] = StreamGroupNorm(min(x.shape[1], 8), self.sd[pre + "net.0.gn.weight"],
                                              self.sd[pre + "net.0.gn.bias"])
        return self.norms[pre](x)

    def decode(self, z):
        return pqmf_inverse(self.sd, decoder_forward(self.sd, z, self.cfg, norm=self._norm), "centered")
