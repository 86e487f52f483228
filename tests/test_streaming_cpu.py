"""CPU checks of the streaming statements the GPU tests rely on (no GPU needed).

cached_conv (acids-ircam/cached_conv, pinned by the reference's requirements) is absent
from /root/reference; its published algorithm is restated here chunk by chunk in plain
torch and compared with the whole-stream forms the oracle uses."""
import torch
import torch.nn.functional as F

import oracle
from after_amd import configs
from fixtures import Fixture, max_abs
from oracle.autoencoder import get_padding

torch.set_grad_enabled(False)


def cached_conv_transpose(chunks, w, b, f):
    """CachedConvTranspose1d.forward: padding-0 transposed conv, the last `stride` output
    samples are held back and overlap-added onto the head of the next chunk, bias after."""
    cache = None
    out = []
    for x in chunks:
        y = F.conv_transpose1d(x, w, None, stride=f, padding=0)
        if cache is None:
            cache = torch.zeros_like(y[..., :f])
        y[..., :f] += cache
        cache = y[..., -f:].clone()
        out.append(y[..., :-f] + b[None, :, None])
    return torch.cat(out, -1)


def cached_conv(chunks, w, b, left, stride=1, dilation=1):
    """CachedConv1d.forward: CachedPadding1d(left) then an unpadded conv."""
    pad = None
    out = []
    for x in chunks:
        if pad is None:
            pad = torch.zeros_like(x[..., :1]).expand(-1, -1, left).clone()
        xx = torch.cat([pad, x], -1)
        pad = xx[..., xx.shape[-1] - left:].clone()
        out.append(F.conv1d(xx, w, b, stride=stride, dilation=dilation))
    return torch.cat(out, -1)


def test_cached_conv_transpose_is_truncated_padding0():
    g = torch.Generator().manual_seed(0)
    for f in (2, 4):
        w = torch.randn(6, 5, 2 * f, generator=g)
        b = torch.randn(5, generator=g)
        x = torch.randn(2, 6, 12, generator=g)
        whole = F.conv_transpose1d(x, w, None, stride=f, padding=0)[..., :f * 12] + b[None, :, None]
        got = cached_conv_transpose(list(x.split([5, 3, 4], -1)), w, b, f)
        assert max_abs(got, whole) < 1e-5


def test_cached_causal_conv_is_offline_causal():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 24, generator=g)
    for k, d, s in ((3, 1, 1), (3, 9, 1), (4, 1, 2), (8, 1, 4)):
        w = torch.randn(3, 4, k, generator=g)
        b = torch.randn(3, generator=g)
        left = get_padding(k, s, d, mode="causal")[0]
        whole = F.conv1d(F.pad(x, (left, 0)), w, b, stride=s, dilation=d)
        got = cached_conv(list(x.split([8, 4, 12], -1)), w, b, left, s, d)
        assert max_abs(got, whole) < 1e-5


def test_oracle_stream_decoder_differs_only_by_convT_alignment():
    """The padding-0 stream form is the offline causal decoder delayed by f/2 per stage; with
    the fixture's weights the two must have the same shape and be finite -- and differ."""
    fx = Fixture("ae_micro_causal")
    sd = fx.state_dict()
    cfg = configs.autoencoder_config("microAE_causal")
    z = torch.randn(1, cfg["z_channels"], 3, generator=torch.Generator().manual_seed(2))
    a = oracle.ae_decode(sd, z, cfg)
    b = oracle.ae_decode(sd, z, dict(cfg, stream_convT=True))
    assert a.shape == b.shape and torch.isfinite(b).all()
    assert max_abs(a, b) > 0
