"""The streaming twin of a NON-causal codec's encoder (export_autoencoder.py:305-312;
after_ae_enable_encoder_streaming) against oracle/cached.py, chunk by chunk (-m gpu).

What pins the oracle: CachedGroupNorm against the reference class (tests/golden/cached_gn.npz), the
cached convs through the delayed-offline identity (tests/test_streaming_cpu.py) -- cached_conv itself
is absent from the reference tree.  Tolerances as in test_autoencoder_gpu.py: 1e-4 x max|oracle| with
GroupNorm, 2e-4 x max for the GroupNorm-free variant (weight-norm gains x 0.5 so that the norm-free stack stays
O(1): fixtures.scale_gains)."""
import pytest
import torch

import oracle
from after_amd import AutoEncoder, configs, pipeline
from fixtures import Fixture, max_abs, rel_l2, scale_gains
from oracle.autoencoder import encoder_forward, pqmf_forward

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
R = 2048


def build(sd, dev, **over):
    cfg = dict(configs.autoencoder_config("microAE"), **over)
    kw = dict(cfg)
    kw.pop("bottleneck")
    ae = AutoEncoder(**kw)
    res = ae.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys or not cfg["use_norm"]
    return ae.to(dev), cfg


def strip_norm(sd):
    return {k: v for k, v in sd.items() if ".gn." not in k}


@pytest.mark.parametrize("chunks", [[2, 1, 3, 2], [1, 1, 1, 1, 1], [4, 4]])
def test_cached_encoder_with_stream_groupnorm_vs_oracle(chunks, hip_device):
    fx = Fixture("ae_micro")
    sd = fx.state_dict()
    ae, cfg = build(sd, hip_device)
    B = 2
    g = torch.Generator().manual_seed(sum(chunks) * 7 + len(chunks))
    x = 0.1 * torch.randn(B, 1, sum(chunks) * R, generator=g)
    ref = oracle.NonCausalStreamEncoder(sd, cfg)
    D = ae.enable_encoder_streaming(B, max(chunks) * R, gn_window_samples=chunks[0] * R)
    assert D == ref.delay == ae.encoder_delay
    outs = []
    for c in x.split([n * R for n in chunks], -1):
        want = ref.encode(c)
        got = ae.encode(c.contiguous().to(hip_device))[0].cpu()
        assert got.shape == want.shape
        assert max_abs(got, want) < 1e-4 * max(1.0, want.abs().max().item()), (len(outs), max_abs(got, want), rel_l2(got, want))
        outs.append(got)
    # a new stream after reset_state reproduces the first one bit for bit
    ae.reset_state()
    again = [ae.encode(c.contiguous().to(hip_device))[0].cpu() for c in x.split([n * R for n in chunks], -1)]
    assert all(torch.equal(a, b) for a, b in zip(outs, again))
    # streams are independent: clip 1 alone
    ae.reset_state()
    solo = torch.cat([ae.encode(c[1:].contiguous().to(hip_device))[0].cpu() for c in x.split([n * R for n in chunks], -1)], -1)
    assert max_abs(solo, torch.cat(outs, -1)[1:]) < 1e-5 * max(1.0, solo.abs().max().item())


@pytest.mark.parametrize("chunks", [[1] * 24, [3, 1, 4, 8, 8], [24]])
def test_cached_encoder_is_the_delayed_offline_encoder(chunks, hip_device):
    """GroupNorm-free: for any chunking the latents are the offline latents of the concatenated
    multiband stream (per-chunk PQMF, as the reference packages it), `delay` frames late, once the
    start-up transient has left the receptive field."""
    fx = Fixture("ae_micro")
    sd = scale_gains(strip_norm(fx.state_dict()))
    ae, cfg = build(sd, hip_device, use_norm=False)
    g = torch.Generator().manual_seed(3)
    x = 0.1 * torch.randn(2, 1, sum(chunks) * R, generator=g)
    D = ae.enable_encoder_streaming(2, max(chunks) * R)
    parts = x.split([n * R for n in chunks], -1)
    z = torch.cat([ae.encode(c.contiguous().to(hip_device))[0].cpu() for c in parts], -1)
    mb = torch.cat([pqmf_forward(sd, c, "centered") for c in parts], -1)
    want = encoder_forward(sd, mb, cfg)
    W = D + 2
    a, b = z[..., D + W:], want[..., W:-D]
    assert a.shape == b.shape and a.shape[-1] >= 4
    assert max_abs(a, b) < 2e-4 * want.abs().max().item(), (D, max_abs(a, b), rel_l2(a, b))
    # and against the chunked oracle from the first frame on
    ref = oracle.NonCausalStreamEncoder(sd, cfg)
    zo = torch.cat([ref.encode(c) for c in parts], -1)
    assert max_abs(z, zo) < 2e-4 * zo.abs().max().item(), rel_l2(z, zo)
    # leaving the mode restores the offline encoder
    ae.enable_encoder_streaming(2, max(chunks) * R, enable=False)
    off = ae.encode(x[..., :4 * R].contiguous().to(hip_device))[0].cpu()
    assert max_abs(off, oracle.ae_encode(sd, x[..., :4 * R], cfg)) < 2e-4 * want.abs().max().item()


def test_cached_encoder_refusals(hip_device):
    fx = Fixture("ae_micro_causal")
    cfg = configs.autoencoder_config("microAE_causal")
    kw = dict(cfg)
    kw.pop("bottleneck")
    ae = AutoEncoder(**kw)
    ae.load_state_dict(fx.state_dict(), strict=False)
    ae = ae.to(hip_device)
    with pytest.raises(RuntimeError, match="causal codec streams through"):
        ae.enable_encoder_streaming(1, R)
    fx = Fixture("ae_micro")
    ae, _ = build(fx.state_dict(), hip_device)
    with pytest.raises(RuntimeError, match="window"):
        ae.enable_encoder_streaming(1, R, gn_window_samples=1000)
    with pytest.raises(RuntimeError, match="enable_encoder_streaming"):
        ae.enable_streaming(1, R)


def test_base_cached_encoder_vs_oracle(hip_device):
    """baseAE (non-causal, GroupNorm): 3 chunks of 4 latent frames, window = the first chunk."""
    model, dcfg, acfg = pipeline.build_models("base", "baseAE", hip_device, seed=9)
    ae = model.emb_model
    sd = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
    g = torch.Generator().manual_seed(10)
    x = 0.1 * torch.randn(1, 1, 3 * 4 * ae.ratio, generator=g)
    ref = oracle.NonCausalStreamEncoder(sd, acfg)
    D = ae.enable_encoder_streaming(1, 4 * ae.ratio, gn_window_samples=4 * ae.ratio)
    assert D == ref.delay
    for c in x.split(4 * ae.ratio, -1):
        want = ref.encode(c)
        got = ae.encode(c.contiguous().to(hip_device))[0].cpu()
        assert max_abs(got, want) < 1e-4 * max(1.0, want.abs().max().item()), (max_abs(got, want), rel_l2(got, want))


@pytest.mark.parametrize("frames", [[4, 2, 6, 4], [3, 3, 3]])
def test_decoder_with_stream_groupnorm_vs_oracle(frames, hip_device):
    """The decoder twin of the same export: offline convs, CachedGroupNorm(stream=True) statistics
    (window = the first call's length) across consecutive decode calls."""
    fx = Fixture("ae_micro")
    sd = fx.state_dict()
    ae, cfg = build(sd, hip_device)
    B = 2
    g = torch.Generator().manual_seed(sum(frames))
    ref = oracle.StreamNormDecoder(sd, cfg)
    ae.set_decoder_gn_window(B, max(frames) * R, window_latent_frames=frames[0])
    outs = []
    for n in frames:
        z = torch.randn(B, cfg["z_channels"], n, generator=g)
        want = ref.decode(z)
        got = ae.decode(z.to(hip_device)).cpu()
        assert got.shape == want.shape
        assert max_abs(got, want) < 1e-4 * max(1.0, want.abs().max().item()), (len(outs), max_abs(got, want), rel_l2(got, want))
        outs.append((z, got))
    ae.reset_state()
    for z, y in outs:
        assert torch.equal(ae.decode(z.to(hip_device)).cpu(), y)
    # window 0: the plain per-call GroupNorm again
    ae.set_decoder_gn_window(B, max(frames) * R, window_latent_frames=0)
    z = outs[1][0]
    assert max_abs(ae.decode(z.to(hip_device)).cpu(), oracle.ae_decode(sd, z, cfg)) < 1e-4 * max(1.0, outs[1][1].abs().max().item())
