"""Parity of the HIP conditioning encoders (Encoder1D as encoder_time, ECAPATDNN as
encoder) against the reference-generated golden vectors and the oracle.  -m gpu.
Tolerance: fp32 round-off of ~20 conv layers on O(1) outputs -> 1e-4 abs."""
import pytest
import torch

import oracle
from after_amd import ECAPATDNN, Encoder1D, configs
from fixtures import Fixture, max_abs

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.mark.parametrize("case", ["encoders_micro", "encoders_tiny", "encoders_base"])
def test_encoders_golden(case, hip_device):
    fx = Fixture(case)
    dcfg = configs.diffusion_config(fx.meta["config"])
    z = fx.t("z").to(hip_device)
    T = z.shape[-1]
    if dcfg["encoder_time"] is not None:
        sd = fx.state_dict("shapes_encoder_time")
        et = Encoder1D(**dcfg["encoder_time"])
        et.load_state_dict(sd, strict=True)
        et = et.to(hip_device)
        got = et(z).cpu()
        want = fx.t("time_cond")
        assert got.shape == want.shape
        assert max_abs(got, want) < 1e-4
    sd = fx.state_dict("shapes_encoder", seed_offset=1)
    ec = ECAPATDNN(**dcfg["encoder"])
    ec.load_state_dict(sd, strict=True)
    ec = ec.to(hip_device)
    got = ec(z[..., :T // 2]).cpu()
    assert max_abs(got, fx.t("cond")) < 1e-4


@pytest.mark.parametrize("B,T", [(1, 2), (3, 37), (2, 130)])
def test_encoders_vs_oracle_shapes(B, T, hip_device):
    fx = Fixture("encoders_micro")
    dcfg = configs.diffusion_config("micro")
    g = torch.Generator().manual_seed(B + 31 * T)
    z = torch.randn(B, 16, T, generator=g)
    sd = fx.state_dict("shapes_encoder_time")
    et = Encoder1D(**dcfg["encoder_time"])
    et.load_state_dict(sd)
    want = oracle.encoder1d_forward(sd, z, dcfg["encoder_time"])
    assert max_abs(et.to(hip_device)(z.to(hip_device)).cpu(), want) < 1e-4
    sd = fx.state_dict("shapes_encoder", seed_offset=1)
    ec = ECAPATDNN(**dcfg["encoder"])
    ec.load_state_dict(sd)
    want = oracle.ecapa_forward(sd, z, dcfg["encoder"])
    assert max_abs(ec.to(hip_device)(z.to(hip_device)).cpu(), want) < 1e-4


@pytest.mark.parametrize("config,B,chunks", [("micro", 2, [4, 4, 4, 4]), ("base", 1, [4, 8, 1, 3, 16])])
def test_encoder1d_streaming_matches_offline_causal(config, B, chunks, hip_device):
    """forward_stream with the cached-conv state: chunk by chunk == the offline causal pass over
    the concatenated stream (every conv of encoder_time is causal, base.gin:55, so cached_conv's
    CachedConv1d is exact)."""
    fx = Fixture("encoders_micro" if config == "micro" else "encoders_base")
    dcfg = configs.diffusion_config(config)
    assert dcfg["encoder_time"]["padding_mode"] == "causal"
    sd = fx.state_dict("shapes_encoder_time")
    et = Encoder1D(**dcfg["encoder_time"])
    et.load_state_dict(sd)
    et = et.to(hip_device)
    T = sum(chunks)
    z = torch.randn(B, et.in_size, T, generator=torch.Generator().manual_seed(T))
    want = oracle.encoder1d_forward(sd, z, dcfg["encoder_time"])
    et.enable_streaming(B, max(chunks))
    et.reset_state()
    outs, pos = [], 0
    for n in chunks:
        outs.append(et.forward_stream(z[..., pos:pos + n].contiguous().to(hip_device)).cpu())
        pos += n
    got = torch.cat(outs, -1)
    assert got.shape == want.shape
    assert max_abs(got, want) < 1e-4
    # a second stream after reset reproduces the first bit for bit
    et.reset_state()
    again = torch.cat([et.forward_stream(c.contiguous().to(hip_device)).cpu()
                       for c in z.split(chunks, -1)], -1)
    assert torch.equal(got, again)
    et.enable_streaming(B, T, enable=False)
    assert max_abs(et(z.to(hip_device)).cpu(), want) < 1e-4
