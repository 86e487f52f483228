"""Large sizes against the oracle (-m gpu): clips far longer than BASELINE's 256 frames and a batch
beyond anything else in the suite -- index arithmetic, workspace growth, XCD tile maps and the
32-bit DMA offsets at sizes where mistakes show.  Small models, so the oracle stays in seconds."""
import os

import pytest
import torch

import oracle
from after_amd import pipeline
from fixtures import max_abs, rel_l2

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("both_gemm_paths")]
torch.set_grad_enabled(False)


@pytest.fixture(scope="module", autouse=True)
def _threads():
    old = torch.get_num_threads()
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    yield
    torch.set_num_threads(old)


def cpu_sd(m):
    return {k: v.detach().cpu() for k, v in m.state_dict().items()}


def test_long_clip_sampler_and_encoders(hip_device):
    """tiny config, T = 4099 frames (190 s of audio, not a multiple of the chunk), 3 clips, 4 steps."""
    model, dcfg, _ = pipeline.build_models("tiny", "microAE", hip_device, seed=11)
    ncfg = dcfg["net"]
    g = torch.Generator().manual_seed(12)
    B, T = 3, 4099
    x0 = torch.randn(B, ncfg["n_channels"], T, generator=g)
    zs = torch.randn(B, ncfg["n_channels"], T, generator=g)
    cond = torch.randn(B, ncfg["cond_dim"], generator=g)
    tc_w = oracle.encoder1d_forward(cpu_sd(model.encoder_time), zs, dcfg["encoder_time"])
    tc = model.encoder_time(zs.to(hip_device)).cpu()
    assert max_abs(tc, tc_w) < 1e-4 * max(1.0, tc_w.abs().max().item())
    got = model.sample(x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), 4, 2.0, 1.0).cpu()
    want = oracle.sample(cpu_sd(model.net), ncfg, x0[1:2], cond[1:2], tc[1:2], 4, 2.0, 1.0)
    assert max_abs(got[1:2], want) < 1e-4, (max_abs(got[1:2], want), rel_l2(got[1:2], want))


def test_many_clips_sampler(hip_device):
    """micro config, 96 clips (288 CFG rows) x 40 frames: clips 0, 50 and 95 against the oracle."""
    model, dcfg, _ = pipeline.build_models("micro", "microAE", hip_device, seed=13)
    ncfg = dcfg["net"]
    g = torch.Generator().manual_seed(14)
    B, T = 96, 40
    x0 = torch.randn(B, ncfg["n_channels"], T, generator=g)
    cond = torch.randn(B, ncfg["cond_dim"], generator=g)
    tc = torch.randn(B, ncfg["tcond_dim"], T, generator=g)
    got = model.sample(x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), 5, 2.0, 1.0).cpu()
    sd = cpu_sd(model.net)
    for i in (0, 50, 95):
        s = slice(i, i + 1)
        want = oracle.sample(sd, ncfg, x0[s], cond[s], tc[s], 5, 2.0, 1.0)
        assert max_abs(got[s], want) < 1e-4, (i, max_abs(got[s], want))


def test_long_clip_codec(hip_device):
    """microAE on 1024 latent frames (2 097 152 samples, 47 s): decode and encode against the oracle."""
    model, _, acfg = pipeline.build_models("micro", "microAE", hip_device, seed=15)
    ae = model.emb_model
    sd = cpu_sd(ae)
    g = torch.Generator().manual_seed(16)
    T = 1024
    z = torch.randn(1, acfg["z_channels"], T, generator=g)
    x = 0.1 * torch.randn(1, 1, T * ae.ratio, generator=g)
    yw = oracle.ae_decode(sd, z, acfg)
    zw = oracle.ae_encode(sd, x, acfg)
    y = ae.decode(z.to(hip_device)).cpu()
    zg = ae.encode(x.to(hip_device))[0].cpu()
    assert y.shape == yw.shape and zg.shape == zw.shape
    assert max_abs(y, yw) < 1e-4 * yw.abs().max().item(), rel_l2(y, yw)
    assert max_abs(zg, zw) < 1e-4 * zw.abs().max().item(), rel_l2(zg, zw)


def test_base_four_times_the_clip_length(hip_device):
    """base config at T = 1024 frames (4 x BASELINE's clip: 47.6 s of audio): 2 Euler steps and the
    codec's decode of 2 097 152 samples (activations of up to 384 x 32768 per stage)."""
    model, dcfg, acfg = pipeline.build_models("base", "baseAE", hip_device, seed=17)
    ncfg = dcfg["net"]
    g = torch.Generator().manual_seed(18)
    T = 1024
    x0 = torch.randn(1, 64, T, generator=g)
    cond = torch.randn(1, 6, generator=g)
    tc = torch.randn(1, 12, T, generator=g)
    got = model.sample(x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), 2, 2.0, 1.0).cpu()
    want = oracle.sample(cpu_sd(model.net), ncfg, x0, cond, tc, 2, 2.0, 1.0)
    assert max_abs(got, want) < 1e-4, (max_abs(got, want), rel_l2(got, want))
    ae = model.emb_model
    z = torch.randn(1, 64, T, generator=g)
    yw = oracle.ae_decode(cpu_sd(ae), z, acfg)
    y = ae.decode(z.to(hip_device)).cpu()
    assert y.shape == yw.shape == (1, 1, T * 2048)
    assert max_abs(y, yw) < 1e-4 * yw.abs().max().item(), rel_l2(y, yw)
