"""The persistent offline sampler for a BATCH of clips (denoiser.hip: sample_clip_kernel -- all Euler steps of every clip in
one launch, one clip per XCD, the Linears on LDS-staged tiles of two-piece fp16 operands, gemm_h3_pipe.h) against the
launch-per-kernel path of the same handle (its Linears: three bf16 planes, gemm_x6.hip) and against the CPU oracle.  Both split forms
reproduce the fp32 products to below the fp32 accumulation error (tests/test_gemm_gpu.py): the paths agree to fp32 round-off.  -m gpu."""
import pytest
import torch

import oracle
from after_amd import _lib, pipeline
from fixtures import max_abs, rel_l2

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def base(hip_device):
    model, dcfg, _ = pipeline.build_models("base", "baseAE", hip_device, seed=5)
    return model, dcfg


def _inputs(B, T, seed, net, uniform_tc=False):
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(B, net.n_channels, T, generator=g)
    cond = torch.randn(B, net.cond_dim, generator=g)
    tc = (torch.rand if uniform_tc else torch.randn)(B, net.tcond_dim, T, generator=g)
    return x0, cond, tc


@pytest.mark.parametrize("B,T,steps", [(8, 256, 4), (5, 128, 3), (8, 64, 3), (6, 192, 2), (16, 64, 2), (11, 48, 2), (9, 256, 2), (17, 64, 2),
                                       (13, 32, 2)])  # (11, 9, 17: eight clips on the kernel + a remainder of 3 / 1 / 1 on its own path; 13: two rounds)
def test_clip_sampler_matches_launch_path_and_oracle(B, T, steps, base, hip_device):
    model, dcfg = base
    net = model.net
    x0, cond, tc = _inputs(B, T, 100 + B + T, net)
    args = (x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), steps, 2.0, 1.0, -4.0)
    net.set_sample_persist(False)
    ref = net.cfg_sample(*args).cpu()
    assert net.sample_path() == 0
    net.set_sample_persist(True)
    got = net.cfg_sample(*args).cpu()
    assert net.sample_path() == 2, "the clip-per-XCD sampler refused an eligible shape"
    again = net.cfg_sample(*args).cpu()
    assert torch.equal(got, again), "not reproducible"
    assert torch.isfinite(got).all()
    assert max_abs(got, ref) < 5e-5, max_abs(got, ref)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for c in sorted({0, B // 2, B - 1}):  # the oracle on single clips (a few seconds each at T = 256)
        want = oracle.sample(sd, dcfg["net"], x0[c:c + 1], cond[c:c + 1], tc[c:c + 1], steps, 2.0, 1.0)
        assert max_abs(got[c:c + 1], want) < 1e-4 and rel_l2(got[c:c + 1], want) < 2e-5, (c, max_abs(got[c:c + 1], want))


def test_clip_sampler_clips_are_independent(base, hip_device):
    """Clip c of a batch == the same clip sampled alone (launch path / segment kernel): nothing leaks between the XCDs."""
    model, _ = base
    net = model.net
    x0, cond, tc = _inputs(8, 256, 9, net)
    net.set_sample_persist(True)
    got = net.cfg_sample(x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), 3, 1.5, 2.5, -4.0).cpu()
    assert net.sample_path() == 2
    net.set_sample_persist(False)
    for c in (0, 3, 7):
        one = net.cfg_sample(x0[c:c + 1].to(hip_device), cond[c:c + 1].to(hip_device), tc[c:c + 1].to(hip_device), 3, 1.5, 2.5, -4.0).cpu()
        assert max_abs(got[c:c + 1], one) < 5e-5, (c, max_abs(got[c:c + 1], one))
    net.set_sample_persist(True)


def test_small_batches_and_other_lengths(base, hip_device):
    """RectifiedFlow.sample takes any batch (model.py:763-785).  One and two clips: the one-clip segment kernel, a launch per clip
    (round 6: two clips were ~1650 launches); three and four: the batch kernel with idle XCDs (round 6: launches through round 5);
    a length that is not a multiple of 16 frames runs by launches.  Every case against the launch path of the same handle and every
    clip of the small batches against the oracle."""
    model, dcfg = base
    net = model.net
    net.set_sample_persist(True)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for B, T, want_path in ((1, 256, 1), (2, 256, 1), (2, 128, 1), (3, 256, 2), (4, 256, 2), (4, 64, 2), (2, 250, 0), (8, 250, 0)):
        x0, cond, tc = _inputs(B, T, 31 + B, net)
        args = (x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), 2, 2.0, 1.0, -4.0)
        got = net.cfg_sample(*args).cpu()
        assert net.sample_path() == want_path, (B, T, net.sample_path())
        assert torch.equal(got, net.cfg_sample(*args).cpu()), "not reproducible"
        net.set_sample_persist(False)
        ref = net.cfg_sample(*args).cpu()
        assert net.sample_path() == 0
        net.set_sample_persist(True)
        assert max_abs(got, ref) < 5e-5, (B, T, max_abs(got, ref))
        if want_path and B > 1:
            for c in range(B):
                want = oracle.sample(sd, dcfg["net"], x0[c:c + 1], cond[c:c + 1], tc[c:c + 1], 2, 2.0, 1.0)
                assert max_abs(got[c:c + 1], want) < 1e-4 and rel_l2(got[c:c + 1], want) < 2e-5, (B, T, c, max_abs(got[c:c + 1], want))


def test_midi_cfg_arrangement_on_the_clip_sampler(hip_device):
    """BASELINE config 4's denoiser (base width, piano-roll time conditioning): CFG_MIDI and CFG_API row arrangements."""
    model, dcfg, _ = pipeline.build_models("midi", "baseAE", hip_device, seed=9)
    net = model.net
    x0, cond, tc = _inputs(8, 256, 77, net, uniform_tc=True)
    x0, cond, tc = x0.to(hip_device), cond.to(hip_device), tc.to(hip_device)
    for mode in (_lib.CFG_MIDI, _lib.CFG_API):
        net.set_sample_persist(False)
        ref = net.cfg_sample(x0, cond, tc, 4, 1.5, 2.0, -4.0, cfg_mode=mode).cpu()
        net.set_sample_persist(True)
        got = net.cfg_sample(x0, cond, tc, 4, 1.5, 2.0, -4.0, cfg_mode=mode).cpu()
        assert net.sample_path() == 2, mode
        assert torch.isfinite(got).all() and max_abs(got, ref) < 5e-5, (mode, max_abs(got, ref))


@pytest.mark.parametrize("B,T,steps", [(5, 512, 2), (6, 1024, 1), (8, 320, 2)])
def test_clip_sampler_long_clips_walk_several_tiles(B, T, steps, hip_device):
    """T > 256: a workgroup walks several tiles per GEMM phase (512 frames: two 192 x 192 tiles and two 96 x 128 tiles each; 1024: four),
    several attention items and tail blocks, and 320 frames leave padding rows in the last row tile.  Against the launch path of
    the same handle and, on one clip, the oracle."""
    model, dcfg, _ = pipeline.build_models("base", "baseAE", hip_device, seed=11)
    net = model.net
    net.reserve(3 * B, T, steps)
    x0, cond, tc = _inputs(B, T, 7 + T, net)
    args = (x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), steps, 2.0, 1.0, -4.0)
    net.set_sample_persist(False)
    ref = net.cfg_sample(*args).cpu()
    net.set_sample_persist(True)
    got = net.cfg_sample(*args).cpu()
    assert net.sample_path() == 2, (B, T)
    assert torch.equal(got, net.cfg_sample(*args).cpu()), "not reproducible"
    assert max_abs(got, ref) < 5e-5, max_abs(got, ref)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    c = B - 1
    want = oracle.sample(sd, dcfg["net"], x0[c:c + 1], cond[c:c + 1], tc[c:c + 1], steps, 2.0, 1.0)
    assert max_abs(got[c:c + 1], want) < 1e-4, max_abs(got[c:c + 1], want)


@pytest.mark.parametrize("cs,window,T", [(4, 16, 64), (2, 8, 64), (4, 4, 48), (2, 16, 32), (1, 8, 32)])
def test_clip_sampler_other_windows_and_chunk_sizes(cs, window, T, hip_device):
    """Base width with attention geometries no shipped config uses (transformerv2.py:62-96: any chunk size / window): W - 1 + 2 cs
    <= 16 runs the paired-chunk attention with other chunk bounds, a wider window (W = 16) the single-chunk item over two key
    blocks on the row-major residual stream.  Against the launch path of the same handle and the oracle."""
    from after_amd import configs
    from after_amd.diffusion.model import RectifiedFlow
    from after_amd.diffusion.networks.transformerv2 import DenoiserV2
    dcfg = configs.diffusion_config("base")
    ncfg = dict(dcfg["net"], attention_chunk_size=cs, local_attention_size=window)
    torch.manual_seed(3 + cs + window)
    net = DenoiserV2(**ncfg)
    model = RectifiedFlow(net=net, sr=dcfg["sr"], drop_value=dcfg["drop_value"], device=hip_device)
    net = model.net
    B, steps = 5, 2
    x0, cond, tc = _inputs(B, T, 11 * cs + window, net)
    args = (x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), steps, 2.0, 1.0, -4.0)
    net.set_sample_persist(False)
    ref = net.cfg_sample(*args).cpu()
    assert net.sample_path() == 0
    net.set_sample_persist(True)
    got = net.cfg_sample(*args).cpu()
    assert net.sample_path() == 2, (cs, window, net.sample_path())
    assert torch.isfinite(got).all() and max_abs(got, ref) < 5e-5, max_abs(got, ref)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    want = oracle.sample(sd, ncfg, x0[1:2], cond[1:2], tc[1:2], steps, 2.0, 1.0)
    assert max_abs(got[1:2], want) < 1e-4, max_abs(got[1:2], want)


@pytest.mark.parametrize("cfg,B,T", [("base", 8, 256), ("midi", 8, 256), ("base", 5, 320), ("base", 6, 48)])
def test_clip_sampler_tile_attention_against_the_item_form(cfg, B, T, hip_device):
    """The two forms of the kernel's attention -- qkv tiles that are heads and attend in place (the default: no qkv rows in
    memory, the previous row tile's last rows handed over behind a sequence word) and qkv rows + (CFG row, chunk pair) items
    (diagnostics bit 7, AFTER_CLIP_FUSE=0) -- compute the same fp32 products in the same order up to the softmax's
    block layout: they agree to round-off, and both with the launch path.  320 frames: five row tiles, CFG-row boundaries inside
    tiles, two rounds of tiles; 48: one row tile holds the three CFG rows; midi: a window of 16 (every slot of the key tile in front)."""
    model, _, _ = pipeline.build_models(cfg, "baseAE", hip_device, seed=12)
    net = model.net
    x0, cond, tc = _inputs(B, T, 900 + T, net, uniform_tc=cfg == "midi")
    args = (x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), 3, 2.0, 1.0, -4.0)
    net.set_sample_persist(False)
    ref = net.cfg_sample(*args).cpu()
    net.set_sample_persist(True)
    fused = net.cfg_sample(*args).cpu()
    assert net.sample_path() == 2
    _lib.check(_lib.lib().after_denoiser_set_sample_persist(net._handle, 1 | (128 << 8)), "set_sample_persist")
    items = net.cfg_sample(*args).cpu()
    assert net.sample_path() == 2
    _lib.check(_lib.lib().after_denoiser_set_sample_persist(net._handle, 1), "set_sample_persist")
    assert torch.isfinite(fused).all()
    assert max_abs(fused, items) < 2e-5, max_abs(fused, items)
    assert max_abs(fused, ref) < 5e-5 and max_abs(items, ref) < 5e-5, (max_abs(fused, ref), max_abs(items, ref))
