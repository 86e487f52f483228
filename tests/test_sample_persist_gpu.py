"""The persistent offline sampler for one clip (denoiser.hip: sample_seg_kernel -- all Euler steps in one launch, the clip's
frames partitioned over the eight XCDs, the attention's left context handed from XCD to XCD through system-scope stores /
loads) against the launch-per-kernel path of the same handle and against the CPU oracle.  Both GPU paths form the big
Linears' fp32 products from exact three-way bf16 splits (gemm_x6 / the in-register split of sample_seg_kernel) with
different fixed K orders: they agree to fp32 round-off.  -m gpu."""
import pytest
import torch

import oracle
from after_amd import pipeline
from fixtures import max_abs, rel_l2

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def base(hip_device):
    model, dcfg, _ = pipeline.build_models("base", "baseAE", hip_device, seed=5)
    return model, dcfg


@pytest.mark.parametrize("T,steps", [(256, 6), (128, 4)])
def test_persistent_offline_sampler_matches_launch_path_and_oracle(T, steps, base, hip_device):
    model, dcfg = base
    net = model.net
    g = torch.Generator().manual_seed(41 + T)
    x0 = torch.randn(1, 64, T, generator=g)
    cond = torch.randn(1, 6, generator=g)
    tc = torch.randn(1, 12, T, generator=g)
    args = (x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), steps, 2.0, 1.0, -4.0)
    net.set_sample_persist(False)
    ref = net.cfg_sample(*args).cpu()
    assert not net.sample_persist()
    net.set_sample_persist(True)
    got = net.cfg_sample(*args).cpu()
    assert net.sample_persist(), "the persistent offline sampler refused an eligible shape"
    again = net.cfg_sample(*args).cpu()
    assert torch.equal(got, again), "not reproducible"
    assert max_abs(got, ref) < 5e-5, max_abs(got, ref)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    want = oracle.sample(sd, dcfg["net"], x0, cond, tc, steps, 2.0, 1.0)
    assert max_abs(got, want) < 1e-4 and rel_l2(got, want) < 2e-5, (max_abs(got, want), rel_l2(got, want))
    # two clips: ONE launch of the kernel for the pair (~1650 launches through round 5) -- each clip as if sampled alone (the pair's
    # Linears split K over the waves differently: fp32 round-off apart)
    x2, c2, t2 = torch.randn(2, 64, T, generator=g).to(hip_device), torch.randn(2, 6, generator=g).to(hip_device), torch.randn(2, 12, T, generator=g).to(hip_device)
    both = net.cfg_sample(x2, c2, t2, 2, 2.0, 1.0, -4.0)
    assert net.sample_path() == 1 and net.sample_launches() == 1, (net.sample_path(), net.sample_launches())
    for c in range(2):
        alone = net.cfg_sample(x2[c:c + 1].contiguous(), c2[c:c + 1].contiguous(), t2[c:c + 1].contiguous(), 2, 2.0, 1.0, -4.0)
        assert net.sample_launches() == 1
        assert max_abs(both[c:c + 1].cpu(), alone.cpu()) < 2e-5, c
    # shapes the kernel does not take fall back silently: a length that is not whole 16-frame segments
    x3 = torch.randn(1, 64, T - 6, generator=g).to(hip_device)
    net.cfg_sample(x3, c2[:1].contiguous(), t2[:1, :, :T - 6].contiguous(), 2, 2.0, 1.0, -4.0)
    assert not net.sample_persist()


def test_midi_cfg_arrangement_on_the_persistent_sampler(hip_device):
    """The midi denoiser (base width, piano-roll time conditioning, CFG_MIDI: rows (c, tc) / (c, -4) / (-4, -4), export_midi.py:329-358)
    is eligible too: the persistent kernel reads the CFG maps of the call.  Against the launch path of the same handle."""
    from after_amd import _lib
    model, dcfg, _ = pipeline.build_models("midi", "baseAE", hip_device, seed=9)
    net = model.net
    g = torch.Generator().manual_seed(77)
    x0 = torch.randn(1, net.n_channels, 256, generator=g).to(hip_device)
    cond = torch.randn(1, net.cond_dim, generator=g).to(hip_device)
    tc = torch.rand(1, net.tcond_dim, 256, generator=g).to(hip_device)
    for mode in (_lib.CFG_MIDI, _lib.CFG_API):
        net.set_sample_persist(False)
        ref = net.cfg_sample(x0, cond, tc, 5, 1.5, 2.0, -4.0, cfg_mode=mode).cpu()
        net.set_sample_persist(True)
        got = net.cfg_sample(x0, cond, tc, 5, 1.5, 2.0, -4.0, cfg_mode=mode).cpu()
        assert net.sample_persist(), mode
        assert torch.isfinite(got).all() and max_abs(got, ref) < 5e-5, (mode, max_abs(got, ref))


@pytest.fixture(scope="module")
def tiny(hip_device):
    model, dcfg, _ = pipeline.build_models("tiny", "baseAE", hip_device, seed=3)
    return model, dcfg


def _persist_vs_launch_vs_oracle(model, dcfg, T, steps, hip_device, seed):
    net = model.net
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(1, 64, T, generator=g)
    cond = torch.randn(1, 6, generator=g)
    tc = torch.randn(1, 12, T, generator=g)
    args = (x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), steps, 2.0, 1.0, -4.0)
    net.set_sample_persist(False)
    ref = net.cfg_sample(*args).cpu()
    assert not net.sample_persist()
    net.set_sample_persist(True)
    got = net.cfg_sample(*args).cpu()
    assert net.sample_persist(), f"the persistent offline sampler refused T = {T}"
    assert torch.equal(got, net.cfg_sample(*args).cpu()), "not reproducible"
    assert max_abs(got, ref) < 5e-5, max_abs(got, ref)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    want = oracle.sample(sd, dcfg["net"], x0, cond, tc, steps, 2.0, 1.0)
    assert max_abs(got, want) < 1e-4 and rel_l2(got, want) < 2e-5, (T, max_abs(got, want), rel_l2(got, want))


@pytest.mark.parametrize("T,steps", [(256, 6), (128, 4), (192, 3), (64, 3)])
def test_tiny_width_on_the_persistent_sampler(T, steps, tiny, hip_device):
    """tiny.gin:65-83 (embed 256, four heads, mlp 768): sample_seg_kernel<., 256> -- the XCD's two row halves dealt over its
    workgroups, four heads on four waves of an attention item."""
    model, dcfg = tiny
    _persist_vs_launch_vs_oracle(model, dcfg, T, steps, hip_device, 300 + T)


@pytest.mark.parametrize("T", [16, 32, 48, 64, 96, 160, 192, 224])
def test_clip_lengths_of_fewer_than_eight_segments(T, base, hip_device):
    """nseg < 8 segments of 16 or 32 frames: XCDs nseg .. 7 leave the kernel after the census; the last segment's XCD has no
    right neighbour to wait for."""
    model, dcfg = base
    _persist_vs_launch_vs_oracle(model, dcfg, T, 3, hip_device, 500 + T)


def test_geometries_the_kernel_does_not_take_run_by_launches(base, tiny, hip_device):
    """Explicit refusals: clip lengths that are not up to eight segments of 16 / 32 frames (512: longer; 40: no whole 16-frame
    segment; 144 = nine segments of 16 and not a multiple of 32).  after_sample serves them by launches, silently (the persistent
    sampler is an acceleration, not a capability), and the results are the launch path's, held to the oracle."""
    g = torch.Generator().manual_seed(5)
    cond = torch.randn(1, 6, generator=g)
    for model, dcfg in (base, tiny):
        model.net.set_sample_persist(True)
        for T in (40, 144, 512):
            x0, tc = torch.randn(1, 64, T, generator=g), torch.randn(1, 12, T, generator=g)
            got = model.net.cfg_sample(x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), 2, 2.0, 1.0, -4.0).cpu()
            assert not model.net.sample_persist(), T
            if T == 40:
                sd = {k: v.detach().cpu() for k, v in model.net.state_dict().items()}
                want = oracle.sample(sd, dcfg["net"], x0, cond, tc, 2, 2.0, 1.0)
                assert max_abs(got, want) < 1e-4, max_abs(got, want)
            assert torch.isfinite(got).all()


@pytest.mark.parametrize("B", [1, 8])
def test_two_piece_fp16_scales_follow_the_weights(B, hip_device):
    """The persistent samplers' Linears carry every operand as two fp16 pieces under a power-of-two scale chosen from GUARANTEED
    bounds of the handle's own weights (denoiser.hip: h3_scales -- LayerNorm outputs <= sqrt(E) max|w| + max|b|, the MLP hidden
    layer <= the largest row L1 norm of the up-projection x that bound + max|bias|).  A checkpoint far from the random-init scale
    -- LayerNorm gains x 40, an up-projection x 25 with biases of +-30, qkv / down-projection x 1 / 64 -- moves the scales by
    many powers of two; the result must stay on the two-piece form and within the usual bars of the launch path (three bf16
    planes) and the oracle.  Bounds beyond fp16's range at the smallest scale: the handle falls back to the three-plane form."""
    model, dcfg, _ = pipeline.build_models("base", "baseAE", hip_device, seed=21)
    net = model.net
    torch.manual_seed(3)
    sd = net.state_dict()
    for k, v in sd.items():
        if ".norm1.weight" in k or ".norm3.weight" in k:
            v.mul_(40.0)
        if ".mlp.mlp.0.weight" in k:
            v.mul_(25.0)
        if ".mlp.mlp.0.bias" in k:
            v.uniform_(-30.0, 30.0)
        if ".mlp.mlp.2.weight" in k or "qkv_linear.weight" in k:
            v.mul_(1.0 / 64.0)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(5 + B)
    T, steps = 128, 3
    x0, cond, tc = torch.randn(B, 64, T, generator=g), torch.randn(B, 6, generator=g), torch.randn(B, 12, T, generator=g)
    args = (x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), steps, 2.0, 1.0, -4.0)
    net.set_sample_persist(True)
    got = net.cfg_sample(*args).cpu()
    assert net.sample_path() == (1 if B == 1 else 2) and net.sample_arith() == 2, (net.sample_path(), net.sample_arith())
    net.set_sample_persist(False)
    ref = net.cfg_sample(*args).cpu()
    assert net.sample_path() == 0
    scale = max(1.0, ref.abs().max().item())
    assert torch.isfinite(got).all() and max_abs(got, ref) < 5e-5 * scale, (max_abs(got, ref), scale)
    sdc = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    want = oracle.sample(sdc, dcfg["net"], x0[:1], cond[:1], tc[:1], steps, 2.0, 1.0)
    assert max_abs(got[:1], want) < 1e-4 * scale and rel_l2(got[:1], want) < 2e-5, (max_abs(got[:1], want), rel_l2(got[:1], want))
    # a hidden-layer bound beyond fp16's range even at the smallest scale (2^-12): the three-plane bf16 form serves the handle
    sd = net.state_dict()
    for k, v in sd.items():
        if ".norm3.weight" in k:
            v.mul_(1e3)
        if ".mlp.mlp.0.weight" in k:
            v.mul_(1e2)
    net.load_state_dict(sd)
    net.set_sample_persist(True)
    got2 = net.cfg_sample(*args).cpu()
    assert net.sample_path() in (1, 2) and net.sample_arith() == 1, (net.sample_path(), net.sample_arith())
    net.set_sample_persist(False)
    ref2 = net.cfg_sample(*args).cpu()
    assert torch.isfinite(got2).all() == torch.isfinite(ref2).all()
    if torch.isfinite(ref2).all():
        assert max_abs(got2, ref2) < 5e-5 * max(1.0, ref2.abs().max().item())


def _pair_case(model, dcfg, B, T, steps, hip_device, seed, launches, cfg_mode=None, tcond_rand=False):
    """B clips through the one-clip kernel, two per launch: against the launch path of the same handle, against every clip sampled
    alone, and clip by clip against the CPU oracle (which samples one clip per call)."""
    net = model.net
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(B, net.n_channels, T, generator=g)
    cond = torch.randn(B, net.cond_dim, generator=g)
    tc = torch.rand(B, net.tcond_dim, T, generator=g) if tcond_rand else torch.randn(B, net.tcond_dim, T, generator=g)
    kw = {} if cfg_mode is None else {"cfg_mode": cfg_mode}
    args = (x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), steps, 2.0, 1.5, -4.0)
    net.set_sample_persist(False)
    ref = net.cfg_sample(*args, **kw).cpu()
    assert net.sample_path() == 0 and net.sample_launches() == 0
    net.set_sample_persist(True)
    got = net.cfg_sample(*args, **kw).cpu()
    assert net.sample_path() == 1 and net.sample_launches() == launches, (net.sample_path(), net.sample_launches(), launches)
    assert torch.equal(got, net.cfg_sample(*args, **kw).cpu()), "not reproducible"
    assert torch.isfinite(got).all() and max_abs(got, ref) < 5e-5, max_abs(got, ref)
    for c in range(B):
        one = tuple(a[c:c + 1].contiguous() for a in args[:3]) + args[3:]
        alone = net.cfg_sample(*one, **kw).cpu()
        assert net.sample_launches() == 1
        assert max_abs(got[c:c + 1], alone) < 2e-5, (c, max_abs(got[c:c + 1], alone))
    if cfg_mode is None:
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        for c in range(B):
            want = oracle.sample(sd, dcfg["net"], x0[c:c + 1], cond[c:c + 1], tc[c:c + 1], steps, 2.0, 1.5)
            assert max_abs(got[c:c + 1], want) < 1e-4 and rel_l2(got[c:c + 1], want) < 2e-5, (c, max_abs(got[c:c + 1], want), rel_l2(got[c:c + 1], want))


@pytest.mark.parametrize("T,steps", [(256, 5), (256, 1), (224, 3), (160, 2), (128, 4), (112, 2), (48, 3), (32, 1), (16, 2)])
def test_two_clips_share_a_launch(T, steps, base, hip_device):
    """StepArgs::nclip = 2: an XCD owns its time segment of BOTH clips' three CFG rows -- 192 rows at 32-frame segments
    (sample_seg_kernel<12, 512, 2>: four row halves, two K slices per wave), 96 at 16-frame segments (the one-clip geometry of T = 256) --
    with per-clip conditioning rows, RoPE positions, left context and Euler state."""
    model, dcfg = base
    _pair_case(model, dcfg, 2, T, steps, hip_device, 900 + T, 1)


@pytest.mark.parametrize("T,launches", [(128, 1), (64, 1), (256, 2)])
def test_two_clips_share_a_launch_tiny_width(T, launches, tiny, hip_device):
    """Width 256: pairs at 16-frame segments; at 32-frame segments (192 rows per XCD exist for the shipped width only) one clip per launch."""
    model, dcfg = tiny
    _pair_case(model, dcfg, 2, T, 3, hip_device, 1300 + T, launches)


def test_pairs_and_an_odd_clip_with_midi_cfg_rows(hip_device, monkeypatch):
    """Three and four clips as pair + single / two pairs (AFTER_SAMPLE_SEG_PAIR_MAXB moves the hand-over to the batch kernel), the
    CFG_MIDI row arrangement (the time-conditioning map differs per CFG row AND per clip), and AFTER_SEG_PAIR=0."""
    from after_amd import _lib
    monkeypatch.setenv("AFTER_SAMPLE_SEG_PAIR_MAXB", "4")
    model, dcfg, _ = pipeline.build_models("midi", "baseAE", hip_device, seed=19)
    _pair_case(model, dcfg, 3, 256, 3, hip_device, 1700, 2, cfg_mode=_lib.CFG_MIDI, tcond_rand=True)
    _pair_case(model, dcfg, 4, 128, 2, hip_device, 1701, 2, cfg_mode=_lib.CFG_MIDI, tcond_rand=True)
    model, dcfg, _ = pipeline.build_models("base", "baseAE", hip_device, seed=19)
    _pair_case(model, dcfg, 3, 256, 2, hip_device, 1702, 2)
    monkeypatch.setenv("AFTER_SEG_PAIR", "0")
    monkeypatch.setenv("AFTER_SAMPLE_SEG_PAIR_MAXB", "2")
    model, dcfg, _ = pipeline.build_models("base", "baseAE", hip_device, seed=19)
    _pair_case(model, dcfg, 2, 256, 2, hip_device, 1703, 2)
