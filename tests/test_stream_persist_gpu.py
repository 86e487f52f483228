"""The persistent streaming step (denoiser.hip: stream_step_kernel -- one launch per cached Euler step, eight XCD-local
pipelines) against the launch-per-kernel path of the SAME handle: identical inputs, K/V rings reset in between.  Both
run the same arithmetic (fp32 MFMA GEMMs with different fixed K splits, the same LayerNorm / attention code shape), so
they agree to fp32 round-off -- 5e-5 abs after 6 steps x 5 chunks on O(1) latents; the oracle bars of either path are
tests/test_baseline_size_gpu.py (config 5) and tests/test_streamer_gpu.py.  -m gpu.

Shapes: one stream (one XCD active, seven idle), fewer streams than provisioned cache rows (the unused rows are copied
through the roll), eight streams (config 5: one clip per XCD), sixteen (two clips per XCD: two 16-row blocks), eight
frames per chunk (two attention chunks per row), two frames (less than an attention chunk), six and ten frames (a ragged last
chunk; thirty rows per XCD)."""
import pytest
import torch

from after_amd import pipeline

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def net(hip_device):
    model, dcfg, _ = pipeline.build_models("cycle", "baseAE_causal", hip_device, seed=11)
    return model.net


def run_chunks(net, B, T, steps, n_chunks, seed, dev):
    g = torch.Generator().manual_seed(seed)
    outs = []
    for _ in range(n_chunks):
        x0 = torch.randn(B, net.n_channels, T, generator=g).to(dev)
        cond = torch.randn(B, net.cond_dim, generator=g).to(dev)
        tc = torch.randn(B, net.tcond_dim, T, generator=g).to(dev)
        outs.append(net.cfg_sample(x0, cond, tc, steps, 2.0, 1.5, -4.0).cpu())
    return torch.cat(outs, -1)


@pytest.mark.parametrize("B,T,max_batch", [(1, 4, 1), (3, 4, 4), (8, 4, 8), (16, 4, 16), (8, 8, 8), (2, 2, 2), (4, 6, 4), (8, 10, 8)])
def test_persistent_step_matches_launch_path(B, T, max_batch, net, hip_device, monkeypatch):
    monkeypatch.delenv("AFTER_STREAM_PERSIST", raising=False)
    steps, n_chunks = 6, 5
    net.set_gemm_path(0)  # fp32 MFMA on both paths (mode 2 would refuse the persistent kernel)
    net.enable_streaming_cache(max_diffusion_steps=steps, max_batch_size=3 * max_batch, max_frames=T)
    res = {}
    for persist in (True, False, True):  # the third pass: a reset ring reproduces the first stream
        net.set_stream_persist(persist)
        net.reset_cache()
        z = run_chunks(net, B, T, steps, n_chunks, 100 + B, hip_device)
        assert net.stream_persist() == persist, (B, T, persist)
        assert torch.isfinite(z).all()
        if persist in res:
            assert torch.equal(z, res[persist]), "persistent step: not reproducible after reset_cache"
        res[persist] = z
    d = (res[True] - res[False]).abs().max().item()
    assert d < 5e-5, (B, T, d)
    # switching paths in the middle of a stream continues it (the rings are shared state): chunks 0-2 persistent,
    # 3-4 by launches
    net.reset_cache()
    g = torch.Generator().manual_seed(100 + B)
    outs = []
    for c in range(n_chunks):
        net.set_stream_persist(c < 3)
        x0 = torch.randn(B, net.n_channels, T, generator=g).to(hip_device)
        cond = torch.randn(B, net.cond_dim, generator=g).to(hip_device)
        tc = torch.randn(B, net.tcond_dim, T, generator=g).to(hip_device)
        outs.append(net.cfg_sample(x0, cond, tc, steps, 2.0, 1.5, -4.0).cpu())
    mixed = torch.cat(outs, -1)
    assert (mixed - res[False]).abs().max().item() < 5e-5
    net.set_stream_persist(True)


def test_persistent_step_refused_shapes_fall_back(net, hip_device):
    """More token rows per XCD than the kernel's two 16-row blocks (32 streams x 4 frames -> 48): the launch path runs,
    silently, with the same results as before."""
    net.set_gemm_path(0)
    net.enable_streaming_cache(max_diffusion_steps=2, max_batch_size=96, max_frames=4)
    net.set_stream_persist(True)
    net.reset_cache()
    z = run_chunks(net, 32, 4, 2, 2, 7, hip_device)
    assert not net.stream_persist() and torch.isfinite(z).all()
    # more Euler steps than the kernel's 128 flip bits: launches, no error
    net.enable_streaming_cache(max_diffusion_steps=130, max_batch_size=3, max_frames=4)
    net.reset_cache()
    z = run_chunks(net, 1, 4, 130, 1, 8, hip_device)
    assert not net.stream_persist() and torch.isfinite(z).all()
    z = run_chunks(net, 1, 4, 100, 1, 8, hip_device)
    assert net.stream_persist()


def test_persistent_step_bad_placement_falls_back_at_first_use(net, hip_device):
    """The placement census (are the 256 workgroups placed 32 per XCD?) runs as a dry launch when the persistent path is
    configured (after_denoiser_enable_cache / _set_stream_persist), not in the first after_sample: a handle whose census fails
    serves every chunk by launches -- no invalid chunk, no error -- until it is re-enabled (which repeats the census).  The
    census failure is simulated (diagnostics bit 3 of after_denoiser_set_stream_persist)."""
    from after_amd import _lib
    net.set_gemm_path(0)
    net.enable_streaming_cache(max_diffusion_steps=3, max_batch_size=6, max_frames=4)
    net.set_stream_persist(False)
    net.reset_cache()
    want = run_chunks(net, 2, 4, 3, 3, 5, hip_device)
    _lib.check(_lib.lib().after_denoiser_set_stream_persist(net._handle, 1 | (8 << 8)), "set_stream_persist")
    net.reset_cache()
    got = run_chunks(net, 2, 4, 3, 3, 5, hip_device)
    assert not net.stream_persist()
    assert torch.equal(got, want)  # the launch path, from the first chunk on
    _lib.check(_lib.lib().after_denoiser_set_stream_persist(net._handle, 1), "set_stream_persist")
    net.set_stream_persist(True)
    net.reset_cache()
    run_chunks(net, 2, 4, 3, 1, 5, hip_device)
    assert net.stream_persist()


def test_persistent_step_midi_window_16(hip_device):
    """The midi denoiser (export_midi.py's Streamer: piano-roll conditioning, window 16 -> 19 keys per chunk: two key blocks
    of the online softmax, a 16-frame ring) under CFG_MIDI, persistent against launches."""
    from after_amd import _lib
    model, dcfg, _ = pipeline.build_models("midi", "baseAE_causal", hip_device, seed=12)
    net = model.net
    net.set_gemm_path(0)
    steps, n_chunks, B, T = 4, 5, 2, 4
    net.enable_streaming_cache(max_diffusion_steps=steps, max_batch_size=3 * B, max_frames=T)
    res = {}
    for persist in (True, False):
        net.set_stream_persist(persist)
        net.reset_cache()
        g = torch.Generator().manual_seed(77)
        outs = []
        for _ in range(n_chunks):
            x0 = torch.randn(B, net.n_channels, T, generator=g).to(hip_device)
            cond = torch.randn(B, net.cond_dim, generator=g).to(hip_device)
            tc = torch.rand(B, net.tcond_dim, T, generator=g).to(hip_device)
            outs.append(net.cfg_sample(x0, cond, tc, steps, 1.5, 2.0, -4.0, cfg_mode=_lib.CFG_MIDI).cpu())
        assert net.stream_persist() == persist
        res[persist] = torch.cat(outs, -1)
    assert torch.isfinite(res[True]).all()
    d = (res[True] - res[False]).abs().max().item()
    assert d < 5e-5, d
