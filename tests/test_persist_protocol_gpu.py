"""The host protocol of the persistent samplers (include/after_hip.h: after_denoiser_set_stream_persist): provisioning by the
configuration calls only (the first after_sample neither allocates nor synchronises), sticky failure words reported by the next
call / after_denoiser_check / the same call in persist_check mode, serialisation of persistent launches of two handles on two
streams.  Failures are injected with diagnostics bit 4 of after_denoiser_set_*_persist (a real launch -- not the dry census --
pretends its placement census failed: it raises the sticky word and returns without touching anything).  -m gpu."""
import time

import pytest
import torch

from after_amd import _lib, pipeline

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
INJECT = 16 << 8


def _inputs(net, B, T, seed, dev):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, net.n_channels, T, generator=g).to(dev), torch.randn(B, net.cond_dim, generator=g).to(dev),
            torch.randn(B, net.tcond_dim, T, generator=g).to(dev))


@pytest.fixture(scope="module")
def stream_net(hip_device):
    model, _, _ = pipeline.build_models("cycle", "baseAE_causal", hip_device, seed=3)
    net = model.net
    net.set_gemm_path(0)
    return net


def _busy(dev, ms):
    """~ms of GPU time on the current stream in front of whatever is enqueued next"""
    torch.cuda._sleep(int(ms * 1e-3 * 2.0e9))


def test_first_streaming_sample_neither_allocates_nor_synchronises(stream_net, hip_device):
    # (another handle first: the process's own lazy work -- code objects, the runtime's signal / kernarg pools, torch's
    #  allocator pools -- is not what this test is about; the handle under test is freshly configured)
    warm, _, _ = pipeline.build_models("cycle", "baseAE_causal", hip_device, seed=33)
    warm.net.set_gemm_path(0)
    warm.net.enable_streaming_cache(max_diffusion_steps=4, max_batch_size=6, max_frames=4)
    warm.net.cfg_sample(*_inputs(warm.net, 2, 4, 1, hip_device), 4, 2.0, 1.0, -4.0)
    assert warm.net.stream_persist()
    net = stream_net
    net.enable_streaming_cache(max_diffusion_steps=4, max_batch_size=6, max_frames=4)
    net.set_stream_persist(True)
    net.reset_cache()
    x0, cond, tc = _inputs(net, 2, 4, 1, hip_device)
    keep = torch.empty_like(x0)  # (the output block comes out of torch's cache, not from hipMalloc)
    del keep
    _busy(hip_device, 0.01)  # (its own first launch loads a code object)
    torch.cuda.synchronize()
    free0, res0 = torch.cuda.mem_get_info(hip_device)[0], torch.cuda.memory_reserved(hip_device)
    _busy(hip_device, 200.0)
    t0 = time.perf_counter()
    out = net.cfg_sample(x0, cond, tc, 4, 2.0, 1.0, -4.0)
    host_ms = (time.perf_counter() - t0) * 1e3
    free1, res1 = torch.cuda.mem_get_info(hip_device)[0], torch.cuda.memory_reserved(hip_device)
    torch.cuda.synchronize()
    assert net.stream_persist()
    # device memory taken during the call = what torch's own allocator reserved for the output tensor, nothing else
    assert free0 - free1 == res1 - res0, (free0 - free1, res1 - res0)
    assert host_ms < 100.0, f"the first streaming after_sample waited for the stream ({host_ms:.1f} ms behind 200 ms of queued work)"
    assert torch.isfinite(out).all()


def test_first_offline_persistent_sample_neither_allocates_nor_synchronises(hip_device):
    warm, _, _ = pipeline.build_models("base", "baseAE", hip_device, seed=34)
    warm.net.cfg_sample(*_inputs(warm.net, 1, 256, 2, hip_device), 4, 2.0, 1.0, -4.0)
    assert warm.net.sample_persist()
    model, _, _ = pipeline.build_models("base", "baseAE", hip_device, seed=4)
    net = model.net
    x0, cond, tc = _inputs(net, 1, 256, 2, hip_device)
    net.reserve(3, 256, 4)  # capacity (re-creates the handle: a configuration step)
    net.set_persist_check(False)  # the deferred protocol (the default looks at the launch's failure words before it returns)
    keep = torch.empty_like(x0)
    del keep
    _busy(hip_device, 0.01)  # (its own first launch loads a code object)
    torch.cuda.synchronize()
    free0, res0 = torch.cuda.mem_get_info(hip_device)[0], torch.cuda.memory_reserved(hip_device)
    _busy(hip_device, 200.0)
    t0 = time.perf_counter()
    out = net.cfg_sample(x0, cond, tc, 4, 2.0, 1.0, -4.0)
    host_ms = (time.perf_counter() - t0) * 1e3
    free1, res1 = torch.cuda.mem_get_info(hip_device)[0], torch.cuda.memory_reserved(hip_device)
    torch.cuda.synchronize()
    assert net.sample_persist(), "the offline persistent sampler is the default for one base clip"
    assert free0 - free1 == res1 - res0, (free0 - free1, res1 - res0)
    assert host_ms < 100.0, host_ms
    assert torch.isfinite(out).all()


def test_streaming_failure_is_sticky_and_reported_once(stream_net, hip_device):
    net = stream_net
    net.enable_streaming_cache(max_diffusion_steps=3, max_batch_size=6, max_frames=4)
    net.set_stream_persist(False)
    net.reset_cache()
    ins = [_inputs(net, 2, 4, 10 + c, hip_device) for c in range(3)]
    want = [net.cfg_sample(*i, 3, 2.0, 1.0, -4.0).cpu() for i in ins]
    _lib.check(_lib.lib().after_denoiser_set_stream_persist(net._handle, 1 | INJECT), "set_stream_persist")
    net.reset_cache()
    # three chunks enqueued without a synchronisation in between: the first launch fails, the later ones must not be able
    # to hide that (they see the sticky word at entry and return; their failure copies show it too)
    _busy(hip_device, 50.0)
    for i in ins:
        net.cfg_sample(*i, 3, 2.0, 1.0, -4.0)
    with pytest.raises(_lib.AFTERHipError, match="persistent sampler"):
        net.check()
    net.check()  # reported once
    assert not net.stream_persist()
    # the handle serves the stream by launches from here on
    net.reset_cache()
    got = [net.cfg_sample(*i, 3, 2.0, 1.0, -4.0).cpu() for i in ins]
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    # ... and the NEXT call reports it when nobody asked (deferred protocol)
    _lib.check(_lib.lib().after_denoiser_set_stream_persist(net._handle, 1 | INJECT), "set_stream_persist")
    net.reset_cache()
    net.cfg_sample(*ins[0], 3, 2.0, 1.0, -4.0)
    torch.cuda.synchronize()
    with pytest.raises(_lib.AFTERHipError, match="persistent sampler"):
        net.cfg_sample(*ins[1], 3, 2.0, 1.0, -4.0)
    # persist_check mode: the failing call itself raises
    _lib.check(_lib.lib().after_denoiser_set_stream_persist(net._handle, 1 | INJECT), "set_stream_persist")
    net.set_persist_check(True)
    net.reset_cache()
    with pytest.raises(_lib.AFTERHipError, match="persistent sampler"):
        net.cfg_sample(*ins[0], 3, 2.0, 1.0, -4.0)
    net.set_persist_check(None)
    _lib.check(_lib.lib().after_denoiser_set_stream_persist(net._handle, 1), "set_stream_persist")
    net.reset_cache()
    got = [net.cfg_sample(*i, 3, 2.0, 1.0, -4.0).cpu() for i in ins]
    assert net.stream_persist()
    assert max((a - b).abs().max().item() for a, b in zip(got, want)) < 5e-5


@pytest.mark.parametrize("B", [1, 8])
def test_offline_failure_is_served_by_launches_within_the_call(B, hip_device):
    """RectifiedFlow.sample returns a valid tensor or raises (model.py:763-785).  DEFAULT mode: a persistent launch that refuses
    (injected: its placement census fails -- it raises the sticky word and returns without touching anything) is seen by the very
    after_sample call, which then runs by launches: the latents it returns are the launch path's, held to the oracle."""
    import oracle
    from fixtures import max_abs
    model, dcfg, _ = pipeline.build_models("base", "baseAE", hip_device, seed=4)
    net = model.net
    x0, cond, tc = _inputs(net, B, 256, 2, hip_device)
    net.set_sample_persist(False)
    want = net.cfg_sample(x0, cond, tc, 3, 2.0, 1.0, -4.0).cpu()
    assert not net.sample_persist()
    _lib.check(_lib.lib().after_denoiser_set_sample_persist(net._handle, 1 | INJECT), "set_sample_persist")
    out = torch.full_like(x0, float("nan"))  # (an untouched output would be seen)
    got = net.cfg_sample(x0, cond, tc, 3, 2.0, 1.0, -4.0, out=out).cpu()
    assert not net.sample_persist()
    assert torch.equal(got, want)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref = oracle.sample(sd, dcfg["net"], x0[:1].cpu(), cond[:1].cpu(), tc[:1].cpu(), 3, 2.0, 1.0)
    assert max_abs(got[:1], ref) < 1e-4
    # the deferred protocol (set_persist_check(False)): the failing call returns without looking; the next one reports
    _lib.check(_lib.lib().after_denoiser_set_sample_persist(net._handle, 1 | INJECT), "set_sample_persist")
    net.set_persist_check(False)
    net.cfg_sample(x0, cond, tc, 3, 2.0, 1.0, -4.0)
    torch.cuda.synchronize()
    with pytest.raises(_lib.AFTERHipError, match="persistent sampler"):
        net.cfg_sample(x0, cond, tc, 3, 2.0, 1.0, -4.0)
    net.set_persist_check(None)
    net.set_sample_persist(True)
    got = net.cfg_sample(x0, cond, tc, 3, 2.0, 1.0, -4.0).cpu()
    assert net.sample_persist()
    assert (got - want).abs().max().item() < 5e-5


def test_two_handles_on_two_streams_do_not_starve_each_other(hip_device):
    """Two persistent samplers enqueued on different streams would each hold part of the CUs and spin on barriers whose other
    workgroups cannot become resident; the library chains the second launch behind the first on the device."""
    nets = []
    for seed in (5, 6):
        model, _, _ = pipeline.build_models("base", "baseAE", hip_device, seed=seed)
        nets.append(model.net)
    ins = [_inputs(n, 1, 256, 30 + k, hip_device) for k, n in enumerate(nets)]
    refs = []
    for n, i in zip(nets, ins):
        n.reserve(3, 256, 10)
        refs.append(n.cfg_sample(*i, 10, 2.0, 1.0, -4.0).cpu())
        assert n.sample_persist()
    streams = [torch.cuda.Stream(hip_device) for _ in nets]
    torch.cuda.synchronize()
    outs = [None, None]
    t0 = time.perf_counter()
    for rep in range(5):
        for k, (n, i, st) in enumerate(zip(nets, ins, streams)):
            with torch.cuda.stream(st):
                outs[k] = n.cfg_sample(*i, 10, 2.0, 1.0, -4.0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    assert wall < 1.0, f"{wall:.2f} s for 10 ten-step clips: persistent kernels starved each other into their spin limit"
    for n, o, r in zip(nets, outs, refs):
        n.check()
        assert n.sample_persist()
        assert torch.equal(o.cpu(), r)
