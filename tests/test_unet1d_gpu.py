"""UNET1D (after_unet1d_* through the C ABI) against the reference-generated golden vectors and the
oracle on ragged shapes.  -m gpu.  Tolerance: ~20 conv layers with GroupNorm in fp32: 2e-4 x max|ref|."""
import pytest
import torch

import oracle
from after_amd import UNET1D, RectifiedFlow, configs
from after_amd import _lib
from fixtures import Fixture, max_abs

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def build(cfg_name, sd, dev):
    net = UNET1D(**configs.unet_config(cfg_name))
    res = net.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return net.to(dev)


@pytest.mark.parametrize("case", ["unet_micro", "unet_micro_flat", "unet_micro_attn"])
def test_unet1d_golden(case, hip_device):
    fx = Fixture(case)
    net = build(fx.meta["config"], fx.state_dict(), hip_device)
    d = lambda n: fx.t(n).to(hip_device)
    got = net(d("x"), time=d("time"), time_cond=d("time_cond"), cond=d("cond")).cpu()
    want = fx.t("y")
    assert got.shape == want.shape
    assert max_abs(got, want) < 2e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("cfg_name,B,T", [("unet_micro", 1, 8), ("unet_micro", 3, 40), ("unet_micro_flat", 2, 6),
                                          ("unet_micro_attn", 2, 8), ("unet_micro_attn", 1, 200)])
def test_unet1d_vs_oracle_shapes(cfg_name, B, T, hip_device):
    fx = Fixture(cfg_name)
    sd = fx.state_dict()
    cfg = configs.unet_config(cfg_name)
    net = build(cfg_name, sd, hip_device)
    g = torch.Generator().manual_seed(B * 100 + T)
    x = torch.randn(B, cfg["in_size"], T, generator=g)
    tc = torch.randn(B, cfg["time_cond_in_channels"], T, generator=g)
    cond = torch.randn(B, cfg["cond_channels"], generator=g)
    time = torch.rand(B, generator=g)
    want = oracle.unet1d_forward(sd, cfg, x, time, cond, tc)
    got = net(x.to(hip_device), time=time.to(hip_device), time_cond=tc.to(hip_device),
              cond=cond.to(hip_device)).cpu()
    assert max_abs(got, want) < 2e-4 * max(1.0, want.abs().max().item())
    if net.total_ratio > 1:  # lengths must be a multiple of the total down-sampling ratio
        with pytest.raises(ValueError):
            net(x[..., :T - 1].contiguous().to(hip_device), time=time.to(hip_device),
                time_cond=tc[..., :T - 1].contiguous().to(hip_device), cond=cond.to(hip_device))


def test_rectified_flow_samples_with_unet(hip_device):
    """RectifiedFlow.sample with a UNET1D net: the Euler / CFG loop of model.py:721-785 around
    after_unet1d_forward, against the same loop on the oracle."""
    fx = Fixture("unet_micro")
    sd = fx.state_dict()
    cfg = configs.unet_config("unet_micro")
    net = build("unet_micro", sd, hip_device)
    model = RectifiedFlow(net=net, sr=44100, device=hip_device)
    g = torch.Generator().manual_seed(5)
    B, T, N = 2, 16, 3
    x0 = torch.randn(B, 16, T, generator=g)
    tc = torch.randn(B, 12, T, generator=g)
    cond = torch.randn(B, 6, generator=g)
    x = x0
    for t in torch.linspace(0, 1, N + 1)[:-1]:
        tt = t.repeat(3 * B)
        dc, dt_ = -4.0 * torch.ones_like(cond), -4.0 * torch.ones_like(tc)
        dx = oracle.unet1d_forward(sd, cfg, x.repeat(3, 1, 1), tt, torch.cat([cond, dc, dc]),
                                   torch.cat([tc, tc, dt_]))
        d_full, d_mid, d_none = dx.chunk(3, 0)
        x = x + (d_none + 1.5 * (d_mid + 2.0 / max(1.0, 0.01) * (d_full - d_mid) - d_none)) * (1 / N)
    got = model.sample(x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), N, 2.0, 1.0).cpu()
    assert max_abs(got, x) < 5e-4 * max(1.0, x.abs().max().item())


@pytest.mark.parametrize("mode", [_lib.CFG_API, _lib.CFG_EXPORT, _lib.CFG_MIDI])
def test_unet_model_forward_cfg_modes(mode, hip_device):
    """model_forward's three CFG arrangements (model.py:730-759, export.py:364-394,
    export_midi.py:329-358) assembled, evaluated and combined on the device."""
    fx = Fixture("unet_micro")
    sd = fx.state_dict()
    cfg = configs.unet_config("unet_micro")
    net = build("unet_micro", sd, hip_device)
    model = RectifiedFlow(net=net, sr=44100, device=hip_device)
    model.cfg_mode = mode
    g = torch.Generator().manual_seed(9)
    B, T = 3, 32
    x = torch.randn(B, 16, T, generator=g)
    tc = torch.randn(B, 12, T, generator=g)
    cond = torch.randn(B, 6, generator=g)
    t = torch.rand(B, generator=g)
    gt, gs = 1.7, 0.05
    dc, dt_ = -4.0 * torch.ones_like(cond), -4.0 * torch.ones_like(tc)
    if mode == _lib.CFG_MIDI:
        c3, t3 = torch.cat([cond, cond, dc]), torch.cat([tc, dt_, dt_])
        factor = gs / max(gt, 0.1)
    else:
        c3, t3 = torch.cat([cond, dc, dc]), torch.cat([tc, tc, dt_])
        factor = gt / max(gs, 0.01 if mode == _lib.CFG_API else 0.1)
    dx = oracle.unet1d_forward(sd, cfg, x.repeat(3, 1, 1), t.repeat(3), c3, t3)
    d_full, d_mid, d_none = dx.chunk(3, 0)
    want = d_none + 0.5 * (gt + gs) * (d_mid + factor * (d_full - d_mid) - d_none)
    got = model.model_forward(x.to(hip_device), t.reshape(B, 1, 1).to(hip_device), cond.to(hip_device),
                              tc.to(hip_device), gt, gs).cpu()
    assert max_abs(got, want) < 5e-4 * max(1.0, want.abs().max().item())


def test_unet_sampler_rejects_oversize_T_before_launching(hip_device):
    """A direct C-ABI caller with T beyond the handle's max_T (or off the down-sampling grid) gets the capacity /
    argument error BEFORE the CFG assembly kernels write the max_T-sized workspaces (after_unet1d_sample /
    after_unet1d_model_forward validate T first)."""
    import ctypes
    fx = Fixture("unet_micro")
    net = build("unet_micro", fx.state_dict(), hip_device)
    h = net._ensure(3, 16)  # handle provisioned for 3 network rows x 16 frames
    B, T = 1, 64
    x = torch.zeros(B, 16, T, device=hip_device)
    tc = torch.zeros(B, 12, T, device=hip_device)
    cond = torch.zeros(B, 6, device=hip_device)
    t = torch.zeros(B, device=hip_device)
    out = torch.empty_like(x)
    L = _lib.lib()
    rc = L.after_unet1d_sample(h, _lib.ptr(x), _lib.ptr(cond), _lib.ptr(tc), _lib.ptr(out), B, T, 2,
                               ctypes.c_float(2.0), ctypes.c_float(1.0), ctypes.c_float(-4.0), 0,
                               _lib.current_stream(hip_device))
    assert rc == -3, (rc, L.after_last_error())  # AFTER_E_CAPACITY
    rc = L.after_unet1d_model_forward(h, _lib.ptr(x), _lib.ptr(t), _lib.ptr(cond), _lib.ptr(tc), _lib.ptr(out), B, T,
                                      ctypes.c_float(2.0), ctypes.c_float(1.0), ctypes.c_float(-4.0), 0,
                                      _lib.current_stream(hip_device))
    assert rc == -3, (rc, L.after_last_error())  # AFTER_E_CAPACITY
    torch.cuda.synchronize()
