"""Error behaviour of the C ABI itself (ctypes calls with raw pointers; -m gpu): negative AFTER_E_* codes
and a message through after_last_error, never an exception, never a crash -- the contract include/after_hip.h
states for bad arguments, capacities and modes."""
import ctypes

import pytest
import torch

from after_amd import _lib, pipeline

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
E_INVALID, E_CAPACITY = -1, -3


def last_error():
    m = _lib.lib().after_last_error()
    return m.decode() if m else ""


@pytest.fixture(scope="module")
def models(hip_device):
    model, dcfg, acfg = pipeline.build_models("micro", "microAE", hip_device, seed=3)
    return model, dcfg, acfg


def test_sampler_error_codes(models, hip_device):
    model, dcfg, _ = models
    net = model.net
    C = dcfg["net"]["n_channels"]
    x0 = torch.randn(2, C, 16, device=hip_device)
    cond = torch.randn(2, dcfg["net"]["cond_dim"], device=hip_device)
    tc = torch.randn(2, dcfg["net"]["tcond_dim"], 16, device=hip_device)
    out = torch.empty_like(x0)
    model.sample(x0, cond, tc, 2, 1.0, 1.0)  # creates the handle for 2 clips x 16 frames
    h, L, s = net._handle, _lib.lib(), _lib.current_stream(hip_device)
    p = _lib.ptr
    ok = L.after_sample(h, p(x0), p(cond), p(tc), p(out), 2, 16, 2, 1.0, 1.0, -4.0, 0, s)
    assert ok == 0
    assert L.after_sample(h, p(x0), p(cond), p(tc), p(out), 0, 16, 2, 1.0, 1.0, -4.0, 0, s) == E_INVALID and last_error()
    assert L.after_sample(h, None, p(cond), p(tc), p(out), 2, 16, 2, 1.0, 1.0, -4.0, 0, s) == E_INVALID
    assert L.after_sample(h, p(x0), p(cond), p(tc), p(out), 2, 16, 0, 1.0, 1.0, -4.0, 0, s) == E_INVALID
    assert L.after_sample(h, p(x0), p(cond), p(tc), p(out), 2, 16, 2, 1.0, 1.0, -4.0, 7, s) == E_INVALID  # cfg_mode
    rc = L.after_sample(h, p(x0), p(cond), p(tc), p(out), 64, 16, 2, 1.0, 1.0, -4.0, 0, s)
    assert rc == E_CAPACITY and "max" in last_error().lower()
    assert L.after_sample(None, p(x0), p(cond), p(tc), p(out), 2, 16, 2, 1.0, 1.0, -4.0, 0, s) == E_INVALID
    # the handle is still usable after refused calls, and the result is the one from before
    ref = out.clone()
    assert L.after_sample(h, p(x0), p(cond), p(tc), p(out), 2, 16, 2, 1.0, 1.0, -4.0, 0, s) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_codec_error_codes(models, hip_device):
    model, _, acfg = models
    ae = model.emb_model
    R = ae.ratio
    x = 0.1 * torch.randn(1, 1, 2 * R, device=hip_device)
    z = ae.encode(x)[0]
    h, L, s = ae._handle, _lib.lib(), _lib.current_stream(hip_device)
    p = _lib.ptr
    y = torch.empty(1, 1, 2 * R, device=hip_device)
    assert L.after_ae_decode(h, p(z), p(y), 1, 2, s) == 0
    assert L.after_ae_decode(h, p(z), p(y), 0, 2, s) == E_INVALID
    assert L.after_ae_decode(h, p(z), None, 1, 2, s) == E_INVALID
    assert L.after_ae_decode(h, p(z), p(y), 1, 10 ** 6, s) == E_CAPACITY
    assert L.after_ae_encode(h, p(x), p(z), 1, 2 * R - 16, s) == E_INVALID and "multiple" in last_error()
    assert L.after_ae_encode(h, p(x), p(z), 100, 2 * R, s) == E_CAPACITY
    # modes: this codec is non-causal with GroupNorm
    assert L.after_ae_enable_streaming(h, 1) == E_INVALID
    assert L.after_ae_enable_encoder_streaming(h, 1, 1000) == E_INVALID  # window not a multiple of the ratio
    assert L.after_ae_encoder_delay(h) == 0
    assert L.after_ae_set_decoder_gn_window(h, -1) == E_INVALID
    assert L.after_ae_reset_state(h, s) == E_INVALID  # nothing to reset yet
    assert L.after_ae_enable_encoder_streaming(h, 1, 4 * R) == 0 and L.after_ae_encoder_delay(h) > 0
    assert L.after_ae_reset_state(h, s) == 0
    assert L.after_ae_enable_encoder_streaming(h, 0, 0) == 0 and L.after_ae_encoder_delay(h) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(ae.decode(z)).all()
