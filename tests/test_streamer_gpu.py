"""The chunk-by-chunk Streamer (after_scripts/export.py:145-506) on the GPU against the oracle's
whole-stream restatement (oracle/streaming.py).  -m gpu.

Random-init micro models (torch default init: well conditioned, unlike the crc-seeded causal
fixture); tolerances: latents 5e-4 abs on O(1) values after nb_steps cached Euler steps, audio
1e-3 of the output range through the ~80-layer norm-free codec."""
import pytest
import torch

import oracle
from after_amd import Streamer, pipeline
from fixtures import max_abs

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def split_sd(model):
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    pick = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    return pick("net."), pick("encoder."), pick("encoder_time.")


@pytest.mark.parametrize("n,share", [(1, True), (2, False)])
def test_streamer_matches_whole_stream_oracle(n, share, hip_device):
    model, dcfg, acfg = pipeline.build_models("micro", "microAE_causal", hip_device, seed=3)
    ae = model.emb_model
    sd_net, sd_enc, sd_et = split_sd(model)
    sd_ae = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
    chunk, steps, n_chunks, nsig = 4, 3, 5, 16
    st = Streamer(model, ae, chunk_size=chunk, n_signal_timbre=nsig, max_batch=n, max_nb_steps=steps,
                  share_first_stream=share)
    st.set_nb_steps(steps)
    st.set_guidance_timbre(2.0)
    st.set_guidance_structure(1.5)
    g = torch.Generator().manual_seed(11)
    L = n_chunks * chunk * ae.ratio
    xs = 0.3 * torch.randn(n, 1, L, generator=g)
    xt = 0.3 * torch.randn(n, 1, L, generator=g)
    noise = torch.randn(n, ae.z_channels, n_chunks * chunk, generator=g)
    want_audio, want_z, want_tc = oracle.stream_forward(
        sd_net, sd_enc, sd_et, sd_ae, dcfg, acfg, xs, xt, noise, chunk, steps, 2.0, 1.5, nsig)
    for rep in range(2):  # a second stream after reset() reproduces the first
        outs, lats = [], []
        for c in range(n_chunks):
            a = slice(c * chunk * ae.ratio, (c + 1) * chunk * ae.ratio)
            x = torch.cat((xs[..., a], xt[..., a]), 1).to(hip_device)
            nz = noise[..., c * chunk:(c + 1) * chunk].contiguous().to(hip_device)
            # step-wise to expose the latents too
            cond = torch.cat((st.structure(x[:, :1].contiguous()), st.timbre(x[:, 1:].contiguous())), 1)
            z = st.diffuse(cond, nz)
            lats.append(z.cpu())
            outs.append(st.decode(z).cpu())
        z = torch.cat(lats, -1)
        y = torch.cat(outs, -1)
        assert z.shape == want_z.shape and y.shape == want_audio.shape
        assert max_abs(z, want_z) < 5e-4, (rep, max_abs(z, want_z))
        assert max_abs(y, want_audio) < 1e-3 * want_audio.abs().max().item(), rep
        st.reset()


def test_streamer_forward_shapes_and_limits(hip_device):
    model, dcfg, acfg = pipeline.build_models("micro", "microAE_causal", hip_device, seed=4)
    st = Streamer(model, model.emb_model, chunk_size=4, n_signal_timbre=16, max_batch=2, max_nb_steps=2)
    x = torch.randn(2, 2, 4 * st.ae_ratio, device=hip_device)
    y = st(x)
    assert y.shape == (2, 1, 4 * st.ae_ratio) and torch.isfinite(y).all()
    assert torch.equal(y[0], y[1])  # export.py:447-452: one diffusion, repeated over the nn~ batch
    with pytest.raises(ValueError):
        st.set_nb_steps(3)
    with pytest.raises(ValueError):
        st(torch.randn(1, 2, 1000, device=hip_device))
