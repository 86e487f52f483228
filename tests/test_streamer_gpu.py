"""The chunk-by-chunk Streamer (after_scripts/export.py:145-506) on the GPU against the oracle's
whole-stream restatement (oracle/streaming.py).  -m gpu.

Random-init micro models (torch default init: well conditioned, unlike the crc-seeded causal
fixture); tolerances: latents 5e-4 abs on O(1) values after nb_steps cached Euler steps, audio
2e-4 of the output range through the ~80-layer norm-free codec (weight-norm gains x 0.5: fixtures.scale_gains)."""
import pytest
import torch

import oracle
from after_amd import Streamer, pipeline
from fixtures import max_abs, scale_gains

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("both_gemm_paths")]
torch.set_grad_enabled(False)


def split_sd(model):
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    pick = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    return pick("net."), pick("encoder."), pick("encoder_time.")


@pytest.mark.parametrize("n,share", [(1, True), (2, False)])
def test_streamer_matches_whole_stream_oracle(n, share, hip_device):
    model, dcfg, acfg = pipeline.build_models("micro", "microAE_causal", hip_device, seed=3)
    ae = model.emb_model
    ae.load_state_dict(scale_gains(ae.state_dict()))  # well-conditioned norm-free codec: the bar below is the code's
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
        assert max_abs(y, want_audio) < 2e-4 * want_audio.abs().max().item(), (rep, max_abs(y, want_audio))
        st.reset()


def test_streamer_forward_one_codec_pass_for_both_inputs(hip_device):
    """`forward` encodes the structure and the timbre input in ONE pass of the codec (two lanes of one streaming encoder,
    after_ae_set_stream_lanes) where the reference runs two copies of it; `structure` / `timbre` called on their own
    advance their lane alone.  Any interleaving of the two ways must give the whole-stream oracle's audio -- including
    chunks where the lanes are a call apart (then `forward` falls back to two passes) -- and a caller-supplied second
    codec (two handles, the reference's arrangement) the same."""
    from after_amd.streaming import clone_codec
    model, dcfg, acfg = pipeline.build_models("micro", "microAE_causal", hip_device, seed=3)
    ae = model.emb_model
    ae.load_state_dict(scale_gains(ae.state_dict()))
    sd_net, sd_enc, sd_et = split_sd(model)
    sd_ae = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
    n, chunk, steps, n_chunks, nsig = 2, 4, 2, 6, 16
    g = torch.Generator().manual_seed(12)
    L = n_chunks * chunk * ae.ratio
    xs = 0.3 * torch.randn(n, 1, L, generator=g)
    xt = 0.3 * torch.randn(n, 1, L, generator=g)
    noise = torch.randn(n, ae.z_channels, n_chunks * chunk, generator=g)
    want_audio, _, _ = oracle.stream_forward(sd_net, sd_enc, sd_et, sd_ae, dcfg, acfg, xs, xt, noise, chunk, steps, 2.0, 1.5, nsig)
    for second in (None, "clone"):
        st = Streamer(model, ae, chunk_size=chunk, n_signal_timbre=nsig, max_batch=n, max_nb_steps=steps, share_first_stream=False,
                      emb_model_timbre=clone_codec(ae) if second else None)
        assert st._lanes == (second is None)
        st.set_nb_steps(steps)
        st.set_guidance_timbre(2.0)
        st.set_guidance_structure(1.5)
        outs = []
        for c in range(n_chunks):
            a = slice(c * chunk * ae.ratio, (c + 1) * chunk * ae.ratio)
            x = torch.cat((xs[..., a], xt[..., a]), 1).to(hip_device)
            nz = noise[..., c * chunk:(c + 1) * chunk].contiguous().to(hip_device)
            if c in (0, 1, 4):
                outs.append(st.forward(x, nz).cpu())
            elif c == 2:  # timbre first, then structure: lanes advance one at a time
                t = st.timbre(x[:, 1:].contiguous())
                s_ = st.structure(x[:, :1].contiguous())
                outs.append(st.generate(torch.cat((s_, t), 1), nz).cpu())
            else:
                s_ = st.structure(x[:, :1].contiguous())
                t = st.timbre(x[:, 1:].contiguous())
                outs.append(st.generate(torch.cat((s_, t), 1), nz).cpu())
        y = torch.cat(outs, -1)
        assert max_abs(y, want_audio) < 2e-4 * want_audio.abs().max().item(), (second, max_abs(y, want_audio))
    # the C ABI refuses a pass over both lanes while they are a chunk apart
    st = Streamer(model, ae, chunk_size=chunk, n_signal_timbre=nsig, max_batch=n, max_nb_steps=steps, share_first_stream=False)
    x = torch.cat((xs[..., :chunk * ae.ratio], xt[..., :chunk * ae.ratio]), 1).to(hip_device)
    st.structure(x[:, :1].contiguous())
    from after_amd import _lib
    with pytest.raises(_lib.AFTERHipError):
        ae.encode(x.transpose(0, 1).reshape(2 * n, 1, -1).contiguous(), row0=0)
    st.timbre(x[:, 1:].contiguous())
    assert st.forward(x).shape == (n, 1, chunk * ae.ratio)


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
    # the remaining nn~ methods of export.py: generate_timbre (audio structure + timbre signals),
    # latent2map / map2latent (identity projection without --latent_project)
    xt = torch.randn(2, 1 + st.zt_channels, 4 * st.ae_ratio, device=hip_device)
    yt = st.generate_timbre(xt)
    assert yt.shape == (2, 1, 4 * st.ae_ratio) and torch.isfinite(yt).all()
    lat = torch.randn(2, st.zt_channels, 4, device=hip_device)
    m = st.latent2map(lat)
    assert torch.equal(m, lat.mean(-1, keepdim=True).repeat(1, 1, 4)) and torch.equal(st.map2latent(m), m)


def test_midi_streamer_matches_oracle(hip_device):
    """export_midi.py's Streamer: piano-roll conditioning + CFG_MIDI cached sampler, chunk by chunk
    against the oracle's K/V-cache sampler (pinned by stream_micro.npz) with the same roll."""
    from after_amd import MidiStreamer
    from oracle.sampler import CFG_MIDI
    model, dcfg, acfg = pipeline.build_models("micro_midi", "microAE_causal", hip_device, seed=8)
    sd_net, sd_enc, _ = split_sd(model)
    ncfg = dcfg["net"]
    chunk, steps, n_chunks, nsig, n_poly = 4, 3, 4, 16, 2
    st = MidiStreamer(model, model.emb_model, n_poly=n_poly, chunk_size=chunk, n_signal_timbre=nsig,
                      max_batch=1, max_nb_steps=steps)
    st.set_nb_steps(steps)
    st.set_guidance_timbre(1.5)
    st.set_guidance_structure(2.0)
    g = torch.Generator().manual_seed(21)
    H = ncfg["embed_dim"] // 64
    cache = oracle.DenoiserCache(ncfg["n_layers"], 3, steps, H, ncfg["local_attention_size"], 64)
    tvals = torch.linspace(0, 1, steps + 1)[:-1]
    for c in range(n_chunks):
        notes = torch.zeros(1, 2 * n_poly, chunk)
        notes[0, 0] = float(5 + c)            # voice 0: one held pitch
        notes[0, 1] = torch.tensor([0., 90., 90., 40.])
        notes[0, 2] = torch.tensor([9., 9., 12., 12.])  # voice 1 changes pitch inside the chunk
        notes[0, 3] = torch.tensor([64., 0., 64., 100.])
        zsem = torch.randn(1, st.zt_channels, generator=g)
        x = torch.cat((notes, zsem.unsqueeze(-1).repeat(1, 1, chunk)), 1)
        noise = torch.randn(1, st.ae_latents, chunk, generator=g)
        # reference recipe, literally (export_midi.py:424-432, n = 1)
        tc = torch.zeros(1, ncfg["tcond_dim"], chunk)
        for i in range(n_poly):
            for j in range(chunk):
                if notes[0, 2 * i + 1, j] > 0:
                    tc[:, notes[:, 2 * i].long(), j] = notes[:, 2 * i + 1, j] / 128
        assert torch.equal(st.piano_roll(notes.to(hip_device)).cpu(), tc)
        want = noise
        for i, t in enumerate(tvals):
            tt = t.reshape(1, 1, 1)
            want = want + oracle.model_forward(sd_net, ncfg, want, tt, zsem, tc, 1.5, 2.0, -4.0, CFG_MIDI,
                                               cache=cache, cache_index=i) * (1 / steps)
            cache.roll(chunk, i)
        got = st.diffuse(x.to(hip_device), noise.to(hip_device)).cpu()
        assert max_abs(got, want) < 5e-4, (c, max_abs(got, want))
    y = st.decode(got.to(hip_device))
    assert y.shape == (1, 1, chunk * st.ae_ratio) and torch.isfinite(y).all()
    z = st.timbre(torch.randn(1, 1, chunk * st.ae_ratio, device=hip_device))
    assert z.shape == (1, st.zt_channels, chunk)
