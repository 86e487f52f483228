"""Parity of the HIP path against the CPU oracle AT BASELINE.json's sizes (-m gpu; the oracle runs
on the GPU box's host cores, each case a few seconds):

  config 2   base codec on a whole clip: decode [1,64,256] -> 524288 samples, encode the other way
             (SimpleNetsStream.py:918-954; full-T GroupNorm over up to 32768 x C/G elements)
  config 3   the per-rank shard of "base B=64 over 8 GPUs": base sampler at B=8 (24 CFG rows ->
             6144 tokens per step: the row-split GEMM tiles end to end), model.py:763-785
  config 4   midi, B=8, T=256, 50 steps (CFG_API as RectifiedFlow.sample runs it, and the
             CFG_MIDI arrangement of export_midi.py:329-358)
  config 5   base/cycle-dim Streamer, 8 independent streams, 100 cached Euler steps, 8 chunks
             (export.py:398-416) against oracle.stream_forward

Tolerances are the ones DESIGN.md states: N-step latents <= 1e-4 abs (sigma ~ 1.4; 2e-4 where
guidance amplifies, 5e-4 for the 100-step cached sampler), codec <= 1e-4 x max|oracle| (2e-4 for the norm-free
causal codec of config 5, on weights conditioned by fixtures.scale_gains)."""
import os

import pytest
import torch

import oracle
from after_amd import Streamer, _lib, pipeline
from fixtures import max_abs, rel_l2, scale_gains

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


@pytest.fixture
def base(hip_device, both_gemm_paths):  # (per test: the handle reads the leg's AFTER_GEMM_X6 when it is created)
    model, dcfg, acfg = pipeline.build_models("base", "baseAE", hip_device, seed=5)
    return model, dcfg, acfg


def test_base_codec_full_clip_vs_oracle(base, hip_device):
    model, dcfg, acfg = base
    ae = model.emb_model
    sd = cpu_sd(ae)
    g = torch.Generator().manual_seed(31)
    z = torch.randn(1, 64, 256, generator=g)
    audio = 0.1 * torch.randn(1, 1, 524288, generator=g)
    yw = oracle.ae_decode(sd, z, acfg)
    zw = oracle.ae_encode(sd, audio, acfg)
    from after_amd import diag
    n0, m0, k0 = diag.conv_x6_launches(), diag.conv1_act_launches(), diag.conv_h3_launches()
    y = ae.decode(z.to(hip_device)).cpu()
    # ... and those behind a GroupNorm on two fp16 pieces per operand (round 6): the same oracle bar holds for THAT arithmetic
    if os.environ.get("AFTER_CONV_X6", "1") != "0" and os.environ.get("AFTER_CONV_H3", "1") != "0":
        assert diag.conv_h3_launches() - k0 >= 6, diag.conv_h3_launches() - k0
    # the decoder's MFMA-bound convs (384 / 192 channels at T >= 4096) run on the bf16 pipe by default (conv_x6.hip):
    # the oracle comparison below is a comparison of THAT path, not of a silent fallback to the fp32 kernel
    if os.environ.get("AFTER_CONV_X6", "1") != "0":
        assert diag.conv_x6_launches() - n0 >= 8, diag.conv_x6_launches() - n0
    # ... and the GroupNorm -> Snake -> k = 1 convs of the last stage's ResnetBlock1ds as ONE launch each (conv1_act_kernel)
    if os.environ.get("AFTER_AE_FUSE_K1", "1") != "0":
        assert diag.conv1_act_launches() - m0 >= 3, diag.conv1_act_launches() - m0
    zg, reg = ae.encode(audio.to(hip_device))
    assert y.shape == yw.shape == (1, 1, 524288) and zg.shape == zw.shape == (1, 64, 256)
    assert max_abs(y, yw) < 1e-4 * yw.abs().max().item(), (max_abs(y, yw), yw.abs().max().item(), rel_l2(y, yw))
    assert max_abs(zg.cpu(), zw) < 1e-4 * zw.abs().max().item(), (max_abs(zg.cpu(), zw), rel_l2(zg.cpu(), zw))
    # batch of 2 whole clips (the batched GroupNorm statistics slots) equals the single clips
    z2 = torch.cat((z, z.flip(-1)))
    y2 = ae.decode(z2.to(hip_device)).cpu()
    assert max_abs(y2[:1], yw) < 1e-4 * yw.abs().max().item()


def test_base_sampler_b8_shard_vs_oracle(base, hip_device):
    """Config 3's per-rank shard: B=8, 10 Euler steps; clips 0 and 7 against the oracle, all 8
    against single-clip runs of the same handle (other tile shapes: fp32 round-off only)."""
    model, dcfg, _ = base
    sd_net = cpu_sd(model.net)
    ncfg = dcfg["net"]
    g = torch.Generator().manual_seed(32)
    B, N = 8, 10
    x0 = torch.randn(B, 64, 256, generator=g)
    cond = torch.randn(B, 6, generator=g)
    tc = torch.randn(B, 12, 256, generator=g)
    got = model.sample(x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), N, 2.0, 1.0).cpu()
    # the shard runs on the path bench.py --batch-per-gpu 8 times: the clip-per-XCD persistent kernel wherever the Linears are on
    # the bf16 pipe (with AFTER_GEMM_X6=0 -- the fixture's fp32 leg -- the launch path)
    assert model.net.sample_path() == (2 if model.net.gemm_path()[0] != 0 else 0), model.net.sample_path()
    for i in (0, 7):
        s = slice(i, i + 1)
        want = oracle.sample(sd_net, ncfg, x0[s], cond[s], tc[s], N, 2.0, 1.0)
        assert max_abs(got[s], want) < 1e-4, (i, max_abs(got[s], want))
        assert rel_l2(got[s], want) < 2e-5
    for i in range(B):
        s = slice(i, i + 1)
        one = model.sample(x0[s].to(hip_device), cond[s].to(hip_device), tc[s].to(hip_device), N, 2.0, 1.0).cpu()
        assert max_abs(one, got[s]) < 5e-5, i


def test_base_sampler_two_clips_50_steps_vs_oracle(both_gemm_paths, hip_device):
    """bench.py's leg `b2` at BASELINE's length: two clips, T = 256, 50 Euler steps in ONE launch of the one-clip kernel (192 rows per
    XCD); both clips against the oracle and against single-clip launches of the same handle."""
    if both_gemm_paths == "fp32mfma":
        pytest.skip("AFTER_GEMM_X6=0 serves every call by launches")
    model, dcfg, _ = pipeline.build_models("base", "baseAE", hip_device, seed=6)  # (its own handle: created under this leg's environment)
    sd_net = cpu_sd(model.net)
    ncfg = dcfg["net"]
    g = torch.Generator().manual_seed(33)
    B, N = 2, 50
    x0 = torch.randn(B, 64, 256, generator=g)
    cond = torch.randn(B, 6, generator=g)
    tc = torch.randn(B, 12, 256, generator=g)
    got = model.sample(x0.to(hip_device), cond.to(hip_device), tc.to(hip_device), N, 2.0, 1.0).cpu()
    assert model.net.sample_path() == 1 and model.net.sample_launches() == 1, (model.net.sample_path(), model.net.sample_launches())
    for i in range(B):
        s = slice(i, i + 1)
        want = oracle.sample(sd_net, ncfg, x0[s], cond[s], tc[s], N, 2.0, 1.0)
        assert max_abs(got[s], want) < 1e-4, (i, max_abs(got[s], want))
        assert rel_l2(got[s], want) < 2e-5
        one = model.sample(x0[s].to(hip_device), cond[s].to(hip_device), tc[s].to(hip_device), N, 2.0, 1.0).cpu()
        assert max_abs(one, got[s]) < 5e-5, i


@pytest.mark.parametrize("mode,gt,gs", [(_lib.CFG_API, 2.0, 1.0), (_lib.CFG_MIDI, 2.0, 3.0)])
def test_midi_b8_t256_50steps_vs_oracle(mode, gt, gs, hip_device):
    """Config 4: midi (tcond 128, window 16), B=8, T=256, 50 steps, synthetic piano roll
    (4 notes per clip, velocities U(0.3,1): SURVEY 8d)."""
    model, dcfg, _ = pipeline.build_models("midi", "baseAE", hip_device, seed=6)
    sd_net = cpu_sd(model.net)
    ncfg = dcfg["net"]
    g = torch.Generator().manual_seed(33)
    B, T, N = 8, 256, 50
    x0 = torch.randn(B, 64, T, generator=g)
    cond = torch.randn(B, 6, generator=g)
    roll = torch.zeros(B, 128, T)
    for b in range(B):
        for _ in range(4):
            p = int(torch.randint(20, 110, (1, ), generator=g))
            a = int(torch.randint(0, T - 32, (1, ), generator=g))
            roll[b, p, a:a + 32] = 0.3 + 0.7 * float(torch.rand((), generator=g))
    model.cfg_mode = mode
    got = model.sample(x0.to(hip_device), cond.to(hip_device), roll.to(hip_device), N, gt, gs).cpu()
    assert model.net.sample_path() == (2 if model.net.gemm_path()[0] != 0 else 0), model.net.sample_path()
    for i in (1, 6):
        s = slice(i, i + 1)
        want = oracle.sample(sd_net, ncfg, x0[s], cond[s], roll[s], N, gt, gs, cfg_mode=mode)
        assert max_abs(got[s], want) < 2e-4, (mode, i, max_abs(got[s], want))
        assert rel_l2(got[s], want) < 4e-5


def test_base_streamer_8_streams_100_steps_vs_oracle(hip_device, both_gemm_paths):
    """Config 5: base + cycle dims (cycle.gin only changes training), causal GroupNorm-free codec,
    8 independent streams, 100-step cached sampler, 8 chunks of 4 frames.  Under `fp32mfma` this IS the persistent
    streaming sampler (stream_step_kernel, the default path of this shape) against the oracle -- asserted, because a
    refused placement falls back to launches silently; under `bf16x6` (gemm path 2) the launch path."""
    model, dcfg, acfg = pipeline.build_models("cycle", "baseAE_causal", hip_device, seed=7)
    ae = model.emb_model
    # weight-norm gains x 0.5: the GroupNorm-free codec stays O(1) through its ~80 layers, so the audio bar below is
    # the codec's (2e-4 x max, as every other codec test), not the conditioning of a random-weight fixture
    ae.load_state_dict(scale_gains(ae.state_dict()))
    sd = cpu_sd(model)
    pick = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    sd_net, sd_enc, sd_et = pick("net."), pick("encoder."), pick("encoder_time.")
    sd_ae = cpu_sd(ae)
    n, chunk, steps, n_chunks, nsig = 8, 4, 100, 8, 128
    st = Streamer(model, ae, chunk_size=chunk, n_signal_timbre=nsig, max_batch=n, max_nb_steps=steps,
                  share_first_stream=False)
    st.set_nb_steps(steps)
    st.set_guidance_timbre(2.0)
    st.set_guidance_structure(1.0)
    g = torch.Generator().manual_seed(34)
    L = n_chunks * chunk * ae.ratio
    xs = 0.1 * torch.randn(n, 1, L, generator=g)
    xt = 0.1 * torch.randn(n, 1, L, generator=g)
    noise = torch.randn(n, ae.z_channels, n_chunks * chunk, generator=g)
    want_audio, want_z, want_tc = oracle.stream_forward(
        sd_net, sd_enc, sd_et, sd_ae, dcfg, acfg, xs, xt, noise, chunk, steps, 2.0, 1.0, nsig)
    lats, outs = [], []
    for c in range(n_chunks):
        a = slice(c * chunk * ae.ratio, (c + 1) * chunk * ae.ratio)
        nz = noise[..., c * chunk:(c + 1) * chunk].contiguous().to(hip_device)
        cond = torch.cat((st.structure(xs[..., a].contiguous().to(hip_device)),
                          st.timbre(xt[..., a].contiguous().to(hip_device))), 1)
        z = st.diffuse(cond, nz)
        assert model.net.stream_persist() == (both_gemm_paths == "fp32mfma"), (c, both_gemm_paths)
        lats.append(z.cpu())
        outs.append(st.decode(z).cpu())
    model.net.check()  # the last chunk's persistent launch did not time out either
    z = torch.cat(lats, -1)
    y = torch.cat(outs, -1)
    assert z.shape == want_z.shape and y.shape == want_audio.shape
    assert max_abs(z, want_z) < 5e-4 * max(1.0, want_z.abs().max().item()), (max_abs(z, want_z), rel_l2(z, want_z))
    assert max_abs(y, want_audio) < 2e-4 * want_audio.abs().max().item(), (max_abs(y, want_audio), rel_l2(y, want_audio))


def test_midi_streamer_base_dims_persistent_vs_oracle(hip_device, both_gemm_paths):
    """export_midi.py's Streamer at the midi config's real width (embed 512, window 16 -> 19 keys per chunk: two key blocks
    of the online softmax, a 16-frame K / V ring; piano-roll conditioning, CFG_MIDI): 8 cached steps, 4 chunks
    against the oracle's K / V-cache sampler (one stream: the reference's MIDI Streamer diffuses x[:1]).  Under `fp32mfma`
    on the persistent streaming sampler -- asserted."""
    from after_amd import MidiStreamer
    from oracle.sampler import CFG_MIDI
    model, dcfg, acfg = pipeline.build_models("midi", "baseAE_causal", hip_device, seed=13)
    sd = cpu_sd(model)
    sd_net = {k[4:]: v for k, v in sd.items() if k.startswith("net.")}
    ncfg = dcfg["net"]
    chunk, steps, n_chunks, nsig, n_poly, n = 4, 8, 4, 128, 4, 1
    st = MidiStreamer(model, model.emb_model, n_poly=n_poly, chunk_size=chunk, n_signal_timbre=nsig, max_batch=n,
                      max_nb_steps=steps)
    st.set_nb_steps(steps)
    st.set_guidance_timbre(1.5)
    st.set_guidance_structure(2.0)
    g = torch.Generator().manual_seed(23)
    H = ncfg["embed_dim"] // 64
    caches = [oracle.DenoiserCache(ncfg["n_layers"], 3, steps, H, ncfg["local_attention_size"], 64) for _ in range(n)]
    tvals = torch.linspace(0, 1, steps + 1)[:-1]
    for c in range(n_chunks):
        notes = torch.zeros(n, 2 * n_poly, chunk)
        for b in range(n):
            for v in range(n_poly):
                notes[b, 2 * v] = float(torch.randint(20, 110, (1, ), generator=g))
                notes[b, 2 * v + 1] = torch.randint(0, 128, (chunk, ), generator=g).float() * (torch.rand(chunk, generator=g) > 0.3)
        zsem = torch.randn(n, st.zt_channels, generator=g)
        x = torch.cat((notes, zsem.unsqueeze(-1).repeat(1, 1, chunk)), 1)
        noise = torch.randn(n, st.ae_latents, chunk, generator=g)
        tc = st.piano_roll(notes.to(hip_device)).cpu()  # (pinned against the reference recipe in test_streamer_gpu.py)
        got = st.diffuse(x.to(hip_device), noise.to(hip_device)).cpu()
        assert model.net.stream_persist() == (both_gemm_paths == "fp32mfma"), (c, both_gemm_paths)
        for b in range(n):
            want = noise[b:b + 1]
            for i, t in enumerate(tvals):
                want = want + oracle.model_forward(sd_net, ncfg, want, t.reshape(1, 1, 1), zsem[b:b + 1], tc[b:b + 1], 1.5, 2.0, -4.0,
                                                   CFG_MIDI, cache=caches[b], cache_index=i) * (1 / steps)
                caches[b].roll(chunk, i)
            assert max_abs(got[b:b + 1], want) < 5e-4, (c, b, max_abs(got[b:b + 1], want))
    model.net.check()


def test_config1_audio_to_audio_vs_oracle(hip_device):
    """BASELINE config 1 (tiny audio-to-audio: audio -> AutoEncoder.encode x 2 -> encoder / encoder_time ->
    RectifiedFlow.sample -> decode, notebooks/audio_to_audio_demo.ipynb cells 5-19) through
    pipeline.audio_to_audio -- the function INTEGRATION.md shows as the drop-in example -- on whole 524288-sample
    clips against the same chain of the CPU oracle (10 Euler steps keep the oracle leg at a few seconds)."""
    model, dcfg, acfg = pipeline.build_models("tiny", "baseAE", hip_device, seed=9)
    g = torch.Generator().manual_seed(77)
    a_s, a_t = (0.1 * torch.randn(1, 1, 524288, generator=g) for _ in range(2))
    x0 = torch.randn(1, 64, 256, generator=g)
    sd_ae, sd_net, sd_enc, sd_et = (cpu_sd(m) for m in (model.emb_model, model.net, model.encoder, model.encoder_time))
    zs, zt = oracle.ae_encode(sd_ae, a_s, acfg), oracle.ae_encode(sd_ae, a_t, acfg)
    cond = oracle.ecapa_forward(sd_enc, zt[..., :128], dcfg["encoder"])
    tc = oracle.encoder1d_forward(sd_et, zs, dcfg["encoder_time"])
    zw = oracle.sample(sd_net, dcfg["net"], x0, cond, tc, 10, 2.0, 1.0)
    yw = oracle.ae_decode(sd_ae, zw, acfg)
    y, z = pipeline.audio_to_audio(model, a_s.to(hip_device), a_t.to(hip_device), x0.to(hip_device), nb_steps=10,
                                   guidance_timbre=2.0, guidance_structure=1.0)
    assert y.shape == (1, 1, 524288) and z.shape == (1, 64, 256)
    assert max_abs(z.cpu(), zw) < 2e-4, (max_abs(z.cpu(), zw), rel_l2(z.cpu(), zw))
    assert max_abs(y.cpu(), yw) < 2e-4 * yw.abs().max().item(), (max_abs(y.cpu(), yw), yw.abs().max().item())
