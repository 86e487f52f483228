"""Convenience bundle of the hot path as the reference's notebooks drive it
(notebooks/audio_to_audio_demo.ipynb cells 5, 19): conditioning encoders ->
RectifiedFlow.sample -> AutoEncoder.decode.  Only composes the drop-in classes."""
import torch

from . import _streams, configs
from .autoencoder import AutoEncoder
from .diffusion import DenoiserV2, ECAPATDNN, Encoder1D, RectifiedFlow

AE_RATIO = configs.AE_RATIO
SR = configs.SR


def build_models(diffusion: str = "base", autoencoder: str = "baseAE", device="cuda:0", seed: int = 0):
    """Random-init models of a named config (BASELINE configs use random weights)."""
    torch.manual_seed(seed)
    dcfg = configs.diffusion_config(diffusion)
    net = DenoiserV2(**dcfg["net"])
    enc = ECAPATDNN(**dcfg["encoder"])
    enc_t = Encoder1D(**dcfg["encoder_time"]) if dcfg["encoder_time"] is not None else None
    acfg = configs.autoencoder_config(autoencoder)
    acfg.pop("bottleneck", None)
    ae = AutoEncoder(**acfg)
    # non-trivial BatchNorm statistics for the random-init encoders
    for mod in ([enc] + ([enc_t] if enc_t is not None else [])):
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    model = RectifiedFlow(net=net, sr=dcfg["sr"], encoder=enc, encoder_time=enc_t,
                          drop_value=dcfg["drop_value"], drop_rate=dcfg["drop_rate"], device=device)
    model.emb_model = ae.to(device)
    return model, dcfg, acfg


@torch.no_grad()
def generate_from_latents(model: RectifiedFlow, z_structure, z_timbre, x0, nb_steps=50,
                          guidance_timbre=2.0, guidance_structure=1.0, n_signal_timbre=128,
                          time_cond=None):
    """latents -> audio.  cond = encoder(z_timbre[..., :n_signal]); time_cond =
    encoder_time(z_structure) (or the given piano-roll for MIDI models)."""
    if time_cond is None and z_structure.is_cuda and _streams.CONCURRENT:
        # the two conditioning encoders are independent, ~40 launches of a few microseconds each (latency chains: 0.26 + 0.20 ms
        # at one clip): the structure encoder runs on a side stream beside the timbre encoder, the sampler waits for both
        cur = torch.cuda.current_stream(z_structure.device)
        side = _streams.side_stream(z_structure.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            time_cond = model.encoder_time(z_structure)
        z_structure.record_stream(side)
        cond = model.encoder(z_timbre[..., :n_signal_timbre])
        cur.wait_stream(side)
        time_cond.record_stream(cur)
    else:
        cond = model.encoder(z_timbre[..., :n_signal_timbre])
        if time_cond is None:
            time_cond = model.encoder_time(z_structure)
    z = model.sample(x0, cond, time_cond, nb_steps, guidance_timbre, guidance_structure)
    return model.emb_model.decode(z), z


@torch.no_grad()
def audio_to_audio(model: RectifiedFlow, audio_structure, audio_timbre, x0, **kw):
    """audio -> audio (BASELINE config 1): encode both inputs with the codec first."""
    ae = model.emb_model
    if audio_structure.shape == audio_timbre.shape and audio_structure.device == audio_timbre.device:
        # one pass of the codec over both inputs (every op of the encoder is per clip: GroupNorm statistics, convs,
        # the PQMF bank): its ~75 launches are latency chains at one clip, two clips cost 1.4x one, not 2x
        n = audio_structure.shape[0]
        z2 = ae.encode(torch.cat((audio_structure, audio_timbre)))[0]
        zs, zt = z2[:n], z2[n:]
    else:
        zs = ae.encode(audio_structure)[0]
        zt = ae.encode(audio_timbre)[0]
    return generate_from_latents(model, zs, zt, x0, **kw)
