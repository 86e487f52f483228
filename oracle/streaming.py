"""Oracle: the chunk-by-chunk Streamer of after_scripts/export.py:145-506, restated on whole
streams.  Test infrastructure -- see oracle/__init__.py.

PARITY UNPINNED for the cached-conv parts: `cached_conv` (acids-ircam/cached_conv, the package
behind `cc.use_cached_conv(True)`, export.py:17) is a third-party dependency absent from
/root/reference, so no reference run of the streaming codec can be generated here.  Its
published algorithm makes a *causal* CachedConv1d chunk-invariant (left context = the whole
padding), hence the streaming encoder / encoder_time outputs are the offline causal outputs of
the concatenated stream; CachedConvTranspose1d is the padding-0 transposed conv with overlap-add
(`stream_convT` in oracle/autoencoder.py).  tests/test_streaming_cpu.py restates both cached
modules chunk by chunk and checks these identities.  The K/V-cache sampler IS pinned
(tests/golden/stream_micro.npz)."""
import torch

from .autoencoder import ae_decode, ae_encode
from .denoiser import DenoiserCache
from .encoders import ecapa_forward, encoder1d_forward
from .sampler import CFG_EXPORT, model_forward


def stream_forward(sd_net, sd_enc, sd_et, sd_ae, dcfg, acfg, audio_structure, audio_timbre, noise,
                   chunk_size, nb_steps, guidance_timbre, guidance_structure, n_signal_timbre,
                   latent_range=1.0, drop_value=-4.0):
    """Streamer.forward over a whole stream (export.py:486-493), one diffusion per batch row.

    audio_*: [n, 1, n_chunks * chunk_size * ratio]; noise: [n, Z, n_chunks * chunk_size].
    Returns (audio, latents, time_cond)."""
    assert acfg["padding_mode"] == "causal" and not acfg["use_norm"]
    n = audio_structure.shape[0]
    ncfg = dcfg["net"]
    # structure (export.py:438-441): causal codec + causal encoder_time == offline on the stream
    zs = ae_encode(sd_ae, audio_structure, acfg)
    time_cond = encoder1d_forward(sd_et, zs, dcfg["encoder_time"])
    zt = ae_encode(sd_ae, audio_timbre, acfg)
    n_chunks = zs.shape[-1] // chunk_size
    prev = torch.zeros(n, zt.shape[1], n_signal_timbre)
    H = ncfg["embed_dim"] // 64
    cache = DenoiserCache(ncfg["n_layers"], 3 * n, nb_steps, H, ncfg["local_attention_size"], 64)
    t_values = torch.linspace(0, 1, nb_steps + 1)[:-1]
    dt = 1 / nb_steps
    lat = []
    for c in range(n_chunks):
        sl = slice(c * chunk_size, (c + 1) * chunk_size)
        # timbre (export.py:418-435): rolling window of the last n_signal_timbre latent frames
        prev = torch.cat((prev, zt[..., sl]), -1)[..., chunk_size:]
        zsem = ecapa_forward(sd_enc, prev, dcfg["encoder"]) / latent_range
        # diffuse (export.py:443-455): the [n, zt, chunk] repeat -> mean round trip
        cond = zsem.unsqueeze(-1).repeat(1, 1, chunk_size).mean(-1) * latent_range
        x = noise[..., sl]
        for i, t in enumerate(t_values):  # export.py:398-416
            tt = t.reshape(1, 1, 1).repeat(n, 1, 1)
            x = x + model_forward(sd_net, ncfg, x, tt, cond, time_cond[..., sl], guidance_timbre,
                                  guidance_structure, drop_value, CFG_EXPORT, cache=cache,
                                  cache_index=i) * dt
            cache.roll(chunk_size, i)
        lat.append(x)
    z = torch.cat(lat, -1)
    audio = ae_decode(sd_ae, z, dict(acfg, stream_convT=True))
    return audio, z, time_cond
