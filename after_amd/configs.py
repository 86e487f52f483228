"""Architecture hyper-parameters of the shipped AFTER configs, restated as
plain Python dicts (there is no gin on the GPU box).

Sources (reference file:line):
  tiny  -- after/diffusion/configs/tiny.gin:27-89
  base  -- after/diffusion/configs/base.gin:27-89
  midi  -- after/diffusion/configs/midi.gin:26-73
  cycle -- after/diffusion/configs/cycle.gin:6-14 (training losses only ->
           inference graph identical to base)
  baseAE -- after/autoencoder/configs/baseAE.gin:13-52

IN_SIZE / N_SIGNAL are bound at run time in the reference from the AE probe
(after_scripts/train.py:71-86): IN_SIZE = AE latent size (64), N_SIGNAL = 128.
`padding_mode` restates the scoped gin binding
`encoder_time/convs.get_padding.mode = 'causal'` (base.gin:55).
"""
import copy

IN_SIZE = 64
N_SIGNAL = 128
SR = 44100
AE_RATIO = 2048  # 16 PQMF bands x strides 2*2*2*4*4


def _denoiser(embed_dim, tcond_dim, window, n_channels=IN_SIZE, cond_dim=6, n_layers=6):
    return dict(n_channels=n_channels, seq_len=N_SIGNAL, embed_dim=embed_dim, cond_dim=cond_dim,
                tcond_dim=tcond_dim, noise_embed_dims=64, n_layers=n_layers, mlp_multiplier=3,
                dropout=0.1, causal=True, pos_emb_type="rotary", local_attention_size=window,
                attention_chunk_size=4)


def _ecapa(channels, in_size=IN_SIZE, out_dim=6):
    return dict(in_size=in_size, out_dim=out_dim, channels=list(channels),
                kernel_sizes=[3, 3, 3, 3], dilations=[1, 1, 1, 1], groups=[1, 1, 1, 1],
                res2net_scale=8, se_channels=128, attention_channels=128, global_context=True,
                pooling=True, use_tanh=False, spherical_normalisation=False,
                regularisation="ac")


def _encoder_time(channels, in_size=IN_SIZE):
    return dict(in_size=in_size, channels=list(channels), ratios=[1, 1, 1, 1], kernel_size=5,
                use_tanh=False, average_out=False, upscale_out=False,
                spherical_normalization=False, vae_regularisation=False,
                ac_regularisation=True, padding_mode="causal")


DIFFUSION = {
    "tiny":
    dict(net=_denoiser(256, 12, 8), encoder=_ecapa([256, 256, 256, 512]),
         encoder_time=_encoder_time([64, 128, 256, 256, 12]), sr=SR, drop_value=-4.0,
         drop_rate=0.2, structure_type="audio"),
    "base":
    dict(net=_denoiser(512, 12, 8), encoder=_ecapa([512, 512, 512, 1024]),
         encoder_time=_encoder_time([64, 128, 256, 512, 12]), sr=SR, drop_value=-4.0,
         drop_rate=0.2, structure_type="audio"),
    "midi":
    dict(net=_denoiser(512, 128, 16), encoder=_ecapa([512, 512, 512, 1024]), encoder_time=None,
         sr=SR, drop_value=-4.0, drop_rate=0.2, structure_type="midi"),
}
DIFFUSION["cycle"] = DIFFUSION["base"]

# reduced-width configs used by the committed golden fixtures (not shipped by
# the reference; same code paths, small enough for the CPU suite)
DIFFUSION["micro"] = dict(
    net=_denoiser(128, 12, 8, n_channels=16, n_layers=2),
    encoder=dict(_ecapa([64, 64, 64, 128], in_size=16), se_channels=16, attention_channels=16),
    encoder_time=_encoder_time([16, 32, 32, 32, 12], in_size=16), sr=SR, drop_value=-4.0,
    drop_rate=0.2, structure_type="audio")
DIFFUSION["micro_midi"] = dict(net=_denoiser(128, 20, 16, n_channels=16, n_layers=2),
                               encoder=DIFFUSION["micro"]["encoder"], encoder_time=None, sr=SR,
                               drop_value=-4.0, drop_rate=0.2, structure_type="midi")

AUTOENCODER = {
    "baseAE":
    dict(in_channels=16, channels=64, pqmf_bands=16, z_channels=64,
         multipliers=[1, 2, 4, 4, 8, 8], factors=[2, 2, 2, 4, 4], dilations=[1, 3, 9],
         kernel_size=3, use_norm=True, decoder_ratio=1.5, use_loudness=True, use_noise=False,
         bottleneck="relu", padding_mode="centered"),
    # reduced width, same topology / strides (ratio 2048) for fixtures
    "microAE":
    dict(in_channels=16, channels=8, pqmf_bands=16, z_channels=16,
         multipliers=[1, 2, 4, 4, 8, 8], factors=[2, 2, 2, 4, 4], dilations=[1, 3, 9],
         kernel_size=3, use_norm=True, decoder_ratio=1.5, use_loudness=True, use_noise=False,
         bottleneck="relu", padding_mode="centered"),
}
# streaming variant of the codec (baseAE.gin:32-33,49): causal padding, no GroupNorm
AUTOENCODER["baseAE_causal"] = dict(AUTOENCODER["baseAE"], use_norm=False,
                                    padding_mode="causal")
AUTOENCODER["microAE_causal"] = dict(AUTOENCODER["microAE"], use_norm=False,
                                     padding_mode="causal")
# no filter bank (SimpleNetsStream.py:853-859, baseAE.gin:15 "Set to 1 if no pqmf"): the codec runs on the mono samples
AUTOENCODER["microAE_nopqmf"] = dict(AUTOENCODER["microAE"], in_channels=1, pqmf_bands=1, multipliers=[1, 2, 4, 4],
                                     factors=[2, 4, 4])
# the decoder's NoiseGenerator branch (SimpleNetsStream.py:499-550, :622-650; baseAE.gin binds use_noise = False)
AUTOENCODER["microAE_noise"] = dict(AUTOENCODER["microAE"], use_noise=True)
# the one-parameter snake of core.py:201-209 as the codec's `activation` (SimpleNetsStream.py:161,169; default: SnakeBeta)
AUTOENCODER["microAE_snake1"] = dict(AUTOENCODER["microAE"], activation="core.Snake")


def diffusion_config(name: str) -> dict:
    if name not in DIFFUSION:
        raise KeyError(f"unknown diffusion config {name!r}; have {sorted(DIFFUSION)}")
    return copy.deepcopy(DIFFUSION[name])


# UNET1D (after/diffusion/networks/unet1d.py:254-268): no shipped gin config selects it; these are
# the constructor defaults adapted to the codec latent sizes, plus a reduced one for the fixtures
UNET = {
    "unet_base": dict(in_size=IN_SIZE, channels=[128, 128, 256, 256], ratios=[2, 2, 2, 2, 2], kernel_size=5,
                      time_channels=64, time_cond_in_channels=12, time_cond_channels=64, cond_channels=6,
                      n_attn_layers=0, use_res_last=False),
    "unet_micro": dict(in_size=16, channels=[32, 32, 64, 64], ratios=[2, 2, 2, 2, 2], kernel_size=5,
                       time_channels=64, time_cond_in_channels=12, time_cond_channels=16, cond_channels=6,
                       n_attn_layers=0, use_res_last=False),
    # two self-attention levels (unet1d.py:339, 350, 372: down_layers 2, 3; up_layers 0, 1; the middle block with 64 // 32 heads)
    "unet_micro_attn": dict(in_size=16, channels=[32, 32, 64, 64], ratios=[2, 2, 2, 2, 2], kernel_size=5,
                            time_channels=64, time_cond_in_channels=12, time_cond_channels=16, cond_channels=6,
                            n_attn_layers=2, use_res_last=False),
    "unet_micro_flat": dict(in_size=16, out_size=16, channels=[32, 64], ratios=[1, 2], kernel_size=3,
                            time_channels=32, time_cond_in_channels=12, time_cond_channels=16,
                            cond_channels=6, n_attn_layers=0, use_res_last=True),
}


def unet_config(name: str = "unet_base") -> dict:
    import copy
    if name not in UNET:
        raise KeyError(f"unknown UNET1D config {name!r}; have {sorted(UNET)}")
    return copy.deepcopy(UNET[name])


def autoencoder_config(name: str = "baseAE") -> dict:
    if name not in AUTOENCODER:
        raise KeyError(f"unknown autoencoder config {name!r}; have {sorted(AUTOENCODER)}")
    return copy.deepcopy(AUTOENCODER[name])
