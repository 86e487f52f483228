#!/usr/bin/env python3
"""Generate the committed golden vectors by running the REFERENCE's own Python
(/root/reference, imported through refimport.py) -- build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [case ...]

Every fixture is `tests/golden/<case>.npz` holding: `meta` (json: config names,
seed, state-dict key->shape of the reference module), the inputs, the reference
outputs, and the few non-random buffers (RoPE freqs, PQMF filters).  Weights are
NOT stored: detweights.fill() regenerates them from (shapes, seed).  Nothing of
the reference's source text is stored -- only numbers it computed.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import detweights  # noqa: E402
import refimport  # noqa: E402
from after_amd import configs  # noqa: E402

torch.set_grad_enabled(False)
R = refimport.modules()


def refill(module, seed, wg_scale=1.0):
    """Overwrite the reference module's parameters/buffers with detweights values."""
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    new = detweights.fill(shapes, seed, wg_scale=wg_scale)
    sd.update(new)
    module.load_state_dict(sd)
    module.eval()
    keep = {k: v.numpy() for k, v in module.state_dict().items() if detweights.KEEP.search(k)
            and not k.endswith(".pad") and "cache" not in k}
    return shapes, keep


def build_denoiser(cfg, cache=0, max_steps=16, max_batch=4):
    kw = dict(cfg)
    if cache:
        orig = R.transformerv2.MHAttention.__init__

        def patched(self, *a, **k):
            k.setdefault("max_cache_size", cache)
            k["max_diffusion_steps"] = max_steps
            k["max_batch_size"] = max_batch
            orig(self, *a, **k)

        R.transformerv2.MHAttention.__init__ = patched
        try:
            net = R.transformerv2.DenoiserV2(**kw)
        finally:
            R.transformerv2.MHAttention.__init__ = orig
        return net
    return R.transformerv2.DenoiserV2(**kw)


def build_encoder_time(cfg):
    kw = {k: v for k, v in cfg.items() if k != "padding_mode"}
    R.cc.set_padding_mode(cfg["padding_mode"])
    try:
        return R.encoder.Encoder1D(**kw)
    finally:
        R.cc.set_padding_mode("centered")


def build_ecapa(cfg):
    return R.ecapa.ECAPATDNN(**cfg)


def build_ae(cfg, bottleneck=None):
    kw = {k: v for k, v in cfg.items() if k not in ("padding_mode", "bottleneck")}
    if kw.get("activation") == "core.Snake":  # the reference takes the class itself
        import importlib
        kw["activation"] = importlib.import_module("after.autoencoder.core").Snake
    R.cc.set_padding_mode(cfg["padding_mode"])
    try:
        return R.ae.AutoEncoder(bottleneck=bottleneck or R.ae.ReluBottleneck(sigma=0.01, scale=3), **kw)
    finally:
        R.cc.set_padding_mode("centered")


def save(case, meta, **arrays):
    path = os.path.join(HERE, case + ".npz")
    arrays = {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
              for k, v in arrays.items()}
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
                        **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def keepdict(keep):
    return {"keep:" + k: v for k, v in keep.items()}


# ---------------------------------------------------------------- denoiser
def denoiser_case(case, cfg_name, B, T, seed, steps_list, with_base_sample=False):
    dcfg = configs.diffusion_config(cfg_name)
    ncfg = dcfg["net"]
    net = build_denoiser(ncfg)
    shapes, keep = refill(net, seed)
    C, ZT, ZS = ncfg["n_channels"], ncfg["cond_dim"], ncfg["tcond_dim"]
    x = detweights.seeded_tensor("x", (B, C, T), seed)
    cond = detweights.seeded_tensor("cond", (B, ZT), seed)
    tc = detweights.seeded_tensor("time_cond", (B, ZS, T), seed)
    if dcfg["structure_type"] == "midi":  # piano-roll like: sparse non-negative
        tc = (tc.abs() * (tc > 1.0)).clamp(max=1.0)
    tvec = torch.linspace(0.05, 0.95, B).reshape(B, 1, 1)
    out = {}
    out["net_out"] = net(x, time=tvec, cond=cond, time_cond=tc)
    model = R.model.RectifiedFlow(net=net, sr=44100, drop_value=-4.0)
    t03 = torch.full((B, 1, 1), 0.3)
    out["mf_2_1"] = model.model_forward(x, t03, cond, tc, 2.0, 1.0)
    out["mf_1_3"] = model.model_forward(x, t03, cond, tc, 1.0, 3.0)
    for n in steps_list:
        out[f"sample_{n}_2_1"] = model.sample(x, cond, tc, n, 2.0, 1.0)
    meta = dict(kind="denoiser", config=cfg_name, seed=seed, B=B, T=T, shapes=shapes,
                steps=steps_list)
    save(case, meta, x=x, cond=cond, time_cond=tc, tvec=tvec, **out, **keepdict(keep))


def mask_case():
    rows = {}
    for (L, cs, W) in [(32, 4, 8), (37, 4, 8), (64, 4, 16), (20, 4, 3), (16, 4, 0), (24, 8, 5)]:
        rows[f"sw_{L}_{cs}_{W}"] = R.transformerv2.combined_sliding_chunkwise_mask(L, cs, W)
    rows["sw_24_4_-1"] = R.transformerv2.combined_sliding_chunkwise_mask(24, 4, -1)
    rows["cw_24_4"] = R.transformerv2.chunk_wise_causal_mask(24, 4)
    rot = R.rotary.RotaryEmbedding(32)
    q = detweights.seeded_tensor("q", (1, 2, 6, 64), 3)
    k = detweights.seeded_tensor("k", (1, 2, 14, 64), 3)
    rq, rk = rot.rotate_queries_with_cached_keys(q, k)
    pe = R.transformerv2.PositionalEmbedding(64, 10_000, 100.0)(torch.linspace(0, 1, 11))
    save("mask_rope", dict(kind="mask_rope"), q=q, k=k, rq=rq, rk=rk, pos_emb=pe,
         freqs=rot.freqs.detach(), **rows)


def stream_case(case, cfg_name, seed, n_chunks=10, chunk=4, steps=3):
    """Streaming path (export.py:398-416): per-diffusion-step KV caches, roll_cache."""
    dcfg = configs.diffusion_config(cfg_name)
    ncfg = dcfg["net"]
    W = ncfg["local_attention_size"]
    net = build_denoiser(ncfg, cache=W, max_steps=steps, max_batch=6)
    shapes, keep = refill(net, seed)
    shapes = {k: v for k, v in shapes.items() if "cache" not in k}
    C, ZT, ZS = ncfg["n_channels"], ncfg["cond_dim"], ncfg["tcond_dim"]
    B = 2
    T = n_chunks * chunk
    x = detweights.seeded_tensor("x", (B, C, T), seed)
    cond = detweights.seeded_tensor("cond", (B, ZT), seed)
    tc = detweights.seeded_tensor("time_cond", (B, ZS, T), seed)
    tvals = torch.linspace(0, 1, steps + 1)[:-1]
    outs = []
    for c in range(n_chunks):
        sl = slice(c * chunk, (c + 1) * chunk)
        per_step = []
        for i, t in enumerate(tvals):
            tt = t.reshape(1, 1, 1).repeat(B, 1, 1)
            per_step.append(net(x[..., sl], time=tt, cond=cond, time_cond=tc[..., sl],
                                cache_index=i))
            net.roll_cache(chunk, i)
        outs.append(torch.stack(per_step, 0))
    out = torch.stack(outs, 0)  # [chunks, steps, B, C, chunk]
    meta = dict(kind="denoiser_stream", config=cfg_name, seed=seed, B=B, T=T, chunk=chunk,
                steps=steps, shapes=shapes)
    save(case, meta, x=x, cond=cond, time_cond=tc, tvals=tvals, out=out, **keepdict(keep))


# ---------------------------------------------------------------- autoencoder
def ae_case(case, cfg_name, B, L, seed, wg_scale=1.0):
    """wg_scale < 1 (the *_wc cases): weight-norm gains scaled so that the activations of the GroupNorm-free
    causal codec stay O(1) through its ~80 layers -- a well-conditioned fixture whose bar is the codec's, not
    the fixture's (with gains ~ 1 the norm-free stack amplifies to |y| ~ 3e3 and the reference's own
    fp32-vs-fp64 error is 2e-3 of the range)."""
    cfg = configs.autoencoder_config(cfg_name)
    ae = build_ae(cfg)
    shapes, keep = refill(ae, seed, wg_scale)
    x = detweights.seeded_tensor("audio", (B, 1, L), seed, 0.1)
    z, _ = ae.encode(x)
    zin = detweights.seeded_tensor("z", tuple(z.shape), seed)
    y = ae.decode(zin)
    if cfg["pqmf_bands"] > 1:
        mb = ae.pqmf(x)
        xr = ae.pqmf.inverse(mb)
    else:  # DummyIdentity: the "multiband" signal is the audio
        mb, xr = x, x
    meta = dict(kind="autoencoder", config=cfg_name, seed=seed, B=B, L=L, shapes=shapes)
    if wg_scale != 1.0:
        meta["wg_scale"] = wg_scale
    save(case, meta, x=x, z=z, zin=zin, y=y, multiband=mb, pqmf_roundtrip=xr, **keepdict(keep))


def noise_case(case="ae_micro_noise", seed=49):
    """Decoder1d with its NoiseGenerator branch (use_noise=True, SimpleNetsStream.py:499-550, :622-650).  The branch
    draws torch.rand_like(ir) inside forward: the draws are recorded here and stored, so that oracle and kernels are
    compared on the reference's own noise.  The last conv's bias is pinned at +5 (stored): with the random-weight
    default the amplitudes are mod_sigmoid(~N(0, 1) - 5) ~ 2e-5 and the branch would vanish below the test's bar."""
    cfg = configs.autoencoder_config("microAE_noise")
    ae = build_ae(cfg)
    shapes, keep = refill(ae, seed)
    sd = ae.state_dict()
    for k in ("decoder.noise_module.net.4.bias", "decoder.synth.branches.1.net.4.bias"):
        sd[k] = sd[k] + 5.0
        keep[k] = sd[k].numpy().copy()
    ae.load_state_dict(sd)
    zin = detweights.seeded_tensor("z", (2, cfg["z_channels"], 8), seed)
    drawn = []
    orig = torch.rand_like

    def recording(t, *a, **k):
        u = orig(t, *a, **k)
        drawn.append(u.clone())
        return u

    torch.rand_like = recording
    try:
        torch.manual_seed(seed)
        y = ae.decode(zin)
    finally:
        torch.rand_like = orig
    assert len(drawn) == 1
    meta = dict(kind="autoencoder_noise", config="microAE_noise", seed=seed, shapes=shapes)
    save(case, meta, zin=zin, y=y, noise_u=drawn[0], **keepdict(keep))


def bottleneck_case(case="ae_micro_bottlenecks", seed=46):
    """The two other bottlenecks of SimpleNetsStream.py:719-785 on the micro codec: TanhBottleneck (sigma = 0:
    deterministic) through AutoEncoder.encode, and VAEBottleneck through encode(return_mean=True) -- mean and KL
    are deterministic, the drawn z is checked through its moments by the tests."""
    cfg = configs.autoencoder_config("microAE")
    x = detweights.seeded_tensor("audio", (2, 1, 8192), seed, 0.1)
    ae_t = build_ae(cfg, R.ae.TanhBottleneck(scale=3, sigma=0.))
    shapes_t, keep = refill(ae_t, seed)
    z_t, reg_t = ae_t.encode(x)
    ae_v = build_ae(cfg, R.ae.VAEBottleneck())
    shapes_v, _ = refill(ae_v, seed + 1)
    torch.manual_seed(0)
    z_v, kl, mean = ae_v.encode(x, return_mean=True)
    meta = dict(kind="bottlenecks", config="microAE", seed=seed, shapes=shapes_t, shapes_vae=shapes_v)
    save(case, meta, x=x, z_tanh=z_t, reg_tanh=reg_t, vae_mean=mean, vae_kl=kl, vae_z=z_v, **keepdict(keep))


def cached_gn_case():
    """CachedGroupNorm(stream=True) (SimpleNetsStream.py:95-147), AFTER's own streaming GroupNorm:
    three chunks of different lengths through one module (window = the first chunk's length)."""
    C, G, B = 12, 4, 3
    gn = R.ae.CachedGroupNorm(G, C, stream=True)
    w = detweights.seeded_tensor("gn_w", (C, ), 71) * 0.3 + 1.0
    b = detweights.seeded_tensor("gn_b", (C, ), 72) * 0.2
    gn.gn.weight.data.copy_(w)
    gn.gn.bias.data.copy_(b)
    lens = (24, 8, 40)
    xs = [detweights.seeded_tensor(f"x{i}", (B, C, n), 73 + i) * (1.0 + i) + 0.5 * i for i, n in enumerate(lens)]
    ys = [gn(x).clone() for x in xs]
    save("cached_gn", dict(kind="cached_gn", C=C, G=G, B=B, lens=list(lens)), weight=w, bias=b,
         **{f"x{i}": x for i, x in enumerate(xs)}, **{f"y{i}": y for i, y in enumerate(ys)})


def pqmf_case():
    p = R.pqmf.CachedPQMF(attenuation=100, n_band=16)
    save("pqmf_bank", dict(kind="pqmf"), hk=p.hk, h=p.h, forward_w=p.forward_conv.weight,
         inverse_w=p.inverse_conv.weight)


# ---------------------------------------------------------------- encoders
def encoders_case(case, cfg_name, B, T, seed):
    dcfg = configs.diffusion_config(cfg_name)
    arrays, meta = {}, dict(kind="encoders", config=cfg_name, seed=seed, B=B, T=T)
    C = dcfg["net"]["n_channels"]
    z = detweights.seeded_tensor("z", (B, C, T), seed)
    arrays["z"] = z
    if dcfg["encoder_time"] is not None:
        et = build_encoder_time(dcfg["encoder_time"])
        meta["shapes_encoder_time"], _ = refill(et, seed)
        arrays["time_cond"] = et(z)
    ec = build_ecapa(dcfg["encoder"])
    meta["shapes_encoder"], _ = refill(ec, seed + 1)
    arrays["cond"] = ec(z[..., :T // 2])
    save(case, meta, **arrays)


# ---------------------------------------------------------------- UNET1D
def unet_case(case, cfg_name, B, T, seed):
    cfg = configs.unet_config(cfg_name)
    net = R.unet1d.UNET1D(**cfg)
    meta = dict(kind="unet1d", config=cfg_name, seed=seed, B=B, T=T)
    meta["shapes"], _ = refill(net, seed)
    x = detweights.seeded_tensor("x", (B, cfg["in_size"], T), seed)
    tc = detweights.seeded_tensor("time_cond", (B, cfg["time_cond_in_channels"], T), seed)
    cond = detweights.seeded_tensor("cond", (B, cfg["cond_channels"]), seed)
    time = torch.linspace(0.1, 0.9, B).reshape(B, 1, 1)
    save(case, meta, x=x, time_cond=tc, cond=cond, time=time, y=net(x, time=time, time_cond=tc, cond=cond))



# ---------------------------------------------------------------- run folders
# config.gin texts in the two styles AFTER leaves next to its checkpoints: the flat
# `gin.operative_config_str()` dump (model.py:262-265) for the diffusion model and the hand-written
# block style of baseAE.gin for the codec.  gin-config is absent from the image, so these are
# written here (dims of a reduced "nano" model so that the REAL checkpoint files fit in the repo).
NANO_RUN_GIN = """# Macros:
# ==============================================================================
IN_SIZE = 16
LOCAL_ATTENTION_SIZE = 8
N_SIGNAL = 128
SR = 44100
ZS_CHANNELS = 12
ZT_CHANNELS = 6

# Parameters for Base:
# ==============================================================================
Base.classifier = @classifier/Encoder1D()
Base.drop_rate = 0.2
Base.drop_value = -4.0
Base.encoder = @encoder/ECAPATDNN()
Base.encoder_time = @encoder_time/Encoder1D()
Base.net = @DenoiserV2()
Base.sr = %SR
Base.time_transform = None

# Parameters for DenoiserV2:
# ==============================================================================
DenoiserV2.attention_chunk_size = 4
DenoiserV2.causal = True
DenoiserV2.cond_dim = %ZT_CHANNELS
DenoiserV2.dropout = 0.1
DenoiserV2.embed_dim = 128
DenoiserV2.local_attention_size = %LOCAL_ATTENTION_SIZE
DenoiserV2.mlp_multiplier = 3
DenoiserV2.n_channels = %IN_SIZE
DenoiserV2.n_layers = 1
DenoiserV2.noise_embed_dims = 64
DenoiserV2.pos_emb_type = 'rotary'
DenoiserV2.seq_len = %N_SIGNAL
DenoiserV2.tcond_dim = %ZS_CHANNELS

# Parameters for encoder/ECAPATDNN:
# ==============================================================================
encoder/ECAPATDNN.attention_channels = 16
encoder/ECAPATDNN.channels = [32, 32, 32, 64]
encoder/ECAPATDNN.dilations = [1, 1, 1, 1]
encoder/ECAPATDNN.global_context = True
encoder/ECAPATDNN.groups = [1, 1, 1, 1]
encoder/ECAPATDNN.in_size = %IN_SIZE
encoder/ECAPATDNN.kernel_sizes = [3, 3, 3, 3]
encoder/ECAPATDNN.out_dim = %ZT_CHANNELS
encoder/ECAPATDNN.pooling = True
encoder/ECAPATDNN.regularisation = 'ac'
encoder/ECAPATDNN.res2net_scale = 8
encoder/ECAPATDNN.se_channels = 16
encoder/ECAPATDNN.spherical_normalisation = False
encoder/ECAPATDNN.use_tanh = False

# Parameters for encoder_time/Encoder1D:
# ==============================================================================
encoder_time/Encoder1D.ac_regularisation = True
encoder_time/Encoder1D.average_out = False
encoder_time/Encoder1D.channels = [16, 16, 32, 32, %ZS_CHANNELS]
encoder_time/Encoder1D.in_size = %IN_SIZE
encoder_time/Encoder1D.kernel_size = 5
encoder_time/Encoder1D.ratios = [1, 1, 1, 1]
encoder_time/Encoder1D.spherical_normalization = False
encoder_time/Encoder1D.upscale_out = False
encoder_time/Encoder1D.use_tanh = False
encoder_time/Encoder1D.vae_regularisation = False

# Parameters for encoder_time/get_padding:
# ==============================================================================
encoder_time/get_padding.mode = 'causal'

# Parameters for classifier/Encoder1D:
# ==============================================================================
classifier/Encoder1D.ac_regularisation = False
classifier/Encoder1D.average_out = True
classifier/Encoder1D.channels = [16, 16, 16, 16, %ZT_CHANNELS]
classifier/Encoder1D.in_size = %ZS_CHANNELS
classifier/Encoder1D.kernel_size = 5
classifier/Encoder1D.ratios = [1, 2, 2, 1]

# Parameters for Base.fit:
# ==============================================================================
Base.fit.lr = 0.0001
Base.fit.use_ema = True
"""

NANO_CODEC_GIN = """from __gin__ import dynamic_registration
from after.autoencoder.networks import SimpleNetsStream
import cached_conv

SR = 44100
LATENT_SIZE = 16
PQMF_BANDS = 16
BASE_CHANNELS = 8
KERNEL_SIZE = 3
DECODER_RATIO = 1.5

SimpleNetsStream.ReluBottleneck:
    sigma = 0.01
    scale = 3

SimpleNetsStream.AutoEncoder:
    in_channels = %PQMF_BANDS
    channels = %BASE_CHANNELS
    pqmf_bands = %PQMF_BANDS
    z_channels = %LATENT_SIZE
    multipliers = [1, 2, 4, 4, 8, 8]
    factors = [2, 2, 2, 4, 4]
    dilations = [1, 3, 9]
    kernel_size = %KERNEL_SIZE
    bottleneck = @SimpleNetsStream.ReluBottleneck()
    use_norm = True
    decoder_ratio = %DECODER_RATIO
    use_loudness = True
    use_noise = False
"""


def checkpoint_case(case="ckpt_nano", seed=71):
    """A run folder as the REFERENCE leaves it: `checkpoint<step>_EMA.pt` written by the
    reference's own `Base.save_model` (model.py:144-176: EMA context, `emb_model.*` stripped,
    `opt_state` beside `model_state`) from reference modules, the codec checkpoint in the layout of
    `Trainer.fit` (trainer.py:350-361), `config.gin` next to each -- plus the reference's outputs
    on seeded inputs.  The .pt files are committed as they were written (a few hundred KiB)."""
    from after_amd.ginfile import GinConfig
    root = os.path.join(HERE, case)
    run, codec = os.path.join(root, "run"), os.path.join(root, "codec")
    os.makedirs(run, exist_ok=True)
    os.makedirs(codec, exist_ok=True)
    open(os.path.join(run, "config.gin"), "w").write(NANO_RUN_GIN)
    open(os.path.join(codec, "config.gin"), "w").write(NANO_CODEC_GIN)
    g = GinConfig.parse_string(NANO_RUN_GIN)
    net = build_denoiser(g.kwargs("DenoiserV2"))
    refill(net, seed)
    ec = build_ecapa(g.kwargs("ECAPATDNN", "encoder"))
    refill(ec, seed + 1)
    et = build_encoder_time(dict(g.kwargs("Encoder1D", "encoder_time"), padding_mode="causal"))
    refill(et, seed + 2)
    cl = R.encoder.Encoder1D(**g.kwargs("Encoder1D", "classifier"))  # outside the sampling path
    refill(cl, seed + 3)
    ga = GinConfig.parse_string(NANO_CODEC_GIN)
    akw = {k: v for k, v in ga.kwargs("AutoEncoder").items() if k != "bottleneck"}
    ae = R.ae.AutoEncoder(bottleneck=R.ae.ReluBottleneck(**ga.kwargs("ReluBottleneck")), **akw)
    refill(ae, seed + 4)
    model = R.model.RectifiedFlow(net=net, sr=44100, encoder=ec, encoder_time=et, classifier=cl,
                                  emb_model=ae, drop_value=-4.0, drop_rate=0.2)
    model.eval()
    # the state save_model expects after fit() has run (model.py:215-231, 401-405)
    model.use_ema = True
    model.ema = R.model.ExponentialMovingAverage(list(model.net.parameters()), decay=0.999)
    model.opt = torch.optim.AdamW(model.net.parameters(), lr=1e-4)
    model.step = 1000
    model.save_model(run)
    opt = torch.optim.AdamW(ae.parameters(), lr=1e-4)
    torch.save({"model_state": ae.state_dict(), "opt_state": opt.state_dict(), "dis_state": {},
                "opt_dis_state": {}}, os.path.join(codec, "checkpoint500.pt"))
    # reference outputs
    B, C, T, nsig = 2, 16, 32, 16
    zs = detweights.seeded_tensor("zs", (B, C, T), seed)
    zt = detweights.seeded_tensor("zt", (B, C, T), seed)
    x0 = detweights.seeded_tensor("x0", (B, C, T), seed)
    audio = detweights.seeded_tensor("audio", (B, 1, 8192), seed, 0.1)
    cond = model.encoder(zt[..., :nsig])
    tc = model.encoder_time(zs)
    z = model.sample(x0, cond, tc, 5, 2.0, 1.5)
    y = ae.decode(z)
    ze, reg = ae.encode(audio)
    saved = torch.load(os.path.join(run, "checkpoint1000_EMA.pt"), map_location="cpu")
    meta = dict(kind="checkpoint", seed=seed, nb_steps=5, guidance=[2.0, 1.5], n_signal_timbre=nsig,
                run_keys=sorted(saved["model_state"].keys()),
                codec_keys=sorted(ae.state_dict().keys()))
    save(case, meta, zs=zs, zt=zt, x0=x0, audio=audio, cond=cond, time_cond=tc, z=z, y=y, z_enc=ze,
         reg=reg)
    for d in (run, codec):
        for f in sorted(os.listdir(d)):
            print(f"  {d}/{f}: {os.path.getsize(os.path.join(d, f)) / 1024:.1f} KiB")


CASES = {
    "pqmf_bank": pqmf_case,
    "cached_gn": cached_gn_case,
    "mask_rope": mask_case,
    "denoiser_micro": lambda: denoiser_case("denoiser_micro", "micro", 2, 32, 11, [4]),
    "denoiser_micro_ragged": lambda: denoiser_case("denoiser_micro_ragged", "micro", 3, 27, 12, [3]),
    "denoiser_micro_midi": lambda: denoiser_case("denoiser_micro_midi", "micro_midi", 2, 40, 13, [4]),
    "denoiser_tiny": lambda: denoiser_case("denoiser_tiny", "tiny", 1, 256, 21, [4, 50]),
    "denoiser_base": lambda: denoiser_case("denoiser_base", "base", 1, 256, 22, [50]),
    "denoiser_midi": lambda: denoiser_case("denoiser_midi", "midi", 2, 64, 23, [4]),
    "stream_micro": lambda: stream_case("stream_micro", "micro", 31),
    "ae_micro": lambda: ae_case("ae_micro", "microAE", 2, 16384, 41),
    "ae_micro_causal": lambda: ae_case("ae_micro_causal", "microAE_causal", 1, 8192, 42),
    "ae_base": lambda: ae_case("ae_base", "baseAE", 1, 32768, 43),
    "ae_micro_bottlenecks": bottleneck_case,
    "ae_micro_noise": noise_case,
    "ae_micro_nopqmf": lambda: ae_case("ae_micro_nopqmf", "microAE_nopqmf", 2, 4096, 47),
    "ae_micro_snake1": lambda: ae_case("ae_micro_snake1", "microAE_snake1", 2, 8192, 48),
    "ae_micro_causal_wc": lambda: ae_case("ae_micro_causal_wc", "microAE_causal", 2, 8192, 44, wg_scale=0.5),
    "ae_base_causal_wc": lambda: ae_case("ae_base_causal_wc", "baseAE_causal", 1, 16384, 45, wg_scale=0.5),
    "encoders_micro": lambda: encoders_case("encoders_micro", "micro", 2, 64, 51),
    "encoders_tiny": lambda: encoders_case("encoders_tiny", "tiny", 1, 256, 52),
    "encoders_base": lambda: encoders_case("encoders_base", "base", 1, 256, 53),
    "unet_micro": lambda: unet_case("unet_micro", "unet_micro", 2, 64, 61),
    "unet_micro_flat": lambda: unet_case("unet_micro_flat", "unet_micro_flat", 3, 24, 62),
    "unet_micro_attn": lambda: unet_case("unet_micro_attn", "unet_micro_attn", 2, 64, 63),
    "ckpt_nano": checkpoint_case,
}

if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    names = sys.argv[1:] or list(CASES)
    for n in names:
        CASES[n]()
