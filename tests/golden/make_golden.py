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


def refill(module, seed):
    """Overwrite the reference module's parameters/buffers with detweights values."""
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    new = detweights.fill(shapes, seed)
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


def build_ae(cfg):
    kw = {k: v for k, v in cfg.items() if k not in ("padding_mode", "bottleneck")}
    R.cc.set_padding_mode(cfg["padding_mode"])
    try:
        return R.ae.AutoEncoder(bottleneck=R.ae.ReluBottleneck(sigma=0.01, scale=3), **kw)
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
def ae_case(case, cfg_name, B, L, seed):
    cfg = configs.autoencoder_config(cfg_name)
    ae = build_ae(cfg)
    shapes, keep = refill(ae, seed)
    x = detweights.seeded_tensor("audio", (B, 1, L), seed, 0.1)
    z, _ = ae.encode(x)
    zin = detweights.seeded_tensor("z", tuple(z.shape), seed)
    y = ae.decode(zin)
    mb = ae.pqmf(x)
    xr = ae.pqmf.inverse(mb)
    meta = dict(kind="autoencoder", config=cfg_name, seed=seed, B=B, L=L, shapes=shapes)
    save(case, meta, x=x, z=z, zin=zin, y=y, multiband=mb, pqmf_roundtrip=xr, **keepdict(keep))


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


CASES = {
    "pqmf_bank": pqmf_case,
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
    "encoders_micro": lambda: encoders_case("encoders_micro", "micro", 2, 64, 51),
    "encoders_tiny": lambda: encoders_case("encoders_tiny", "tiny", 1, 256, 52),
    "encoders_base": lambda: encoders_case("encoders_base", "base", 1, 256, 53),
    "unet_micro": lambda: unet_case("unet_micro", "unet_micro", 2, 64, 61),
    "unet_micro_flat": lambda: unet_case("unet_micro_flat", "unet_micro_flat", 3, 24, 62),
}

if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    names = sys.argv[1:] or list(CASES)
    for n in names:
        CASES[n]()
