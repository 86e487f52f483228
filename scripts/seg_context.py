"""Where does a clip's time go in the bench's context?  The persistent offline sampler alone (random conditioning / the
encoders' conditioning), then with the encoders and / or the decode between its launches (same process, same box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from after_amd import pipeline

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
model, dcfg, acfg = pipeline.build_models("base", "baseAE", dev, seed=0)
g = torch.Generator(device="cpu").manual_seed(1000)
T = 256
zs, zt, x0 = (torch.randn(1, 64, T, generator=g).to(dev) for _ in range(3))
cond_r = torch.randn(1, 6, device=dev)
tc_r = torch.randn(1, dcfg["net"]["tcond_dim"], T, device=dev)
cond_e = model.encoder(zt[..., :128])
tc_e = model.encoder_time(zs)
print("cond_e", tuple(cond_e.shape), float(cond_e.abs().max()), "tc_e", tuple(tc_e.shape), float(tc_e.abs().max()), float(tc_e.abs().mean()))
z = model.sample(x0, cond_e, tc_e, 50, 2.0, 1.0)


def t(label, fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{label}: min {min(ts) * 1e3:.2f} ms, median {sorted(ts)[len(ts) // 2] * 1e3:.2f} ms", flush=True)


for rep in range(2):
    t("sample, random cond", lambda: model.sample(x0, cond_r, tc_r, 50, 2.0, 1.0))
    t("sample, encoder cond", lambda: model.sample(x0, cond_e, tc_e, 50, 2.0, 1.0))
    t("sample, encoder cond x 0 (zeros)", lambda: model.sample(x0, cond_e * 0, tc_e * 0, 50, 2.0, 1.0))
    t("encoders", lambda: (model.encoder(zt[..., :128]), model.encoder_time(zs)))
    t("decode", lambda: model.emb_model.decode(z))
    t("encoders + sample", lambda: model.sample(x0, model.encoder(zt[..., :128]), model.encoder_time(zs), 50, 2.0, 1.0))
    t("sample + decode", lambda: model.emb_model.decode(model.sample(x0, cond_e, tc_e, 50, 2.0, 1.0)))
    t("full step", lambda: pipeline.generate_from_latents(model, zs, zt, x0, nb_steps=50, guidance_timbre=2.0, guidance_structure=1.0))
    t("10 full steps / 10", lambda: [pipeline.generate_from_latents(model, zs, zt, x0, nb_steps=50, guidance_timbre=2.0, guidance_structure=1.0) for _ in range(10)], reps=3)
