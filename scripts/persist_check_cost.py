"""Cost of the default (checked) mode of the offline persistent samplers: after_sample synchronises its stream once behind the
persistent launch and looks at the launch's failure words (include/after_hip.h: after_denoiser_set_persist_check).  The whole
clip -- encoders, sampler, decode: bench.py's step -- with the check (default) and without (set_persist_check(False)),
interleaved on one box, one clip and eight."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from after_amd import pipeline

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
model, dcfg, acfg = pipeline.build_models("base", "baseAE", dev, seed=0)
for B in (1, 8):
    g = torch.Generator().manual_seed(B)
    zs, zt, x0 = (torch.randn(B, 64, 256, generator=g).to(dev) for _ in range(3))

    def step():
        return pipeline.generate_from_latents(model, zs, zt, x0, nb_steps=50, guidance_timbre=2.0, guidance_structure=1.0)[0]

    res = {True: [], False: []}
    for rep in range(6):
        for checked in (True, False):
            model.net.set_persist_check(None if checked else False)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            n = 20 if B == 1 else 6
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            res[checked].append((time.perf_counter() - t0) / n * 1e3)
    model.net.set_persist_check(None)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(json.dumps({"workload": f"base, {B} clip(s): encoders + 50-step sampler + decode (bench.py's step)", "sampler_path": model.net.sample_path(),
                      "ms_per_step_checked_default": round(med[True], 3), "ms_per_step_deferred": round(med[False], 3),
                      "overhead_pct": round(100.0 * (med[True] / med[False] - 1.0), 2), "rounds": 6}))
