import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from after_amd import AutoEncoder, configs
dev = torch.device("cuda:0")
for name in ("baseAE", "baseAE_causal"):
    cfg = configs.autoencoder_config(name); cfg.pop("bottleneck")
    ae = AutoEncoder(**cfg).to(dev)
    z = torch.randn(1, 64, 256, device=dev)
    ae.decode(z); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ae.decode(z); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(name, "decode", round(min(ts) * 1e3, 3), "ms")
