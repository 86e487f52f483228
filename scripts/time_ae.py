"""Ad-hoc timing of the HIP autoencoder (baseAE, random init)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from after_amd import AutoEncoder, configs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = configs.autoencoder_config("baseAE"); cfg.pop("bottleneck")
ae = AutoEncoder(**cfg).to(dev)
x = 0.1 * torch.randn(B, 1, 524288, device=dev)
z = torch.randn(B, 64, 256, device=dev)
for name, fn in (("encode", lambda: ae.encode(x)), ("decode", lambda: ae.decode(z))):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    gf = {"encode": 45.2, "decode": 95.3}[name] * B
    print(f"AE {name} B={B}: {min(ts)*1e3:.2f} ms  ({gf/min(ts)/1e3:.1f} TFLOP/s)")
