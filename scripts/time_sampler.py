"""Ad-hoc timing of the HIP sampler (base config, random init)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from after_amd import DenoiserV2, RectifiedFlow, configs

cfgname = sys.argv[1] if len(sys.argv) > 1 else "base"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
dcfg = configs.diffusion_config(cfgname)
net = DenoiserV2(**dcfg["net"])
model = RectifiedFlow(net=net, sr=44100, device=dev)
T = int(os.environ.get("AFTER_T", "256"))
x0 = torch.randn(B, 64, T, device=dev)
cond = torch.randn(B, 6, device=dev)
tc = torch.randn(B, dcfg["net"]["tcond_dim"], T, device=dev)
if os.environ.get("AFTER_TIME_GEMM_PATH"):  # 3: the opt-in bf16 tolerance tier
    net.set_gemm_path(int(os.environ["AFTER_TIME_GEMM_PATH"]))
out = model.sample(x0, cond, tc, steps, 2.0, 1.0)
torch.cuda.synchronize()
if os.environ.get("AFTER_PROFILE_GEMM"):
    net.profile(True)
ts = []
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model.sample(x0, cond, tc, steps, 2.0, 1.0)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ms, n, fl = net.gemm_time() if os.environ.get("AFTER_PROFILE_GEMM") else (1e-9, reps, 0.0)
print(f"{cfgname} B={B} steps={steps}: sample {min(ts)*1e3:.2f} ms (min of {reps}), {min(ts)/steps*1e6:.1f} us/step; "
      f"GEMM {ms/reps:.2f} ms/sample over {n//reps} launches, {fl/ms/1e9:.1f} TFLOP/s in-GEMM; "
      f"audio {B*11.889/min(ts):.1f}x RT (latent stage only)")
