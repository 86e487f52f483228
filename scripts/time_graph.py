"""Does hipGraph capture of sample() help? (torch.cuda.graph around the C-ABI call)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from after_amd import DenoiserV2, RectifiedFlow, configs
dev = torch.device("cuda:0")
torch.manual_seed(0)
dcfg = configs.diffusion_config("base")
net = DenoiserV2(**dcfg["net"]); model = RectifiedFlow(net=net, sr=44100, device=dev)
B, T, steps = 1, 256, 50
x0 = torch.randn(B, 64, T, device=dev); cond = torch.randn(B, 6, device=dev); tc = torch.randn(B, 12, T, device=dev)
out = torch.empty_like(x0)
for _ in range(3): net.cfg_sample(x0, cond, tc, steps, 2.0, 1.0, -4.0, out=out)
torch.cuda.synchronize()
def timeit(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
eager = timeit(lambda: net.cfg_sample(x0, cond, tc, steps, 2.0, 1.0, -4.0, out=out))
ref = out.clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    net.cfg_sample(x0, cond, tc, steps, 2.0, 1.0, -4.0, out=out)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    net.cfg_sample(x0, cond, tc, steps, 2.0, 1.0, -4.0, out=out)
out.zero_()
graph = timeit(lambda: g.replay())
print(f"eager {eager:.2f} ms, graph replay {graph:.2f} ms, same result: {torch.equal(out, ref)}")
