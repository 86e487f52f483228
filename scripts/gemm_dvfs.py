"""Is the small-GEMM rate clock-limited?  Time the same launch in short and long back-to-back
trains, and right after a long MFMA-heavy warm-up."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from after_amd import diag
dev = torch.device("cuda:0")
M, N, K = 768, 1536, 512
a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
A = torch.randn(6144, 512, device=dev); W = torch.randn(6144, 512, device=dev); O = torch.empty(6144, 6144, device=dev)
def t(tile, reps):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): diag.gemm(a, w, tile=tile, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for tile in [(1, 1), (103, 33)]:
    t(tile, 10)
    print(tile, "train of 20:", round(t(tile, 20), 2), "us;  200:", round(t(tile, 200), 2), "us;  3000:", round(t(tile, 3000), 2), "us")
    for _ in range(40): diag.gemm(A, W, tile=(2, 2), out=O)   # ~40 x 0.4 ms of dense MFMA
    print(tile, "after heavy warm-up, train of 20:", round(t(tile, 20), 2), "us")
    torch.cuda.synchronize()
    import time; time.sleep(0.5)
    print(tile, "after 0.5 s idle, train of 20:", round(t(tile, 20), 2), "us")
