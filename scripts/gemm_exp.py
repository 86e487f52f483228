import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from after_amd import diag
dev = torch.device("cuda:0")
M, N, K = 768, 1536, 512
a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
def t(tile, epi, reps=200):
    for _ in range(5): diag.gemm(a, w, tile=tile, out=out, epilogue=epi)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): diag.gemm(a, w, tile=tile, out=out, epilogue=epi)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 2)
for tile in [(103, 33), (103, 23)]:
    print(tile, {f"flags{f}": t(tile, f << 8) for f in (0, 1, 2, 3, 4, 5, 6, 7)})
