import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from after_amd import diag
dev = torch.device("cuda:0")
def t(shape, tile, epi, reps=300):
    M, N, K = shape
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
    for _ in range(5): diag.gemm(a, w, tile=tile, out=out, epilogue=epi)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): diag.gemm(a, w, tile=tile, out=out, epilogue=epi)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 2)
for shape, tile in [((768, 1536, 512), (203, 23)), ((768, 1536, 512), (103, 21)), ((768, 512, 1536), (203, 21))]:
    print(shape, tile, {f"epi{f:#x}": t(shape, tile, f) for f in (0, 0x200)})
