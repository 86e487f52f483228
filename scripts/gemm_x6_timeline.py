"""Per-workgroup phase cycles of the experimental gemm_x6 kernel (wave 0: DMA wait, barrier, DMA issue, LDS reads,
split + MFMAs).  python scripts/gemm_x6_timeline.py M N K tile"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from after_amd import _lib, diag

dev = torch.device("cuda:0")
M, N, K, tile = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (6144, 1536, 512, 431)
a = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev)
w3 = diag.split_x6(w)
out = torch.empty(M, N, device=dev)
for _ in range(5):
    diag.gemm_x6(a, w3, tile=tile, out=out)
torch.cuda.synchronize()
dbg = torch.zeros(1 << 16, dtype=torch.int64, device=dev)
_lib.lib().after_gemm_x6_set_debug(dbg.data_ptr())
diag.gemm_x6(a, w3, tile=tile, out=out)
torch.cuda.synchronize()
_lib.lib().after_gemm_x6_set_debug(None)
d = dbg.cpu().numpy().reshape(-1, 8).astype(np.float64)
d = d[d[:, 5] > 0]
names = ["dma wait", "barrier", "dma issue", "lds reads", "split+mfma", "lifetime"]
print(f"{M}x{N}x{K} tile {tile}: {len(d)} workgroups; median cycles per workgroup (wave 0):",
      {n: int(np.median(d[:, i])) for i, n in enumerate(names)})
