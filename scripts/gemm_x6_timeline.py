"""Per-workgroup cycle stamps of gemm_x6_kernel (diagnostics): prologue (launch -> slab 0 landed), main loop,
epilogue (k-part reduction + stores), start / end spread over the chip.
    python scripts/gemm_x6_timeline.py M N K tile [epilogue]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from after_amd import _lib, diag

dev = torch.device("cuda:0")
M, N, K, tile = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (768, 1536, 512, 1)
epi = int(sys.argv[5]) if len(sys.argv) > 5 else 0
a = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev)
a3, w3 = diag.split_x6(a), diag.split_x6(w)
out = torch.empty(M, N, device=dev)
res = torch.randn(M, N, device=dev) if epi == 2 else None
for _ in range(5):
    diag.gemm_x6(a3, w3, tile=tile, out=out, epilogue=epi, residual=res)
torch.cuda.synchronize()
dbg = torch.zeros(1 << 16, dtype=torch.int64, device=dev)
_lib.lib().after_gemm_x6_set_debug(dbg.data_ptr())
diag.gemm_x6(a3, w3, tile=tile, out=out, epilogue=epi, residual=res)
torch.cuda.synchronize()
_lib.lib().after_gemm_x6_set_debug(None)
d = dbg.cpu().numpy().reshape(-1, 8).astype(np.float64)
d = d[d[:, 5] > 0]
rs = (d[:, 4] - d[:, 4].min()) * 10.0  # ns (100 MHz clock)
re = (d[:, 5] - d[:, 4].min()) * 10.0
print(f"{M}x{N}x{K} tile {tile} epi {epi}: {len(d)} workgroups")
print(f"  per-WG cycles (wave 0): prologue med {np.median(d[:,1]-d[:,0]):.0f}  loop med {np.median(d[:,2]-d[:,1]):.0f} "
      f"(min {np.min(d[:,2]-d[:,1]):.0f} max {np.max(d[:,2]-d[:,1]):.0f})  epilogue med {np.median(d[:,3]-d[:,2]):.0f}  "
      f"lifetime med {np.median(d[:,3]-d[:,0]):.0f} max {np.max(d[:,3]-d[:,0]):.0f}")
print(f"  wall: last start {rs.max():.0f} ns, first end {re.min():.0f} ns, last end {re.max():.0f} ns")
u, c = np.unique(d[:, 6].astype(np.int64), return_counts=True)
print(f"  distinct CU ids {len(u)}; WGs per CU histogram:", dict(zip(*np.unique(c, return_counts=True))))
