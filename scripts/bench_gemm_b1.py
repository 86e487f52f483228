"""B = 1 GEMM shapes (768 tokens) over tile configurations: trains of launches, and correctness vs fp64.
    python scripts/bench_gemm_b1.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from after_amd import diag

dev = torch.device("cuda:0")
shapes = [(768, 1536, 512), (768, 512, 1536), (768, 512, 512)]
tiles = [(2, 2), (103, 21), (103, 23), (203, 21), (203, 23), (103, 21)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    out = torch.empty(M, N, device=dev)
    ref = a.double() @ w.double().T
    res = []
    for tile in tiles:
        out.zero_()
        try:
            diag.gemm(a, w, tile=tile, out=out)
        except Exception as e:
            res.append(f"{tile}:n/a")
            continue
        err = (out.double() - ref).abs().max().item()
        for _ in range(50):
            diag.gemm(a, w, tile=tile, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        reps = 200
        e0.record()
        for _ in range(reps):
            diag.gemm(a, w, tile=tile, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        res.append(f"{tile}:{us:.2f}us/{2 * M * N * K / us / 1e6:.0f}TF" + ("" if err < 1e-3 else f"/ERR{err:.1e}"))
    print(f"M={M} N={N} K={K}  " + "  ".join(res), flush=True)
