"""Streaming-size GEMMs (M = 48 / 96 tokens: 4 frames x 3 CFG rows x 4 / 8 streams) over tile configurations.
    python scripts/bench_gemm_stream.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from after_amd import diag
dev = torch.device("cuda:0")
for (M, N, K) in [(96, 1536, 512), (96, 512, 1536), (96, 512, 512), (48, 1536, 512)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
    ref = a.double() @ w.double().T
    res = []
    for tile in [(2, 2), (0, 0), (203, 21), (203, 31), (203, 41), (103, 21), (103, 41), (202, 22), (202, 32), (403, 0), (406, 0), (0, 0)]:
        try:
            out.zero_(); diag.gemm(a, w, tile=tile, out=out)
        except Exception as e:
            res.append(f"{tile}:n/a"); continue
        err = (out.double() - ref).abs().max().item()
        for _ in range(50): diag.gemm(a, w, tile=tile, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): diag.gemm(a, w, tile=tile, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 300 * 1e3
        res.append(f"{tile}:{us:.2f}" + ("" if err < 1e-3 else f"/ERR{err:.1e}"))
    print(f"M={M} N={N} K={K}  " + "  ".join(res), flush=True)
