"""GEMM microbench: python scripts/bench_gemm.py  (env knobs AFTER_GEMM_BK / AFTER_GEMM_LDS_MIN).
Tiles (mt, nt): classic 2x2-wave kernels; (100+MB, 10*NS+NB): balanced split-K kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from after_amd import diag
dev = torch.device("cuda:0")
shapes = [(6144,1536,512),(6144,512,1536)]
tiles = [(2,2),(104,23),(104,22),(304,23),(304,33),(304,22),(306,23),(306,22)]
for (M,N,K) in shapes:
    a = torch.randn(M,K,device=dev); w = torch.randn(N,K,device=dev); out = torch.empty(M,N,device=dev)
    ref = (a.double() @ w.double().T)
    res = []
    for tile in tiles:
        out.zero_()
        diag.gemm(a,w,tile=tile,out=out)
        err = (out.double()-ref).abs().max().item()
        for _ in range(30): diag.gemm(a,w,tile=tile,out=out)   # also lets the clocks settle: the first config of a run is otherwise ~15 % slow
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        reps=50
        e0.record()
        for _ in range(reps): diag.gemm(a,w,tile=tile,out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1)/reps*1e3
        res.append(f"{tile}:{us:.1f}us/{2*M*N*K/us/1e6:.0f}TF" + ("" if err < 1e-3 else f"/ERR{err:.1e}"))
    print(f"M={M} N={N} K={K}  " + "  ".join(res), flush=True)
