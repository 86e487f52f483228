"""GEMM microbench: python scripts/bench_gemm.py  (env knobs AFTER_GEMM_BK / AFTER_GEMM_LDS_MIN)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from after_amd import diag
dev = torch.device("cuda:0")
shapes = [(768,1536,512),(768,512,1536),(6144,1536,512),(6144,512,1536)]
tiles = [(1,1),(1,2),(2,2),(4,2),(4,4)]
want = sys.argv[1:] 
for (M,N,K) in shapes:
    a = torch.randn(M,K,device=dev); w = torch.randn(N,K,device=dev); out = torch.empty(M,N,device=dev)
    res = []
    for tile in tiles:
        for _ in range(3): diag.gemm(a,w,tile=tile,out=out)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        reps=50
        e0.record()
        for _ in range(reps): diag.gemm(a,w,tile=tile,out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1)/reps*1e3
        res.append(f"{tile}:{us:.1f}us/{2*M*N*K/us/1e6:.0f}TF")
    print(f"BK={os.environ.get('AFTER_GEMM_BK','32')} LDSMIN={os.environ.get('AFTER_GEMM_LDS_MIN','0')} M={M} N={N} K={K}  " + "  ".join(res))
