"""EXPERIMENTAL gemm_x6 (fp32 products on the bf16 matrix pipe) against the fp32 MFMA GEMM: error vs fp64
and trains of launches at the sampler's shapes.
    python scripts/bench_gemm_x6.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from after_amd import diag

dev = torch.device("cuda:0")


def timeit(fn, reps=200):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator(device="cpu").manual_seed(0)
for (M, N, K) in [(768, 1536, 512), (768, 512, 1536), (6144, 1536, 512), (6144, 512, 1536), (100, 96, 64)]:
    a = (1.3 * torch.randn(M, K, generator=g)).to(dev)
    a[::7, ::13] *= 30.0
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ref = a.double() @ w.double().T + bias.double()
    out = torch.empty(M, N, device=dev)
    w3 = diag.split_x6(w)
    diag.gemm(a, w, bias=bias, out=out)
    e32 = (out.double() - ref).abs().max().item()
    t32 = timeit(lambda: diag.gemm(a, w, bias=bias, out=out)) if M >= 768 else 0.0
    line = f"M={M} N={N} K={K}  fp32: {t32:.2f}us err {e32:.2e} |"
    tiles = [0, 332, 312, 1332, 1312, 431, 1431, 631, 1631, 831, 1831, 861] if M >= 768 else [0, 431, 332, 1431, 831]
    for t in tiles:
        try:
            out.zero_()
            diag.gemm_x6(a, w3, bias=bias, tile=t, out=out)
            torch.cuda.synchronize()
        except Exception as e:
            line += f" {t}:n/a"
            continue
        err = (out.double() - ref).abs().max().item()
        us = timeit(lambda: diag.gemm_x6(a, w3, bias=bias, tile=t, out=out)) if M >= 768 else 0.0
        line += f" {t}:{us:.2f}us/{err:.1e}"
    print(line, flush=True)
