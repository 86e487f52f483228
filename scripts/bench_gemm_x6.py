"""gemm_x6 (fp32 products on the bf16 matrix pipe, pre-split operands) against the fp32 MFMA GEMM at the sampler's
shapes: error vs fp64 and trains of launches, every tile of the table.  One JSON line per (shape, tile).
    python scripts/bench_gemm_x6.py [out.jsonl]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from after_amd import _lib, diag

dev = torch.device("cuda:0")
out_path = sys.argv[1] if len(sys.argv) > 1 else None
PEAK32, PEAK6 = 157.3, 2500.0 / 6.0


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
rows = []
SHAPES = [(768, 1536, 512, 0), (768, 1536, 512, 1), (768, 512, 1536, 2), (1536, 1536, 512, 0), (1536, 512, 1536, 2),
          (3072, 1536, 512, 0), (3072, 512, 1536, 2), (6144, 1536, 512, 0), (6144, 1536, 512, 1), (6144, 512, 1536, 2),
          (12288, 1536, 512, 1)]
for (M, N, K, epi) in SHAPES:
    a = (1.3 * torch.randn(M, K, generator=g)).to(dev)
    a[::7, ::13] *= 30.0
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev) if epi == 2 else None
    ref = a.double() @ w.double().T + bias.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + res.double()
    out = torch.empty(M, N, device=dev)
    out3 = diag.X6Planes.empty(M, N, dev)
    w3, a3 = diag.split_x6(w), diag.split_x6(a)
    kw = dict(bias=bias, residual=res, epilogue=epi)
    diag.gemm(a, w, out=out, **kw)
    e32 = (out.double() - ref).abs().max().item()
    r32 = (out.double() - ref).pow(2).mean().sqrt().item()
    split_err = {}
    for kparts, ft in ((1, (304, 23)), (2, (103, 21)), (4, (203, 21))):  # the fp32 kernel with 1 / 2 / 4 k-parts
        diag.gemm(a, w, out=out, tile=ft, **kw)
        d = out.double() - ref
        split_err[kparts] = (d.abs().max().item(), d.pow(2).mean().sqrt().item())
    t32 = timeit(lambda: diag.gemm(a, w, out=out, **kw))
    fl = 2.0 * M * N * K
    auto = _lib.lib().after_gemm_x6_pick_tile(M, N, K)
    line = f"M={M} N={N} K={K} epi={epi}  fp32 {t32:.2f}us ({fl / t32 * 1e-6:.0f} TF) err {e32:.1e} | auto={auto} |"
    for t in list(range(1, 10)) + [11, 12, 15]:
        try:
            out.zero_()
            diag.gemm_x6(a3, w3, tile=t, out=out, **kw)
            torch.cuda.synchronize()
        except Exception:
            line += f" {t}:n/a"
            continue
        err = (out.double() - ref).abs().max().item()
        rms = (out.double() - ref).pow(2).mean().sqrt().item()
        us = timeit(lambda: diag.gemm_x6(a3, w3, tile=t, out=out, **kw))
        us3 = timeit(lambda: diag.gemm_x6(a3, w3, tile=t, out=out3, planes=True, **kw)) if epi == 1 else None
        line += f" {t}:{us:.2f}" + (f"/{us3:.2f}" if us3 else "") + f"us/{err:.1e}"
        rows.append({"M": M, "N": N, "K": K, "epilogue": epi, "tile": t, "auto_tile": auto, "us": round(us, 2),
                     "us_plane_out": round(us3, 2) if us3 else None, "tflops": round(fl / us * 1e-6, 1),
                     "frac_of_bf16_peak_div6": round(fl / us * 1e-6 / PEAK6, 3), "err_vs_fp64": err, "rms_err_vs_fp64": rms,
                     "fp32_kernel_rms_err": r32, "fp32_err_by_kparts_max_rms": split_err,
                     "fp32_kernel_us": round(t32, 2), "fp32_kernel_err": e32,
                     "fp32_kernel_frac": round(fl / t32 * 1e-6 / PEAK32, 3)})
    print(line, flush=True)
if out_path:
    with open(out_path, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
