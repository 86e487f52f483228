"""Per-layer timing of the time-major conv path over the codec's layer shapes x tile configurations.

    python scripts/bench_conv.py [--batch 1] [--tiles 0,1,2,...] [--reps 30]

One JSON line per (layer, tile): conv-only time (trains of `reps` launches between two events), TFLOP/s,
fraction of the fp32 MFMA peak; plus the activate+halo pass of the same tensor (GB/s)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from after_amd import diag  # noqa: E402

PEAK = 157.3e12
# (name, Cin, Cout, T_in, k, dil, stride): the baseAE decoder / encoder at one whole clip
LAYERS = [
    ("dec0 k3 768@1024", 768, 768, 1024, 3, 1, 1), ("dec0 k3d9 768@1024", 768, 768, 1024, 3, 9, 1),
    ("dec0 k1 768@1024", 768, 768, 1024, 1, 1, 1),
    ("dec1 k3 384@4096", 384, 384, 4096, 3, 1, 1), ("dec1 k1 384@4096", 384, 384, 4096, 1, 1, 1),
    ("dec2 k3 384@8192", 384, 384, 8192, 3, 1, 1), ("dec2 k3d9 384@8192", 384, 384, 8192, 3, 9, 1),
    ("dec2 k1 384@8192", 384, 384, 8192, 1, 1, 1),
    ("dec3 k3 192@16384", 192, 192, 16384, 3, 1, 1), ("dec3 k1 192@16384", 192, 192, 16384, 1, 1, 1),
    ("dec4 k3 64@32768", 64, 64, 32768, 3, 1, 1), ("dec4 k1 64@32768", 64, 64, 32768, 1, 1, 1),
    ("up0 phase 768->768 @256", 768, 768, 256, 2, 1, 1), ("up1 phase 768->384 @1024", 768, 384, 1024, 2, 1, 1),
    ("up2 phase 384->384 @4096", 384, 384, 4096, 2, 1, 1), ("up3 phase 384->192 @8192", 384, 192, 8192, 2, 1, 1),
    ("up4 phase 192->64 @16384", 192, 64, 16384, 2, 1, 1),
    ("enc1 k3 128@16384", 128, 128, 16384, 3, 1, 1), ("enc2 k3 256@8192", 256, 256, 8192, 3, 1, 1),
    ("enc4 k3 512@1024", 512, 512, 1024, 3, 1, 1), ("enc5 k3 512@256", 512, 512, 256, 3, 1, 1),
    ("enc1 k1 128@16384", 128, 128, 16384, 1, 1, 1), ("enc2 k1 256@8192", 256, 256, 8192, 1, 1, 1),
    ("enc4 k1 512@1024", 512, 512, 1024, 1, 1, 1),
    ("down 256->512 f4 @4096", 256, 512, 4096, 8, 1, 4),
    ("enct k5 512@256", 512, 512, 256, 5, 1, 1), ("enct k5 256@256", 256, 256, 256, 5, 1, 1),
    ("enct k5 64@256", 64, 64, 256, 5, 1, 1), ("ecapa k3 512@128", 512, 512, 128, 3, 1, 1),
    ("ecapa k1 512@128", 512, 512, 128, 1, 1, 1), ("ecapa k3 64@128", 64, 64, 128, 3, 1, 1),
    ("ecapa k3 1024@128", 1024, 1024, 128, 3, 1, 1),
]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--tiles", default="0,1,2,3,4,5,6,9,15,16")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--layers", default="")
    ap.add_argument("--x6", action="store_true", help="also time the bf16-pipe form (conv_x6.hip) of the stride-1 layers of <= 3 taps, per x6 tile")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    tiles = [int(t) for t in a.tiles.split(",")]
    for name, cin, cout, T, k, dil, stride in LAYERS:
        if a.layers and not any(s in name for s in a.layers.split(",")):
            continue
        w = torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5
        b = torch.randn(cout, device=dev)
        lp = (k - 1) * dil // 2
        c = diag.ConvTm(w, b, a.batch, T, dil, stride, lp, (k - 1) * dil - lp + (stride - 1), act=1)
        c.run(None, None, 1)  # fill the haloed buffer once
        t_act = timeit(lambda: c.run(None, None, 1), a.reps)
        line = {"layer": name, "B": a.batch, "gflop": round(c.flops / 1e9, 2),
                "act_us": round(t_act * 1e6, 1),
                "act_GBs": round(2 * 4.0 * a.batch * T * cin / t_act / 1e9)}
        for mode, tag in ((2 | 4 | 8, "full"), (2, "bare")):
            best = None
            for t in tiles:
                diag.set_conv_tile(t)
                try:
                    c.run(None, None, mode)
                    torch.cuda.synchronize()
                except Exception:
                    continue
                dt = timeit(lambda: c.run(None, None, mode), a.reps)
                line[f"{tag}_t{t}_us"] = round(dt * 1e6, 1)
                if t and (best is None or dt < best[1]):
                    best = (t, dt)
            diag.set_conv_tile(0)
            if best:
                line[f"{tag}_best"] = best[0]
                line[f"{tag}_best_frac"] = round(c.flops / best[1] / PEAK, 3)
        if a.x6 and stride == 1 and k <= 3 and cout % 4 == 0:
            t_act6 = timeit(lambda: c.run(None, None, 1 | 16), a.reps)  # activate + halo into bf16 planes
            line["x6_act_us"] = round(t_act6 * 1e6, 1)
            for mode, tag in ((2 | 4 | 8 | 16, "x6_full"), (2 | 16, "x6_bare"), (2 | 4 | 16, "x6_stats"), (2 | 8 | 16, "x6_res")):
                for t in (1, 2, 3, 4, 5, 6, 7, 8):
                    diag.set_conv_x6_tile(t)
                    try:
                        dt = timeit(lambda: c.run(None, None, mode), a.reps)
                        line[f"{tag}_t{t}_us"] = round(dt * 1e6, 1)
                    except Exception:
                        pass
                diag.set_conv_x6_tile(0)
            v = [line[k2] for k2 in line if k2.startswith("x6_full_t")]
            if v:
                line["x6_full_TF"] = round(c.flops / (min(v) * 1e-6) / 1e12, 1)
        print(json.dumps(line), flush=True)
        c.close()


if __name__ == "__main__":
    main()
