"""Codec timings (base AutoEncoder, whole clips): decode [B,64,256] -> 524288 samples and encode,
B = 1 and 8, median over rounds; achieved fraction of the fp32 MFMA peak from the algorithmic
conv flops (decode 95.3 GFLOP, encode 45.2 GFLOP per clip).

    python scripts/time_codec.py [--rounds 20] [--batches 1,8]"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from after_amd import pipeline  # noqa: E402

PEAK = 157.3e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--batches", default="1,8")
    ap.add_argument("--config", default="baseAE")
    ap.add_argument("--only", default="", help="decode | encode")
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    model, dcfg, acfg = pipeline.build_models("base", a.config, dev, seed=0)
    ae = model.emb_model
    for B in [int(b) for b in a.batches.split(",")]:
        z = torch.randn(B, 64, 256, device=dev)
        x = 0.1 * torch.randn(B, 1, 524288, device=dev)
        res = {"workload": f"{a.config} whole clips, B={B}"}
        for name, fn, gflop in (("decode", lambda: ae.decode(z), 95.3), ("encode", lambda: ae.encode(x), 45.2)):
            if a.only and a.only != name:
                continue
            for _ in range(3):
                fn()
            ts = []
            for _ in range(a.rounds):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ms = statistics.median(ts)
            res[name + "_ms"] = round(ms, 3)
            res[name + "_ms_min"] = round(min(ts), 3)
            res[name + "_clips_per_s"] = round(B / (ms * 1e-3), 1)
            res[name + "_frac_of_fp32_mfma_peak"] = round(gflop * 1e9 * B / (ms * 1e-3) / PEAK, 3)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
