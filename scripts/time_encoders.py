"""Conditioning encoders (base config): ECAPATDNN on [B,64,128], Encoder1D on [B,64,256]; median ms.

    python scripts/time_encoders.py [--rounds 20] [--batches 1,8]"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from after_amd import pipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--batches", default="1,8")
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    model, dcfg, acfg = pipeline.build_models("base", "baseAE", dev, seed=0)
    for B in [int(b) for b in a.batches.split(",")]:
        z = torch.randn(B, 64, 256, device=dev)
        zt = z[..., :128].contiguous()
        res = {"workload": f"base conditioning encoders, B={B}"}
        for name, fn in (("ecapa", lambda: model.encoder(zt)), ("encoder1d", lambda: model.encoder_time(z))):
            for _ in range(3):
                fn()
            ts = []
            for _ in range(a.rounds):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            res[name + "_ms"] = round(statistics.median(ts), 3)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
