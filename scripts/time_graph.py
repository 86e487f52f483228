"""hipGraph replay vs plain launches for the sampler's step loop (north_star: "the step loop
captured as a hipGraph").  Times after_sample (50 Euler steps, base config) both ways, B = 1 and 8,
interleaved rounds in ONE process (median and min over rounds), checks bit-equality, and prints
one JSON line per batch size -> profiles/r2_graph_vs_eager.json.

    python scripts/time_graph.py [--rounds 7]"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from after_amd import DenoiserV2, RectifiedFlow, _lib, configs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    dcfg = configs.diffusion_config("base")
    for B in (1, 8):
        net = DenoiserV2(**dcfg["net"])
        model = RectifiedFlow(net=net, sr=44100, device=dev)
        T = 256
        x0 = torch.randn(B, 64, T, device=dev)
        cond = torch.randn(B, 6, device=dev)
        tc = torch.randn(B, 12, T, device=dev)
        out_e, out_g = torch.empty_like(x0), torch.empty_like(x0)
        net.reserve(3 * B, T, a.steps)

        def run(graph, out):
            _lib.check(_lib.lib().after_denoiser_set_graph(net._handle, int(graph)), "set_graph")
            net.cfg_sample(x0, cond, tc, a.steps, 2.0, 1.0, -4.0, out=out)

        for g in (0, 1, 0, 1):  # warm-up both paths (the first graph call captures + instantiates)
            run(g, out_g if g else out_e)
        torch.cuda.synchronize()
        te, tg = [], []
        for _ in range(a.rounds):
            for g, out, ts in ((0, out_e, te), (1, out_g, tg)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(g, out)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
        print(json.dumps({"workload": f"after_sample base, {a.steps} steps, B={B}, T=256", "rounds": a.rounds,
                          "eager_ms_median": round(statistics.median(te), 3), "eager_ms_min": round(min(te), 3),
                          "graph_ms_median": round(statistics.median(tg), 3), "graph_ms_min": round(min(tg), 3),
                          "graph_over_eager": round(statistics.median(tg) / statistics.median(te), 4),
                          "bit_equal": bool(torch.equal(out_e, out_g))}), flush=True)


if __name__ == "__main__":
    main()
