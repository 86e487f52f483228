"""Per-chunk latency of the streaming path (BASELINE config 5: base + cycle config, 100-step
cached sampler, causal cached-conv codec, batch 8) on one MI355X.

    python scripts/time_stream.py [--config cycle] [--batch 8] [--steps 100] [--chunks 20]

Prints one JSON line: ms per chunk (all `batch` streams together), x real-time per stream and
aggregate, and the split structure / timbre / diffuse / decode."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from after_amd import Streamer, pipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cycle")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--chunks", type=int, default=20)
    ap.add_argument("--chunk-size", type=int, default=4)
    ap.add_argument("--shared", action="store_true", help="export.py behaviour: one diffusion, repeated")
    ap.add_argument("--no-parts", action="store_true", help="whole chunks only (for a kernel trace of one chunk)")
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    dev = "cuda:0"
    model, dcfg, acfg = pipeline.build_models(a.config, "baseAE_causal", dev)
    st = Streamer(model, model.emb_model, chunk_size=a.chunk_size, n_signal_timbre=128,
                  max_batch=a.batch, max_nb_steps=a.steps, share_first_stream=a.shared)
    st.set_nb_steps(a.steps)
    st.set_guidance_timbre(2.0)
    n = a.chunk_size * st.ae_ratio
    x = 0.1 * torch.randn(a.batch, 2, n, device=dev)
    for _ in range(3):
        st(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.chunks):
        st(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.chunks * 1e3
    parts = {}
    if a.no_parts:
        print(json.dumps({"ms_per_chunk": round(ms, 3)}))
        return
    xs, xt = x[:, :1].contiguous(), x[:, 1:].contiguous()
    cond = torch.cat((st.structure(xs), st.timbre(xt)), 1)
    z = st.diffuse(cond)
    for name, fn in (("structure", lambda: st.structure(xs)), ("timbre", lambda: st.timbre(xt)),
                     ("diffuse", lambda: st.diffuse(cond)), ("decode", lambda: st.decode(z))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        parts[name + "_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    audio_s = n / st.sr
    print(json.dumps({"workload": f"{a.config} streaming, {a.steps}-step cached sampler, batch {a.batch}, "
                                  f"chunk {a.chunk_size} frames ({audio_s * 1e3:.1f} ms audio)",
                      "ms_per_chunk": round(ms, 3), "xrt_per_stream": round(audio_s / (ms / 1e3), 2),
                      "xrt_aggregate": round(a.batch * audio_s / (ms / 1e3), 2),
                      "independent_streams": not a.shared, **parts}))


if __name__ == "__main__":
    main()
