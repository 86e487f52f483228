"""Soak of the persistent streaming sampler: many chunks on BASELINE config 5 (8 streams, 100 steps), checking after every
block that the handle is still on the persistent path (a barrier timeout or a failed placement census would have switched it
to launches and raised) and that the output stays finite.  python scripts/stream_soak.py [--chunks 400]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from after_amd import Streamer, pipeline

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=400)
ap.add_argument("--streams", type=int, default=8)
ap.add_argument("--offline", action="store_true", help="soak the opt-in persistent OFFLINE sampler instead (one clip, 50 steps)")
ap.add_argument("--clips", type=int, default=0,
                help="soak the clip-per-XCD persistent sampler instead: batches of this many clips (>= 5), several lengths / configs, every "
                     "repeat compared bit for bit with the first (its qkv tiles hand rows to each other behind sequence words)")
args = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
if args.clips:
    t0, n = time.time(), 0
    for cfg, T, steps in (("base", 256, 50), ("midi", 256, 10), ("base", 320, 6), ("base", 1024, 2), ("base", 48, 10)):
        model, dcfg, _ = pipeline.build_models(cfg, "baseAE", dev, seed=7)
        net = model.net
        g = torch.Generator().manual_seed(T)
        B = args.clips
        x0 = torch.randn(B, net.n_channels, T, generator=g).to(dev)
        cond = torch.randn(B, net.cond_dim, generator=g).to(dev)
        tc = torch.rand(B, net.tcond_dim, T, generator=g).to(dev)
        ref = net.cfg_sample(x0, cond, tc, steps, 2.0, 1.0, -4.0).clone()
        assert net.sample_path() == 2, (cfg, T)
        reps = max(1, args.chunks * 50 // (steps * max(1, T // 64)) // 4)
        for i in range(reps):
            z = net.cfg_sample(x0, cond, tc, steps, 2.0, 1.0, -4.0)
            if i % 20 == 19 or i == reps - 1:
                torch.cuda.synchronize()
                assert net.sample_path() == 2 and torch.equal(z, ref), (cfg, T, i)
        net.check()
        n += reps
        print(f"{cfg} T={T} x {B} clips, {steps} steps: {reps} repeats bit-identical")
    print(f"clip-per-XCD persistent sampler: {n} launches in {time.time() - t0:.1f} s")
    sys.exit(0)
if args.offline:
    model, dcfg, _ = pipeline.build_models("base", "baseAE", dev, seed=7)
    model.net.set_sample_persist(True)
    g = torch.Generator().manual_seed(1)
    x0, cond, tc = (torch.randn(1, 64, 256, generator=g).to(dev), torch.randn(1, 6, generator=g).to(dev),
                    torch.randn(1, 12, 256, generator=g).to(dev))
    ref = model.net.cfg_sample(x0, cond, tc, 50, 2.0, 1.0, -4.0).clone()
    t0 = time.time()
    for i in range(args.chunks):
        z = model.net.cfg_sample(x0, cond, tc, 50, 2.0, 1.0, -4.0)
        if i % 50 == 49:
            torch.cuda.synchronize()
            assert model.net.sample_persist() and torch.equal(z, ref), i
    torch.cuda.synchronize()
    print(f"offline persistent sampler: {args.chunks} clips, bit-identical every time, {(time.time() - t0) / args.chunks * 1e3:.2f} ms per clip")
    sys.exit(0)
model, dcfg, acfg = pipeline.build_models("cycle", "baseAE_causal", dev, seed=7)
st = Streamer(model, model.emb_model, chunk_size=4, n_signal_timbre=128, max_batch=args.streams, max_nb_steps=100,
              share_first_stream=False)
st.set_nb_steps(100)
x = 0.1 * torch.randn(args.streams, 2, 4 * st.ae_ratio, device=dev)
t0 = time.time()
for i in range(args.chunks):
    y = st(x)
    if i % 50 == 49:
        torch.cuda.synchronize()
        assert model.net.stream_persist(), f"chunk {i}: the sampler left the persistent path"
        assert torch.isfinite(y).all(), i
torch.cuda.synchronize()
print(f"persistent streaming sampler: {args.chunks} chunks x {args.streams} streams, still persistent, "
      f"{(time.time() - t0) / args.chunks * 1e3:.2f} ms per chunk")
