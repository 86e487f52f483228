"""Per-phase timeline of the persistent streaming step (stream_step_kernel, denoiser.hip) on BASELINE config 5:
AFTER_STEP_TRACE=1 makes every workgroup stamp the 100 MHz wall clock when it arrives at / leaves each XCD-local
barrier.  Prints, per phase and for the workgroups of ONE XCD (--xcd): the slowest workgroup's work time, the median, and
the barrier cost (first exit minus last arrival); then the step time of every XCD.  python scripts/stream_step_trace.py [--streams 8]"""
import argparse
import ctypes
import os
import sys

os.environ["AFTER_STEP_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from after_amd import Streamer, _lib, pipeline

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=8)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--xcd", type=int, default=0)
ap.add_argument("--detail", action="store_true", help="per phase: the five slowest workgroups (index in the XCD: work us)")
ap.add_argument("--offline", action="store_true", help="the persistent OFFLINE sampler (one clip, base, 50 steps) instead")
ap.add_argument("--clips", type=int, default=1, help="with --offline: clips of the call (1 - 2: the one-clip kernel, the pair in one launch; >= 3: the clip-per-XCD kernel)")
args = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
if args.offline:
    os.environ.setdefault("AFTER_SAMPLE_PERSIST", "1")
    model, dcfg, acfg = pipeline.build_models("base", "baseAE", dev, seed=7)
    TT = int(os.environ.get("AFTER_T", "256"))
    NB = args.clips
    x0, cond, tc = torch.randn(NB, 64, TT, device=dev), torch.randn(NB, 6, device=dev), torch.randn(NB, 12, TT, device=dev)
    model.net.set_sample_persist(True)
    for _ in range(3):
        model.net.cfg_sample(x0, cond, tc, 50, 2.0, 1.0, -4.0)
    torch.cuda.synchronize()
    assert model.net.sample_persist()
    kernel = model.net.sample_path()  # 1: the one-clip kernel (one or two clips per launch), 2: the clip-per-XCD kernel
else:
    model, dcfg, acfg = pipeline.build_models("cycle", "baseAE_causal", dev, seed=7)
if not args.offline:
  st = Streamer(model, model.emb_model, chunk_size=4, n_signal_timbre=128, max_batch=args.streams, max_nb_steps=args.steps,
              share_first_stream=False)
  st.set_nb_steps(args.steps)
  x = 0.1 * torch.randn(args.streams, 2, 4 * st.ae_ratio, device=dev)
  for _ in range(3):
    st(x)
  torch.cuda.synchronize()
h = model.net._handle
n = torch.cuda.get_device_properties(0).multi_processor_count
buf = np.zeros((n, 128), dtype=np.uint64)
_lib.check(_lib.lib().after_denoiser_step_trace(h, buf.ctypes.data_as(ctypes.c_void_p), n), "step_trace")
L = dcfg["net"]["n_layers"]
names = ["patchify"] + sum(([f"L{l} ln", f"L{l} qkv", f"L{l} attn", f"L{l} up+roll", f"L{l} down"] for l in range(L)), []) + ["tail"]
xcc = buf[:, 127].astype(int)
t_all = buf[:, :2 * len(names)].astype(np.int64)
t = t_all[xcc == args.xcd]
t0 = t[:, 0].min()
tot_work = tot_bar = 0.0
print(f"workgroups {n}, XCC populations {np.bincount(buf[:, 127].astype(int), minlength=8).tolist()}")
print(f"{'phase':>12} {'work max':>9} {'work med':>9} {'barrier':>8}   (us)")
for p, name in enumerate(names):
    start = t[:, 2 * p]                      # leaves the previous barrier (or kernel start)
    end = t[:, 2 * p + 1]                    # arrives at the next barrier (or kernel end)
    work = (end - start) / 100.0
    line = f"{name:>12} {work.max():9.2f} {np.median(work):9.2f}"
    tot_work += (end.max() - start.min()) / 100.0
    if p + 1 < len(names):
        bar = (t[:, 2 * p + 2].min() - end.max()) / 100.0
        tot_bar += (t[:, 2 * p + 2].max() - end.max()) / 100.0
        line += f" {bar:8.2f}"
    if args.detail:
        o = np.argsort(-work)[:5]
        line += "   slowest: " + " ".join(f"{i}:{work[i]:.2f}(+{(start[i] - start.min()) / 100.0:.2f})" for i in o)
    print(line)
print("step per XCD (us): " + ", ".join(f"{(t_all[xcc == x][:, 2 * len(names) - 1].max() - t_all[xcc == x][:, 0].min()) / 100.0:.1f}"
                                         for x in range(8) if (xcc == x).any()))
if args.offline:  # per XCD: the step's time by kind of phase (sum over the layers; first start -> last arrival of each phase)
    kinds = ["ln", "qkv", "attn", "up+roll", "down"]
    for x in range(8):
        tx = t_all[xcc == x]
        if not len(tx) or tx[:, 1].max() == 0:
            continue
        tot = {k: 0.0 for k in kinds}
        for p_, nm in enumerate(names):
            for k in kinds:
                if nm.endswith(" " + k):
                    tot[k] += (tx[:, 2 * p_ + 1].max() - tx[:, 2 * p_].min()) / 100.0
        print(f"XCD {x}: " + "  ".join(f"{k} {v:.1f}" for k, v in tot.items()))
print(f"XCD {args.xcd} step: {(t[:, 2 * len(names) - 1].max() - t0) / 100.0:.1f} us; phases (first start -> last arrival) {tot_work:.1f} us; "
      f"barriers (last arrival -> last exit) {tot_bar:.1f} us")


if not args.offline:  # the last layer's attention phase: stamps of thread 0 inside the item (workgroups with an item)
    pa = 3 + 5 * (L - 1)
    ga = buf[:, 80:85].astype(np.int64)
    has = ga[:, 0] > 0
    if has.any():
        rela = (ga[has] - t_all[has][:, 2 * pa][:, None]) / 100.0
        arra = (t_all[has][:, 2 * pa + 1] - t_all[has][:, 2 * pa]) / 100.0
        print("attention phase, %d workgroups with an item, median / max us after the barrier: entry %.2f / %.2f, first pass landed %.2f / %.2f, keys done %.2f / %.2f, "
              "rows exchanged %.2f / %.2f, LayerNorm tail stored %.2f / %.2f, arrival %.2f / %.2f" % ((int(has.sum()),) + tuple(
                  v for k in range(5) for v in (np.median(rela[:, k]), rela[:, k].max())) + (np.median(arra), arra.max())))

if args.offline and kernel == 1:  # effective shader clock over the step; inside the last layer's qkv phase (workgroup-local stamps of wave 0)
    cyc = (buf[:, 71].astype(np.int64) - buf[:, 70].astype(np.int64))
    wall = (t_all[:, 2 * len(names) - 1] - t_all[:, 0]) / 100.0
    print("effective shader clock over the step: %.0f MHz (median over workgroups)" % np.median(cyc / wall))
    ph = 2 + 5 * (L - 1)
    gq = buf[:, 64:68].astype(np.int64)
    rel = (gq - t_all[:, 2 * ph][:, None]) / 100.0
    arr = (t_all[:, 2 * ph + 1] - t_all[:, 2 * ph]) / 100.0
    print("qkv phase, median us after the barrier: start %.2f, MFMAs issued %.2f, partials exchanged %.2f, stores issued %.2f, "
          "arrival %.2f" % tuple(np.median(rel, 0).tolist() + [np.median(arr)]))
    pa = 3 + 5 * (L - 1)  # the last layer's attention phase: stamps of thread 0 inside the item (workgroups with an item)
    ga = buf[:, 80:85].astype(np.int64)
    has = ga[:, 0] > 0
    if has.any():
        rela = (ga[has] - t_all[has][:, 2 * pa][:, None]) / 100.0
        arra = (t_all[has][:, 2 * pa + 1] - t_all[has][:, 2 * pa]) / 100.0
        print("attention phase, median / max us after the barrier: entry %.2f / %.2f, first pass landed %.2f / %.2f, keys done %.2f / %.2f, "
              "rows exchanged %.2f / %.2f, LayerNorm tail stored %.2f / %.2f, arrival %.2f / %.2f" % (tuple(
                  v for k in range(5) for v in (np.median(rela[:, k]), rela[:, k].max())) + (np.median(arra), arra.max())))
    gw = buf[:, 72:80].astype(np.int64)
    print("qkv phase, end of each wave's MFMAs (median us after the barrier): " + " ".join("%.2f" % v for v in np.median((gw - t_all[:, 2 * ph][:, None]) / 100.0, 0)))

if args.offline and kernel == 2:  # the clip-per-XCD kernel: effective shader clock, anatomy of the last layer's qkv phase (first tile of a workgroup)
    cyc = (buf[:, 71].astype(np.int64) - buf[:, 70].astype(np.int64))
    wall = (t_all[:, 2 * len(names) - 1] - t_all[:, 0]) / 100.0
    okw = wall > 0
    print("effective shader clock over the step: %.0f MHz (median over workgroups)" % np.median(cyc[okw] / wall[okw]))
    ph = 2 + 5 * (L - 1)
    gq = buf[okw][:, 64:68].astype(np.int64)
    rel = (gq - t_all[okw][:, 2 * ph][:, None]) / 100.0
    arr = (t_all[okw][:, 2 * ph + 1] - t_all[okw][:, 2 * ph]) / 100.0
    lc = (buf[okw][:, 69].astype(np.int64) - buf[okw][:, 68].astype(np.int64))
    m = np.median(rel, 0).tolist()
    slab0 = "%.2f" % m[1] if (gq[:, 1] > 0).all() else "n/a (not stamped by this kernel form)"
    print("qkv phase, first tile, median us after the barrier: entry %.2f, slab 0 published %s, K loop done %.2f, epilogue issued %.2f; "
          "phase %.2f; shader cycles entry -> loop done %.0f (%.0f MHz)" % (m[0], slab0, m[2], m[3], np.median(arr), np.median(lc),
                                                                          np.median(lc / ((gq[:, 2] - gq[:, 0]) / 100.0))))
    if os.environ.get("AFTER_CLIP_GSTAG"):  # experiment: the stamps are MLP-up's; per row tile (rank & 3 -- the census rank is not in the
        pu = 4 + 5 * (L - 1)                # stamps: group by the delay seen = entry - barrier exit)
        x = xcc == args.xcd
        g4 = buf[x][:, 64:68].astype(np.int64)
        ex, ar = t_all[x][:, 2 * pu], t_all[x][:, 2 * pu + 1]
        delay = (g4[:, 0] - ex) / 100.0
        order = np.argsort(delay)
        for q in range(4):
            sel = order[8 * q:8 * q + 8]
            print("MLP-up, delay group %d (entry %.1f us after the barrier): K loop %.2f, epilogue issue %.2f, drain to arrival %.2f, entry -> arrival %.2f us (medians of 8 workgroups)"
                  % (q, np.median(delay[sel]), np.median((g4[sel, 2] - g4[sel, 0]) / 100.0), np.median((g4[sel, 3] - g4[sel, 2]) / 100.0),
                     np.median((ar[sel] - g4[sel, 3]) / 100.0), np.median((ar[sel] - g4[sel, 0]) / 100.0)))
    pa = 3 + 5 * (L - 1)  # the last layer's attention phase: thread 0's stamps inside the workgroup's items (pairs of chunks)
    ga = buf[okw][:, 80:88].astype(np.int64)
    if (ga[:, 0] > 0).all():
        rela = np.median((ga - t_all[okw][:, 2 * pa][:, None]) / 100.0, 0)
        print("attention phase, median us after the barrier: item 0 entry %.2f, operands landed %.2f, keys done %.2f, rows exchanged %.2f, "
              "LayerNorm tail issued %.2f; item 1 entry %.2f; item 2 entry %.2f, end %.2f; arrival %.2f"
              % (tuple(rela.tolist()) + (np.median((t_all[okw][:, 2 * pa + 1] - t_all[okw][:, 2 * pa]) / 100.0),)))
    pr = buf[okw][:, 88:120].astype(np.int64).reshape(-1, 8, 4)
    if pr.any():  # -DX6R_PROF=1 builds: per-wave cycle counters of the traced GEMM phase's K loop
        med = np.median(pr, 0)
        print("traced GEMM phase, per wave, median shader cycles [fragment wait, DMA wait, barrier, MFMA stream]:")
        for w in range(8):
            print("  wave %d: %s  (sum %d)" % (w, " ".join("%7d" % v for v in med[w]), med[w].sum()))
