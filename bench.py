#!/usr/bin/env python3
"""Benchmark of the AFTER latent-diffusion sampling path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic clips per GPU:
conditioning encoders -> 50-step CFG rectified-flow sampler -> AutoEncoder.decode
(BASELINE.json configs[1]: base audio-to-audio, 50 steps, batch 1 per GPU, random
init, synthetic latents; one clip = 524288 samples = 11.889 s @ 44.1 kHz).
Other BASELINE configs: `--batch-per-gpu 8` (configs[2]'s per-GPU shard),
`--config midi --batch-per-gpu 8` (configs[3]), `--stream` (configs[4]: base + cycle,
100-step cached sampler, causal cached-conv codec, 8 independent streams; one step =
one 4-frame chunk of every stream), `--from-audio [--config tiny]` (configs[0]'s chain:
two audio clips -> AutoEncoder.encode x 2 -> encoders -> sampler -> decode).
Prints ONE JSON line on rank 0.

`--pmc` (single GPU): re-measures the HBM traffic of the dominant GEMM with two
rocprofv3 counter passes of this same command and writes profiles/<round>_pmc_hbm_<cfg>.json
together with the hash of the kernel sources; the default run reports `roofline.traffic`
from that file only while the hash still matches."""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CLIP_SAMPLES = 524288
CLIP_SECONDS = CLIP_SAMPLES / 44100.0
T_FRAMES = 256
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 MFMA peak; gemm_x6 spends six bf16 MFMAs per fp32 product block
PEAK_X6_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0  # three bf16 planes per operand: six MFMAs per fp32 product block (gemm_x6.hip)
PEAK_H3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0  # two fp16 pieces per operand: three MFMAs per block (gemm_h3_pipe.h; f16 and bf16 MFMAs issue alike)
ARITH = {0: "fp32 MFMA chain (v_mfma_f32_16x16x4_f32)",
         1: "three bf16 planes per operand, six exact v_mfma_f32_16x16x32_bf16 per fp32 product block, fp32 accumulate",
         2: "two fp16 pieces per operand (22-bit significand, exact power-of-two scales from guaranteed bounds), three exact "
            "v_mfma_f32_16x16x32_f16 per product block, fp32 accumulate",
         3: "bf16 operands, one MFMA per block (opt-in tolerance tier)"}


def split_ceiling(arith):
    """Matrix-pipe ceiling of the Linears' arithmetic in fp32-equivalent TFLOP/s: the dense 16-bit MFMA peak / MFMAs per product block."""
    return {0: PEAK_FP32_MFMA_TFLOPS, 1: PEAK_X6_TFLOPS, 2: PEAK_H3_TFLOPS, 3: PEAK_BF16_MFMA_TFLOPS}[arith]
ROUND = "r6"
# sources whose change invalidates a committed traffic measurement of the dominant GEMM
TRAFFIC_SOURCES = ["after_amd/csrc/gemm.hip", "after_amd/csrc/gemm_x6.hip", "after_amd/csrc/gemm_pipe.h",
                   "after_amd/csrc/gemm_x6_pipe.h", "after_amd/csrc/gemm_h3_pipe.h", "after_amd/csrc/denoiser.hip"]


def source_hash():
    h = hashlib.sha256()
    for rel in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_path(config, batch):
    return os.path.join(ROOT, "profiles", f"{ROUND}_pmc_hbm_{config}_b{batch}.json")


def piano_roll(n, tcond_dim, dev, seed=77):
    """Synthetic piano roll: zeros with 4 notes per clip, velocities U(0.3, 1)."""
    g = torch.Generator().manual_seed(seed)
    roll = torch.zeros(n, tcond_dim, T_FRAMES)
    for b in range(n):
        for _ in range(4):
            p = int(torch.randint(20, min(110, tcond_dim), (1, ), generator=g))
            a = int(torch.randint(0, T_FRAMES - 64, (1, ), generator=g))
            roll[b, p, a:a + 64] = 0.3 + 0.7 * float(torch.rand((), generator=g))
    return roll.to(dev)


def cpu_baseline(nb_steps, state_dicts, dcfg, acfg, tcond=None):
    """The CPU oracle (port of the reference's algorithm, oracle/) on this host's cores,
    on ONE clip of the same workload (full path), bounded to a few seconds of CPU."""
    import oracle
    # torch's intra-op pool stops scaling (and the many small ops of a B=1 step start
    # to thrash) well before the host's full thread count: cap at 16
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1234)
    zs = torch.randn(1, 64, T_FRAMES, generator=g)
    zt = torch.randn(1, 64, T_FRAMES, generator=g)
    x0 = torch.randn(1, 64, T_FRAMES, generator=g)
    sd_net, sd_enc, sd_et, sd_ae = state_dicts
    with torch.no_grad():
        t0 = time.perf_counter()
        cond = oracle.ecapa_forward(sd_enc, zt[..., :128], dcfg["encoder"])
        tc = tcond if tcond is not None else oracle.encoder1d_forward(sd_et, zs, dcfg["encoder_time"])
        z = oracle.sample(sd_net, dcfg["net"], x0, cond, tc, nb_steps, 2.0, 1.0)
        y = oracle.ae_decode(sd_ae, z, acfg)
        dt = time.perf_counter() - t0
        t8 = None
        if cores > 8:  # comparability with BASELINE.md section 2 (the reference's own numbers: 8 cores)
            torch.set_num_threads(8)
            t1 = time.perf_counter()
            cond = oracle.ecapa_forward(sd_enc, zt[..., :128], dcfg["encoder"])
            tc = tcond if tcond is not None else oracle.encoder1d_forward(sd_et, zs, dcfg["encoder_time"])
            oracle.ae_decode(sd_ae, oracle.sample(sd_net, dcfg["net"], x0, cond, tc, nb_steps, 2.0, 1.0), acfg)
            t8 = time.perf_counter() - t1
            torch.set_num_threads(cores)
    assert y.shape[-1] == CLIP_SAMPLES
    return {"value": CLIP_SECONDS / dt, "unit": "audio_s_per_wall_s", "cores": cores, "kind": "port",
            "sample": f"1 clip, full path (encoders + {nb_steps}-step sampler + decode), "
                      f"{dt:.2f} s on {cores} threads, torch {torch.__version__} CPU fp32",
            "threads8": ({"value": round(CLIP_SECONDS / t8, 3), "cores": 8,
                          "sample": f"the same clip on 8 threads, {t8:.2f} s (BASELINE.md section 2: the reference's "
                                    "own path on 8 cores of the survey container, 3.6x RT)"} if t8 else None)}


def cpu_baseline_stream(model, dcfg, acfg, chunk, nb_steps, nsig):
    """oracle.stream_forward on a bounded sample: 1 stream, 2 chunks, the same step count."""
    import oracle
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    pick = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    sd_ae = {k: v.detach().cpu() for k, v in model.emb_model.state_dict().items()}
    n, n_chunks = 1, 2
    g = torch.Generator().manual_seed(5)
    L = n_chunks * chunk * 2048
    xs, xt = 0.1 * torch.randn(n, 1, L, generator=g), 0.1 * torch.randn(n, 1, L, generator=g)
    noise = torch.randn(n, 64, n_chunks * chunk, generator=g)
    with torch.no_grad():
        t0 = time.perf_counter()
        oracle.stream_forward(pick("net."), pick("encoder."), pick("encoder_time."), sd_ae, dcfg, acfg, xs, xt,
                              noise, chunk, nb_steps, 2.0, 1.0, nsig)
        dt = time.perf_counter() - t0
    return {"value": n * L / 44100.0 / dt, "unit": "audio_s_per_wall_s", "cores": cores, "kind": "port",
            "sample": f"{n} stream x {n_chunks} chunks of {chunk} frames, {nb_steps}-step cached sampler + causal "
                      f"codec (oracle.stream_forward), {dt:.2f} s on {cores} threads"}


def _trains(dev, M, shapes, x6):
    """The dominant launches without per-launch event bracketing: trains of 100 back-to-back launches of the
    given (N, K) shapes, one event pair per train -> TFLOP/s."""
    from after_amd import diag
    tot_t, tot_f = 0.0, 0.0
    for (n_, k_) in shapes:
        a_ = torch.randn(M, k_, device=dev)
        w_ = torch.randn(n_, k_, device=dev)
        o_ = torch.empty(M, n_, device=dev)
        if x6:
            a3, w3 = diag.split_x6(a_), diag.split_x6(w_)
            fn = lambda: diag.gemm_x6(a3, w3, out=o_)
        else:
            fn = lambda: diag.gemm(a_, w_, out=o_)
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        tot_t += e0.elapsed_time(e1) * 1e-3 / 100
        tot_f += 2.0 * M * n_ * k_
    return round(tot_f / tot_t / 1e12, 2)


def gemm_roofline(model, run_once, dev, dcfg, B, config, args):
    """Roofline of the dominant kernel -- the GEMM behind the qkv / MLP Linears -- measured live with HIP events on
    the launch stream in extra, untimed passes.  On the default path that kernel is gemm_x6 (fp32 products as six
    exact bf16 MFMAs: priced against the dense bf16 MFMA peak / 6); `fp32_mfma_path` carries the same measurement
    with every Linear forced onto the fp32 MFMA kernel (the rounds-1/2 path, priced against the fp32 MFMA peak)."""
    # the dominant launches: qkv / MLP-up / MLP-down (1.208 GFLOP each for base at B=1); the patchify /
    # AdaLN / out_proj launches share the fp32 kernel template but are a tenth of the size
    E_, ME_ = dcfg["net"]["embed_dim"], dcfg["net"]["embed_dim"] * dcfg["net"]["mlp_multiplier"]
    big = 0.5 * 2.0 * (3 * B * T_FRAMES) * E_ * ME_

    def timed_pass(kernel, min_flops):
        model.net.profile(True, min_flops=min_flops, kernel=kernel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_once()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ms, launches, flops, nbytes = model.net.gemm_time(with_bytes=True)
        model.net.profile(False)
        return ms, launches, flops, nbytes, wall

    if args.stream:
        if model.net.stream_persist():
            # the dominant kernel is the persistent sampler: ONE launch per chunk runs all Euler steps
            ms, launches, flops, nbytes, _ = timed_pass(3, 0.0)
            if not launches:
                return None
            steps = args.nb_steps or 100
            gbs = nbytes / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": "stream_step_kernel (persistent streaming sampler: eight XCD-local pipelines, "
                                              "fp32 MFMA GEMMs on <= 32 token rows per XCD: weight streaming)",
                    "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                    "traffic": None, "launches": int(launches), "avg_launch_us": round(ms * 1e3 / launches, 1),
                    "us_per_euler_step": round(ms * 1e3 / launches / steps, 2),
                    "bytes_per_launch": round(nbytes / launches),
                    "fabric_bytes_per_launch_by_construction": round(8 * nbytes / launches),
                    "achieved_fabric_GBps": round(8 * gbs, 1),
                    "gemm_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
                    "note": "algorithmic bytes = every Linear weight once per Euler step (54 MB; activations are KBs) / "
                            "HIP-event duration of the launch.  Each of the eight XCDs streams all weights (no cross-XCD "
                            "traffic inside the kernel): 8 x that crosses the fabric by construction, served by the "
                            "memory-side cache, not HBM.  The step is latency-bound: 32 phases behind XCD-local barriers "
                            "(profiles/r3_stream_step_trace.txt), ~54 us of its ~160 us is the weight stream"}
        ms, launches, flops, nbytes, _ = timed_pass(0, 0.0)
        if not launches:
            return None
        # <= 96 tokens per launch: weight-streaming GEMMs (3 MB of W against 0.2 MB of activations),
        # priced against HBM bandwidth; in practice they sit at the launch / dependency floor
        gbs = nbytes / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "gemm_f32_skinny_kernel / gemm_f32_bal_kernel on <= 96 tokens "
                                          "(weight streaming; launch-per-kernel path of the streaming sampler)",
                "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                "traffic": None, "launches": int(launches), "avg_launch_us": round(ms * 1e3 / launches, 2),
                "bytes_per_launch": round(nbytes / launches),
                "note": "algorithmic bytes (A + W + C) / HIP-event launch duration; the launches are 6-9 us each, "
                        "i.e. latency-bound, and the weights are Infinity-Cache resident after the first step"}

    seg = None
    if model.net.sample_persist():
        seg = persist_roofline(model, run_once, dcfg, B, config, args.nb_steps or 50)
    prof = None
    pth = pmc_path(config, B)
    traffic_note = "no committed PMC profile for this configuration (python bench.py --pmc)"
    if os.path.exists(pth):
        prof = json.load(open(pth))
        if prof.get("source_hash") != source_hash():
            prof, traffic_note = None, f"{os.path.basename(pth)} is stale (kernel sources changed since it was measured)"
        else:
            traffic_note = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command ({os.path.basename(pth)}, "
                            f"sources {prof.get('source_hash')}): (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, "
                            "Infinity-Cache hits included")
    M = 3 * B * T_FRAMES
    note = ("achieved = algorithmic fp32 flops / per-launch HIP-event bracketing inside the sampler (launch latency "
            "included: the rocprofv3 kernel durations in profiles/ are ~2 us shorter per launch); "
            "achieved_back_to_back = the same kernel in trains of 100 launches of the qkv / MLP-up and MLP-down shapes")

    def fp32_leg(ms, launches, flops, extra=None):
        ach = flops / (ms * 1e-3) / 1e12
        d = {"kernel": "gemm_f32_bal_kernel (v_mfma_f32_16x16x4_f32, exact fp32 fma chain)",
             "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
             "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "launches": int(launches),
             "avg_launch_us": round(ms * 1e3 / launches, 2), "flops_per_launch": round(flops / launches)}
        d.update(extra or {})
        return d

    if seg is not None:
        note = "the launch-per-kernel path of the same sampler (AFTER_SAMPLE_PERSIST=0), one extra untimed pass: " + note
    mode, min_rows = model.net.gemm_path()
    x_ms, x_n, x_fl, _, _ = timed_pass(1, big) if mode != 0 else (0.0, 0, 0.0, 0.0, 0.0)
    f_ms, f_n, f_fl, _, _ = timed_pass(2, big)  # big launches the default path leaves on the fp32 MFMA kernel
    if not x_n:  # fp32 MFMA everywhere (AFTER_GEMM_X6=0, or too few token rows)
        if not f_n:
            return None
        r = fp32_leg(f_ms, f_n, f_fl, {"achieved_back_to_back": _trains(dev, M, ((ME_, E_), (E_, ME_)), False)})
        r.update({"bound": "mfma", "traffic": prof.get("gemm_big_mean_bytes_per_launch") if prof else None,
                  "traffic_unit": "bytes per launch", "traffic_note": traffic_note, "note": note,
                  "kernel": r["kernel"] + f": the qkv / MLP launches (>= {big / 1e9:.2f} GFLOP each)"})
        return r
    ach = x_fl / (x_ms * 1e-3) / 1e12
    roof = {"bound": "mfma",
            "kernel": "gemm_x6_kernel (fp32 product blocks as 6 x v_mfma_f32_16x16x32_bf16 on exact three-way bf16 "
                      f"splits, fp32 accumulate): the qkv / MLP launches (>= {big / 1e9:.2f} GFLOP each) it runs",
            "achieved": round(ach, 2), "peak": round(PEAK_X6_TFLOPS, 1), "unit": "TFLOP/s",
            "frac": round(ach / PEAK_X6_TFLOPS, 4),
            "peak_note": "dense bf16 MFMA peak 2500 TFLOP/s / 6 MFMAs per fp32 product block; against the fp32 MFMA peak "
                         f"(157.3) the same launches are at {ach / PEAK_FP32_MFMA_TFLOPS:.3f}",
            "traffic": prof.get("gemm_x6_mean_bytes_per_launch") if prof else None, "traffic_unit": "bytes per launch",
            "traffic_note": traffic_note, "launches": int(x_n), "avg_launch_us": round(x_ms * 1e3 / x_n, 2),
            "flops_per_launch": round(x_fl / x_n),
            "achieved_back_to_back": _trains(dev, M, ((ME_, E_), (E_, ME_)) if f_n == 0 else ((ME_, E_), ), True),
            "gemm_path": {"mode": mode, "min_rows": min_rows},
            "note": note}
    if f_n:
        roof["same_run_fp32_kernel_launches"] = fp32_leg(
            f_ms, f_n, f_fl, {"what": "launches of the same size class that stay on the fp32 MFMA kernel: at one clip "
                                      "MLP-down (N = 512, K = 1536: per-shape dispatch, DESIGN.md section 4), at larger "
                                      "batches only the once-per-sample AdaLN projection of all steps"})
    # continuity with rounds 1-2: the same step with EVERY Linear on the fp32 MFMA kernel
    model.net.set_gemm_path(0)
    run_once()
    a_ms, a_n, a_fl, _, wall = timed_pass(2, big)
    model.net.set_gemm_path(mode, min_rows)
    if a_n:
        roof["fp32_mfma_path"] = fp32_leg(a_ms, a_n, a_fl, {
            "ms_per_step": round(wall * 1e3, 3), "achieved_back_to_back": _trains(dev, M, ((ME_, E_), (E_, ME_)), False),
            "what": "one extra untimed step with after_denoiser_set_gemm_path(0): every Linear on the fp32 MFMA kernel "
                    "(the rounds-1/2 product path; ms_per_step here includes the event bracketing)"})
    mf = mfma_profile(config, B)
    if seg is not None:  # the dominant kernel of this leg is the persistent one; the launch path rides along for continuity
        if prof and prof.get("seg_bytes_per_launch"):
            nb = args.nb_steps or 50
            seg["traffic"] = prof["seg_bytes_per_launch"]
            seg["traffic_unit"] = "bytes per launch (one launch = all Euler steps of all clips)"
            seg["traffic_note"] = traffic_note
            # algorithmic bytes of one Euler step as SURVEY.md 8(d) counts them: the denoiser's weights ONCE in fp32 (all CFG rows
            # and clips share them) + every tensor that crosses a phase boundary of a layer written once and read once in fp32
            # (3 x B x 256 rows x 512 x 4 B = 1.6 MB per E-wide tensor and clip): norm1 output (E), qkv (3E), the residual stream
            # after attention (E), norm3 output (E), the MLP hidden layer (ME), the residual stream after the MLP (E)
            E2, L2 = dcfg["net"]["embed_dim"], dcfg["net"]["n_layers"]
            ME2 = E2 * dcfg["net"]["mlp_multiplier"]
            rows = 3 * B * T_FRAMES
            wbytes = 4 * sum(p_.numel() for p_ in model.net.parameters())
            abytes = L2 * rows * 4 * (E2 + 3 * E2 + E2 + E2 + ME2 + E2) * 2
            counted = prof["seg_bytes_per_launch"] / nb
            one_clip = model.net.sample_path() == 1
            h3 = model.net.sample_arith() == 2
            seg["algorithmic_bytes_per_euler_step"] = {"weights_fp32_once": wbytes, "activations_fp32_written_and_read_once": abytes,
                                                       "rows": rows}
            seg["traffic_ratio"] = round(counted / (wbytes + abytes), 2)
            seg["traffic_over_weights_once"] = round(counted / wbytes, 2)
            # what the DESIGN adds on top of that by construction (not waste inside the kernel: the price of XCD-local pipelines)
            lin = 4 * L2 * (3 * E2 * E2 + 2 * E2 * ME2)  # the qkv / MLP Linears' weights, fp32 bytes
            seg["replication_by_construction"] = {
                "xcds_streaming_the_linear_weights_each_step": 8,
                "bytes_per_weight_element_as_read": 4 if (one_clip or h3) else 6,
                "linear_weight_bytes_per_euler_step": (8 * lin) if (one_clip or h3) else (8 * lin * 6 // 4),
                "activation_bytes_per_element_between_gemm_phases": 4 if h3 else 6,
                "what": ("each of the eight XCDs streams every Linear weight once per Euler step (one clip: its time segment needs all "
                         "of them; a batch: its clip does), served by the memory-side cache, not HBM; in the two-piece fp16 form weights "
                         "and GEMM inputs are 4 B per element (two fp16 pieces: split at create / written by the producers); in the "
                         "three-plane bf16 form one clip reads fp32 tiles and splits them in registers, a batch reads planes split at "
                         "create (6 B per element), and GEMM inputs travel as three planes (6 B per element)")}
            seg["traffic_ratio_note"] = ("counted bytes ((2 x FETCH_SIZE + WRITE_SIZE) KiB, Infinity-Cache hits included) per Euler step "
                                         "/ SURVEY 8(d)'s algorithmic bytes per Euler step (weights once in fp32 + phase-crossing "
                                         "activations once each way in fp32); replication_by_construction itemises the part of the excess "
                                         "that is the design's (8 XCD-local weight streams, 6-byte planes)")
        if mf:
            seg["mfma_busy"] = mf
        seg["launch_path"] = roof
        return seg
    if mf:
        roof["mfma_busy"] = mf
    return roof


def mfma_path(config, batch):
    return os.path.join(ROOT, "profiles", f"{ROUND}_pmc_mfma_{config}_b{batch}.json")


def mfma_profile(config, B):
    """MFMA-busy counters of the committed rocprofv3 --pmc pass (profiles/<round>_pmc_mfma_<cfg>_b<B>.json, regenerated by
    `python bench.py --pmc-mfma`), if its source hash matches this tree."""
    pth = mfma_path(config, B)
    if not os.path.exists(pth):
        return None
    p = json.load(open(pth))
    if p.get("source_hash") != source_hash():
        return {"stale": os.path.basename(pth)}
    return {"file": os.path.basename(pth), "dominant": p.get("dominant"), "what": p.get("what")}


def persist_roofline(model, run_once, dcfg, B, config, nb_steps):
    """The persistent offline samplers -- sample_seg_kernel (one clip: time segments over the XCDs) and
    sample_clip_kernel (a batch: one clip per XCD): ONE launch runs every Euler step of every clip.  The launch
    is priced as a whole by HIP events on the launch stream (`achieved` / `frac`: algorithmic fp32 flops of all Euler steps /
    launch duration / the split-bf16 ceiling); `gemm_phases` carries the rate of its qkv / MLP GEMM phases alone, from the
    kernel's own per-phase stamps (every workgroup stamps the 100 MHz wall clock around each XCD-local barrier; last Euler step
    of an extra untimed pass)."""
    import numpy as np
    net = model.net
    path = net.sample_path()
    E_, L_ = dcfg["net"]["embed_dim"], dcfg["net"]["n_layers"]
    ME_ = E_ * dcfg["net"]["mlp_multiplier"]
    M = 3 * B * T_FRAMES
    gemm_fl = 2.0 * M * E_ * ME_  # one qkv / MLP-up / MLP-down GEMM over all clips (3E = ME at mlp x 3)
    arith = net.sample_arith()
    peak = split_ceiling(arith)
    net.profile(True, min_flops=0.0, kernel=3)
    torch.cuda.synchronize()
    for _ in range(4):  # (back to back like the timed region: a single step behind a synchronisation starts at idle clocks -- its
        run_once()      #  launch measured 5 % longer than the same launch inside a train of steps)
    torch.cuda.synchronize()
    ms, launches, flops, nbytes = net.gemm_time(with_bytes=True)
    net.profile(False)
    if not launches:
        return None
    net.set_step_trace(True)
    run_once()
    torch.cuda.synchronize()
    buf = net.step_trace()
    net.set_step_trace(False)
    names = ["patchify"] + sum(([f"ln{l}", f"qkv{l}", f"attn{l}", f"up{l}", f"down{l}"] for l in range(L_)), []) + ["tail"]
    xcc = buf[:, 127].astype(int)
    t = buf[:, :2 * len(names)].astype(np.int64)
    live = [x for x in range(8) if (xcc == x).any() and t[xcc == x][:, 1].max() > 0]  # (XCDs that ran a clip)
    dur = {}  # phase -> us, per XCD: first start -> last arrival at the closing barrier
    for x in live:
        tx = t[xcc == x]
        for p, nme in enumerate(names):
            dur.setdefault(nme, []).append((tx[:, 2 * p + 1].max() - tx[:, 2 * p].min()) / 100.0)
    med = {k: float(np.median(v)) for k, v in dur.items()}
    kinds = {k: float(np.mean([med[f"{k}{l}"] for l in range(L_)])) for k in ("ln", "qkv", "attn", "up", "down")}
    gemm_us = sum(med[f"{k}{l}"] for k in ("qkv", "up", "down") for l in range(L_))
    step_us = float(np.median([(t[xcc == x][:, 2 * len(names) - 1].max() - t[xcc == x][:, 0].min()) / 100.0 for x in live]))
    # (the phases of the eight XCDs run side by side: all clips' flops in one XCD's phase time)
    ach_gemm = 3 * L_ * gemm_fl / (gemm_us * 1e-6) / 1e12
    whole = flops / (ms * 1e-3) / 1e12
    kname = ("sample_seg_kernel (persistent offline sampler: one launch per clip, eight XCD-local pipelines over time segments"
             if path == 1 else
             "sample_clip_kernel (persistent offline sampler: one launch per batch, one clip per XCD, LDS-staged tiles fed by loader waves")
    return {"bound": "mfma",
            "kernel": kname + "; Linears: " + ARITH[arith] + "): the whole launch",
            "arithmetic": arith,
            "achieved": round(whole, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(whole / peak, 4),
            "peak_note": ("dense 16-bit MFMA peak 2500 TFLOP/s / " + {1: "6", 2: "3", 3: "1"}.get(arith, "-") + " MFMAs per fp32 product block "
                          f"(the three-plane bf16 form of rounds 3 - 5 was priced at 2500 / 6 = {PEAK_X6_TFLOPS:.1f}: this launch is at "
                          f"{whole / PEAK_X6_TFLOPS:.3f} of THAT ceiling); against the fp32 MFMA peak (157.3) it is at {whole / PEAK_FP32_MFMA_TFLOPS:.3f}"),
            "launches": int(launches), "avg_launch_us": round(ms * 1e3 / launches, 1),
            "us_per_euler_step": round(ms * 1e3 / launches / nb_steps, 2),
            "flops_per_launch": round(flops / launches),
            "gemm_phases": {"achieved": round(ach_gemm, 2), "frac": round(ach_gemm / peak, 4),
                            "what": "algorithmic fp32 flops of the 18 GEMM phases of one Euler step / the sum of their durations "
                                    "(first workgroup's start to last workgroup's arrival at the closing XCD-local barrier, median "
                                    "over the XCDs at work; stamps of the last Euler step of an extra untimed pass)",
                            "flops_per_gemm_phase": round(gemm_fl)},
            "phase_us": {k: round(v, 2) for k, v in kinds.items()}, "patchify_us": round(med["patchify"], 2),
            "tail_us": round(med["tail"], 2), "us_per_euler_step_in_kernel": round(step_us, 1),
            "traffic": None,
            "note": "achieved = algorithmic fp32 flops of every Euler step of the launch (GEMMs of the denoiser) / HIP-event duration of "
                    "the launch on the launch stream (launch-inclusive = kernel-only to 0.1 %: one launch of 13 - 55 ms); LayerNorm, "
                    "banded attention and the barriers between the phases (phase_us) are inside that time"}


def codec_record(model, dev, B):
    """The other ~14 % of a clip: AutoEncoder.decode / .encode on B clips, HIP-event timed in untimed extra passes, priced
    against both matrix peaks (95.3 / 45.2 GFLOP per clip; the decoder's stride-1 convs run on the split-bf16 pipe
    (conv_x6.hip), everything else on fp32 MFMA: neither ceiling applies to the whole pass, both are given)."""
    ae = model.emb_model
    z = torch.randn(B, 64, T_FRAMES, device=dev)
    x = 0.1 * torch.randn(B, 1, CLIP_SAMPLES, device=dev)
    out = {}
    for name, fn, gf in (("decode", lambda: ae.decode(z), 95.3), ("encode", lambda: ae.encode(x), 45.2)):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        tf = B * gf * 1e9 / (ms * 1e-3) / 1e12
        out[name] = {"ms": round(ms, 3), "tflops": round(tf, 1), "frac_of_fp32_mfma_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 3),
                     "frac_of_split_bf16_ceiling": round(tf / PEAK_X6_TFLOPS, 3)}
    out["what"] = f"{B} clip(s) per pass, mean of 5 passes behind 2 warm-up passes, torch.cuda.Event on the launch stream"
    return out


def run_pmc_mfma(args):
    """One rocprofv3 counter pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; kernel trace only) of this
    very command: per kernel, the fraction of its SIMD-cycles in which the matrix pipe was busy (rocprofv3's MfmaUtil
    expression: MFMA_BUSY summed over the SIMDs / (GRBM_GUI_ACTIVE x SIMDs)), summarised into profiles/."""
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", f"{ROUND}_pmc_mfma")
    os.makedirs(out, exist_ok=True)
    inner = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-legs",
             "--batch-per-gpu", str(args.batch_per_gpu), "--config", args.config]
    d = os.path.join(out, "pass")
    shutil.rmtree(d, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    ctrs = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--"] + inner,
                       cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(f"rocprofv3 --pmc {ctrs} failed:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    agg = {}
    for row in csv.DictReader(open(files[0])):
        key = (row["Kernel_Name"], int(row["Grid_Size"]))
        agg.setdefault(key, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    n_simd, n_xcc = 256 * 4, 8  # (GRBM_GUI_ACTIVE comes summed over its 8 XCC instances: / 8 = the launch's active cycles)
    rows = []
    for (name, grid), v in agg.items():
        mb, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0]), v.get("GRBM_GUI_ACTIVE", [0.0])
        sb = v.get("SQ_BUSY_CYCLES", [0.0])
        mbm, guim, sbm = sum(mb) / len(mb), sum(gui) / max(1, len(gui)), sum(sb) / max(1, len(sb))
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("after::", "")[:100]
        rows.append({"kernel": short, "grid_threads": grid, "launches": len(mb), "SQ_VALU_MFMA_BUSY_CYCLES_mean": round(mbm),
                     "GRBM_GUI_ACTIVE_mean": round(guim), "SQ_BUSY_CYCLES_mean": round(sbm),
                     "mfma_busy_frac": round(mbm / (guim / n_xcc * n_simd), 4) if guim else None,
                     "mfma_busy_total": mbm * len(mb)})
    rows.sort(key=lambda r_: -r_["mfma_busy_total"])
    for r_ in rows:
        del r_["mfma_busy_total"]
    dom = next((r_ for r_ in rows if "sample_seg_kernel" in r_["kernel"] or "sample_clip_kernel" in r_["kernel"] or "gemm_x6" in r_["kernel"]),
               rows[0] if rows else None)
    prof = {"what": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES (summed over the SIMDs) / (GRBM_GUI_ACTIVE / 8 XCC instances x 1024 "
                    "SIMDs), per launch, mean over the launches of a kernel -- rocprofv3's MfmaUtil expression (it takes the max "
                    "over the XCC instances of GRBM_GUI_ACTIVE; the CSV carries their sum): the fraction of SIMD-cycles in which "
                    "the matrix pipe was busy",
            "command": " ".join(inner[1:]), "counters": ctrs, "source_hash": source_hash(), "sources": TRAFFIC_SOURCES,
            "dominant": dom, "kernels": rows[:24]}
    json.dump(prof, open(mfma_path(args.config, args.batch_per_gpu), "w"), indent=1)
    print(json.dumps({"dominant": dom}))


def run_pmc(args):
    """Two rocprofv3 counter passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: separate runs) of this
    very command; kernel trace only, no other trace domain."""
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", f"{ROUND}_pmc")
    os.makedirs(out, exist_ok=True)
    inner = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-legs",
             "--batch-per-gpu", str(args.batch_per_gpu), "--config", args.config]
    agg = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, ctr)
        shutil.rmtree(d, ignore_errors=True)  # (a second --pmc run on the same box must not pick up the first one's CSV)
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--"] + inner,
                           cwd="/tmp", env=env, capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(f"rocprofv3 --pmc {ctr} failed:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            raise SystemExit(f"no counter_collection.csv under {d}")
        for row in csv.DictReader(open(files[0])):
            if row["Counter_Name"] != ctr:
                continue
            key = (row["Kernel_Name"], int(row["Grid_Size"]))
            agg.setdefault(key, {}).setdefault(ctr, []).append(float(row["Counter_Value"]))
    rows, big_bytes, big_n, x6_bytes, x6_n, seg_bytes, seg_n = [], 0.0, 0, 0.0, 0, 0.0, 0
    for (name, grid), v in agg.items():
        f, w = v.get("FETCH_SIZE", [0.0]), v.get("WRITE_SIZE", [0.0])
        fm, wm = sum(f) / len(f), sum(w) / len(w)
        total = (2 * fm + wm) * 1024  # KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md)
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("after::", "")[:100]
        rows.append({"kernel": short, "grid_threads": grid, "launches": len(f), "FETCH_SIZE_KiB_mean": round(fm, 1),
                     "WRITE_SIZE_KiB_mean": round(wm, 1), "bytes_per_launch_corrected": round(total)})
        # the dominant launches: split-K balanced GEMM with the 768 / 256 workgroup grids of the qkv / MLP shapes
        m = re.search(r"gemm_f32_bal_kernel<([^>]*)>", name)
        if m and m.group(1).split(",")[-1].strip() == "0" and total > 5e6:  # MODE 0, not the small launches
            big_bytes += total * len(f)
            big_n += len(f)
        if "gemm_x6" in name and "_kernel" in name:  # gemm_x6_kernel / gemm_x6w_kernel / gemm_x6p_kernel
            x6_bytes += total * len(f)
            x6_n += len(f)
        if "sample_seg_kernel" in name or "sample_clip_kernel" in name:  # the persistent offline samplers: one launch per call
            seg_bytes += total * len(f)
            seg_n += len(f)
    rows.sort(key=lambda r: -r["bytes_per_launch_corrected"] * r["launches"])
    prof = {"note": "bytes_per_launch_corrected = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; includes Infinity-Cache hits",
            "command": " ".join(inner[1:]), "source_hash": source_hash(), "sources": TRAFFIC_SOURCES,
            "gemm_big_mean_bytes_per_launch": round(big_bytes / max(1, big_n)), "gemm_big_launches": big_n,
            "gemm_x6_mean_bytes_per_launch": round(x6_bytes / max(1, x6_n)), "gemm_x6_launches": x6_n,
            "seg_bytes_per_launch": round(seg_bytes / max(1, seg_n)), "seg_launches": seg_n,
            "kernels": rows[:24]}
    json.dump(prof, open(pmc_path(args.config, args.batch_per_gpu), "w"), indent=1)
    json.dump(prof, open(os.path.join(out, os.path.basename(pmc_path(args.config, args.batch_per_gpu))), "w"), indent=1)
    print(json.dumps({k: prof[k] for k in ("gemm_x6_mean_bytes_per_launch", "gemm_x6_launches", "seg_bytes_per_launch", "seg_launches",
                                           "gemm_big_mean_bytes_per_launch", "gemm_big_launches", "source_hash")}))


class Leg:
    """One workload of the benchmark: models of a config, this rank's synthetic inputs, and `step()` = one pass of the hot path."""

    def sampler_path(self):
        if self.stream:
            return ("persistent (one launch per chunk: stream_step_kernel)" if self.model.net.stream_persist()
                    else "one launch per kernel (33 per Euler step)")
        n = self.model.net.sample_launches()
        return {0: "one launch per kernel (33 per Euler step)",
                1: f"persistent, time segments over the XCDs (sample_seg_kernel: {self.B} clip{'s' if self.B > 1 else ''} in {n} launch{'es' if n > 1 else ''})",
                2: "persistent, one clip per XCD (sample_clip_kernel)"}[self.model.net.sample_path()]

    def persistent(self):
        return bool(self.model.net.stream_persist()) if self.stream else self.model.net.sample_path() in (1, 2)


def build_leg(args, dev, world, rank, reuse=None):
    """Models + inputs + step() of the workload `args` names (config, stream, from_audio, batch_per_gpu, global_batch, nb_steps,
    chunk, bf16_tier).  `reuse`: a Leg whose models serve this one too (same config and codec)."""
    from after_amd import parallel, pipeline
    leg = Leg()
    leg.stream = bool(args.stream)
    nb_steps = args.nb_steps or (100 if args.stream else 50)
    codec = "baseAE_causal" if args.stream else "baseAE"
    if reuse is not None:
        model, dcfg, acfg = reuse.model, reuse.dcfg, reuse.acfg
    else:
        model, dcfg, acfg = pipeline.build_models(args.config, codec, dev, seed=0)
        if world > 1:  # identical models everywhere: one RCCL broadcast at start-up
            parallel.broadcast_module(model)  # net, both encoders and the codec (a registered sub-module)
    if args.bf16_tier:
        if args.stream:
            raise SystemExit("--bf16-tier: the tier exists in the offline persistent samplers only")
        if dcfg["net"]["embed_dim"] != 512:
            raise SystemExit("--bf16-tier: the tier exists at the base width (embed_dim 512) only")
        model.net.set_gemm_path(3)
    n_clips = args.global_batch if args.global_batch else args.batch_per_gpu * world
    g = torch.Generator(device="cpu").manual_seed(1000)
    lo, hi = parallel.shard_bounds(n_clips, rank, world)
    B = hi - lo  # this rank's clips (ragged when world does not divide the global batch)
    if B < 1:
        raise SystemExit("every rank needs at least one clip")

    if args.stream:
        # ---- BASELINE config 5: every rank runs B independent streams; one step = one chunk of each
        from after_amd import Streamer
        if dcfg["encoder_time"] is None:
            raise SystemExit("--stream needs an audio-structure config (base / cycle / tiny)")
        st = Streamer(model, model.emb_model, chunk_size=args.chunk, n_signal_timbre=128, max_batch=B,
                      max_nb_steps=nb_steps, share_first_stream=False)
        st.set_nb_steps(nb_steps)
        st.set_guidance_timbre(2.0)
        st.set_guidance_structure(1.0)
        n_samp = args.chunk * st.ae_ratio
        x_all = 0.1 * torch.randn(n_clips, 2, n_samp, generator=g)
        x = x_all[lo:hi].to(dev)
        unit_seconds = n_samp / 44100.0
        out_shape = (n_clips, 1, n_samp)
        gather_buf = torch.empty(out_shape, device=dev) if world > 1 else None

        def step(gather=True):
            audio = st(x)
            if world > 1 and gather:
                audio = parallel.gather_clips(audio, n_clips, out=gather_buf)
            return audio
        workload = (f"{args.config} streaming audio-to-audio: {B} independent streams per GPU, chunk = {args.chunk} "
                    f"latent frames ({unit_seconds * 1e3:.1f} ms of audio), {nb_steps}-step cached CFG sampler "
                    f"(per-step K/V ring caches), causal cached-conv codec + structure encoder, random-init weights")
    else:
        zs_all = torch.randn(n_clips, 64, T_FRAMES, generator=g)
        zt_all = torch.randn(n_clips, 64, T_FRAMES, generator=g)
        x0_all = torch.randn(n_clips, 64, T_FRAMES, generator=g)
        zs, zt, x0 = (t[lo:hi].to(dev) for t in (zs_all, zt_all, x0_all))
        tcond = None
        audio_s = audio_t = None
        if args.from_audio:  # two 0.1 N(0,1) clips, encoded by the codec inside the step
            audio_s = (0.1 * torch.randn(n_clips, 1, CLIP_SAMPLES, generator=g))[lo:hi].to(dev)
            audio_t = (0.1 * torch.randn(n_clips, 1, CLIP_SAMPLES, generator=g))[lo:hi].to(dev)
        if dcfg["encoder_time"] is None:  # midi: synthetic piano roll
            tcond = piano_roll(n_clips, dcfg["net"]["tcond_dim"], dev)[lo:hi].contiguous()
        unit_seconds = CLIP_SECONDS
        out_shape = (n_clips, 1, CLIP_SAMPLES)
        gather_buf = torch.empty(out_shape, device=dev) if world > 1 else None

        def step(gather=True):
            if args.from_audio:
                audio, z = pipeline.audio_to_audio(model, audio_s, audio_t, x0, nb_steps=nb_steps,
                                                   guidance_timbre=2.0, guidance_structure=1.0, time_cond=tcond)
            else:
                audio, z = pipeline.generate_from_latents(model, zs, zt, x0, nb_steps=nb_steps,
                                                          guidance_timbre=2.0, guidance_structure=1.0,
                                                          time_cond=tcond)
            if world > 1 and gather:
                audio = parallel.gather_clips(audio, n_clips, out=gather_buf)
            return audio
        src = "synthetic piano roll + timbre latents" if tcond is not None else "synthetic latents"
        if args.from_audio:
            src = "two synthetic audio clips per item (AutoEncoder.encode x 2 inside the step)"
        workload = (f"{args.config} {'midi' if tcond is not None else 'audio'}-to-audio from {src}, "
                    f"{nb_steps} Euler steps with 3-way CFG (g_t=2, g_s=1), "
                    f"T=256 frames = 11.889 s clips, encoders + sampler + AE decode, random-init weights")
    leg.model, leg.dcfg, leg.acfg, leg.B, leg.n_clips, leg.nb_steps = model, dcfg, acfg, B, n_clips, nb_steps
    leg.step, leg.unit_seconds, leg.out_shape, leg.workload = step, unit_seconds, out_shape, workload
    return leg


def time_leg(leg, warmup, steps, world, dev):
    """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides; the slowest rank's time."""
    out = None
    for _ in range(warmup):
        out = leg.step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = leg.step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed_min = elapsed
    if world > 1:  # the slowest rank is the job's time; the fastest beside it makes a straggler visible
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        tn = te.clone()
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        dist.all_reduce(tn, op=dist.ReduceOp.MIN)
        elapsed, elapsed_min = te.item(), tn.item()
    assert tuple(out.shape) == leg.out_shape and torch.isfinite(out).all()
    return elapsed, elapsed_min


LAUNCH_PATH_SWITCHES = ("AFTER_SAMPLE_PERSIST", "AFTER_SAMPLE_CLIP", "AFTER_STREAM_PERSIST", "AFTER_GEMM_X6")


def expects_persistent(leg, args):
    """Shapes the persistent samplers take by design: any supported width at one or two clips (the one-clip kernel, a
    pair per launch at the shipped width), base / midi width from three clips on (the batch kernel), T = 256; the streaming sampler
    always."""
    if leg.stream:
        return True
    if args.bf16_tier:
        return True  # (the tier exists in the persistent kernels only: a launch-path run would publish the wrong dtype)
    e = leg.dcfg["net"]["embed_dim"]
    return leg.B <= 2 or e == 512


def require_persistent(leg, args):
    """A box in another partition mode (CPX / NPS) or with a co-tenant silently serves every call by launches (the fallback is
    correct, and 10-30 % slower): such a run must not produce a quietly different benchmark line."""
    if leg.persistent() or not expects_persistent(leg, args):
        return
    if args.allow_launch_path or any(os.environ.get(k) == "0" for k in LAUNCH_PATH_SWITCHES):
        return  # an explicit A/B run of the launch path
    raise SystemExit(f"bench.py: the sampler ran as '{leg.sampler_path()}', not on the persistent kernel this workload is quoted on "
                     "(device partitioned, busy, or not 256 CUs?); --allow-launch-path prints the line anyway")


def leg_roofline(leg):
    """The dominant kernel of a short leg, HIP-event timed in two extra untimed passes: the persistent sampler's launch priced as a
    whole (offline: algorithmic fp32 flops / duration / the split-bf16 ceiling; streaming: weight bytes per Euler step / duration / HBM)."""
    net = leg.model.net
    if not leg.persistent():
        return None
    net.profile(True, min_flops=0.0, kernel=3)
    torch.cuda.synchronize()
    for _ in range(2):
        leg.step(gather=False)
    torch.cuda.synchronize()
    ms, launches, flops, nbytes = net.gemm_time(with_bytes=True)
    net.profile(False)
    if not launches:
        return None
    if leg.stream:
        gbs = nbytes / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "stream_step_kernel", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(gbs / 8000.0, 4), "avg_launch_us": round(ms * 1e3 / launches, 1)}
    tf = flops / (ms * 1e-3) / 1e12
    arith = net.sample_arith()
    return {"bound": "mfma", "kernel": "sample_seg_kernel" if net.sample_path() == 1 else "sample_clip_kernel", "arithmetic": arith,
            "achieved": round(tf, 2), "peak": round(split_ceiling(arith), 1), "unit": "TFLOP/s", "frac": round(tf / split_ceiling(arith), 4),
            "frac_of_fp32_mfma_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 3), "avg_launch_us": round(ms * 1e3 / launches, 1)}


def extra_legs(args, dev, head):
    """BASELINE.json's other configs as short legs of the SAME process (the driver runs `bench.py --gpus 1` only): configs[2]'s per-GPU
    shard (base, 8 clips), configs[3] (midi, 8 clips), configs[4] (base + cycle streaming, 8 streams x 100 steps, one step = one
    chunk), configs[0]'s chain (tiny, from audio).  Same timing rule as the headline (warm-up, then K steps between synchronisations);
    no CPU baseline, no counter files."""
    specs = [("b8", dict(batch_per_gpu=8), 1, 3, "BASELINE configs[2]: the per-GPU shard (8 clips) of base B=64 over 8 GPUs"),
             ("midi_b8", dict(config="midi", batch_per_gpu=8), 1, 3, "BASELINE configs[3]: midi, 8 clips"),
             ("b2", dict(batch_per_gpu=2), 1, 4, "a ragged shard's remainder: base, 2 clips -- both in ONE launch of the one-clip kernel"),
             ("stream", dict(config="cycle", stream=True, batch_per_gpu=8), 4, 12,
              "BASELINE configs[4]: base + cycle streaming, 8 streams, 100 cached steps; one step = one 4-frame chunk of every stream"),
             ("tiny_from_audio", dict(config="tiny", from_audio=True, batch_per_gpu=1), 2, 5,
              "BASELINE configs[0]'s chain on the GPU: tiny, two audio clips -> encode x 2 -> encoders -> sampler -> decode")]
    out = {}
    t_all = time.perf_counter()
    for name, over, warm, steps, what in specs:
        a = argparse.Namespace(**vars(args))
        a.stream, a.from_audio, a.global_batch, a.nb_steps, a.bf16_tier = False, False, None, None, False
        for k, v in over.items():
            setattr(a, k, v)
        try:
            leg = build_leg(a, dev, 1, 0, reuse=head if (a.config == args.config and not a.stream) else None)
            el, _ = time_leg(leg, warm, steps, 1, dev)
            if expects_persistent(leg, a) and not leg.persistent():  # (the headline leg exits for this; a side leg says so and moves on)
                raise RuntimeError(f"the sampler ran as '{leg.sampler_path()}', not on the persistent kernel this leg is quoted on")
            ms = el / steps * 1e3
            out[name] = {"what": what, "ms_per_step": round(ms, 3), "steps": steps, "warmup": warm,
                         "value": round(leg.n_clips * leg.unit_seconds / (ms * 1e-3), 2), "unit": "audio_s_per_wall_s",
                         "clips_per_s": None if a.stream else round(leg.n_clips / (ms * 1e-3), 2),
                         "sampler_path": leg.sampler_path(), "roofline": leg_roofline(leg), "workload": leg.workload}
            del leg
        except (Exception, SystemExit) as e:  # a leg must not take the headline line with it
            out[name] = {"what": what, "error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    out["seconds_for_all_legs"] = round(time.perf_counter() - t_all, 2)
    return out


def main():
    import faulthandler
    faulthandler.dump_traceback_later(900, exit=True)  # never hang a GPU box silently
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=None)
    ap.add_argument("--global-batch", type=int, default=None,
                    help="total clips over all ranks (default batch-per-gpu x ranks); ragged shards allowed")
    ap.add_argument("--config", default="base")
    ap.add_argument("--stream", action="store_true", help="BASELINE config 5: streaming, 100 cached steps")
    ap.add_argument("--nb-steps", type=int, default=None, help="Euler steps (50; 100 with --stream)")
    ap.add_argument("--chunk", type=int, default=4, help="--stream: latent frames per chunk")
    ap.add_argument("--from-audio", action="store_true",
                    help="BASELINE config 1's chain: audio -> AutoEncoder.encode x 2 -> encoders -> sampler -> decode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true",
                    help="default headline run only: skip the short legs of the other BASELINE configs (`legs` of the line)")
    ap.add_argument("--allow-launch-path", action="store_true",
                    help="print the line even if the sampler did not run on the persistent kernel the workload is quoted on")
    ap.add_argument("--bf16-tier", action="store_true",
                    help="a SEPARATE leg, never the default: the opt-in bf16 tolerance tier of the persistent offline samplers "
                         "(after_denoiser_set_gemm_path(h, 3); latents within 5e-2 abs / 1e-2 rel-L2 of the fp32 reference)")
    ap.add_argument("--pmc", action="store_true", help="measure the GEMM's HBM traffic with rocprofv3 and exit")
    ap.add_argument("--pmc-mfma", action="store_true", help="measure the kernels' matrix-pipe busy fraction with rocprofv3 and exit")
    args = ap.parse_args()
    if args.batch_per_gpu is None:
        args.batch_per_gpu = 8 if args.stream else 1
    if args.stream and args.config == "base":
        args.config = "cycle"
    nb_steps = args.nb_steps or (100 if args.stream else 50)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    if args.pmc:
        if world != 1:
            raise SystemExit("--pmc is a single-GPU measurement")
        return run_pmc(args)
    if args.pmc_mfma:
        if world != 1:
            raise SystemExit("--pmc-mfma is a single-GPU measurement")
        return run_pmc_mfma(args)
    # test hook: AFTER_BENCH_SHARE_GPU=1 runs all ranks on cuda:0 over gloo, to exercise the
    # multi-rank flow on a single-GPU box (the numbers of such a run mean nothing)
    share = os.environ.get("AFTER_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
        # several processes on ONE GPU: the persistent samplers assume the device to themselves (denoiser.hip: PersistGuard -- "kernels
        # of other processes sharing the GPU are outside this guard: such deployments select the launch path"), and so does this hook
        os.environ.setdefault("AFTER_SAMPLE_PERSIST", "0")
        os.environ.setdefault("AFTER_STREAM_PERSIST", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from after_amd import parallel
    torch.set_grad_enabled(False)
    leg = build_leg(args, dev, world, rank)
    n_ranks_seen = parallel.ranks_seen() if world > 1 else 1
    if n_ranks_seen != world:  # RCCL did not connect every rank: a "scaling" number of this run would be fiction
        raise SystemExit(f"rank {rank}: {n_ranks_seen} ranks answered the all-reduce, expected {world}")
    model, dcfg, acfg, B, n_clips, nb_steps = leg.model, leg.dcfg, leg.acfg, leg.B, leg.n_clips, leg.nb_steps

    elapsed, elapsed_min = time_leg(leg, args.warmup, args.steps, world, dev)
    require_persistent(leg, args)
    head_path = leg.sampler_path()  # (now: a later leg on the same models changes what "the last call" was)
    head_arith = 0 if args.stream else model.net.sample_arith()

    roof = None
    if rank == 0:  # rank 0 only: no collective in this pass
        roof = gemm_roofline(model, lambda: leg.step(gather=False), dev, dcfg, B, args.config, args)
        if roof is not None and not args.stream and model.emb_model is not None:
            roof["codec"] = codec_record(model, dev, B)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if args.stream:
            cpu = cpu_baseline_stream(model, dcfg, acfg, args.chunk, nb_steps, 128)
        else:
            sds = tuple(({k: v.detach().cpu() for k, v in m.state_dict().items()} if m is not None else None)
                        for m in (model.net, model.encoder, model.encoder_time, model.emb_model))
            tc_cpu = piano_roll(1, dcfg["net"]["tcond_dim"], "cpu") if dcfg["encoder_time"] is None else None
            cpu = cpu_baseline(nb_steps, sds, dcfg, acfg, tc_cpu)

    # the other BASELINE configs in the same process (single GPU, the default headline leg only): short timed legs behind the headline
    legs = None
    headline = (world == 1 and not (args.stream or args.from_audio or args.bf16_tier or args.global_batch)
                and args.config == "base" and args.batch_per_gpu == 1 and args.nb_steps is None)
    if rank == 0 and headline and not args.no_legs:
        legs = extra_legs(args, dev, leg)

    if rank == 0:
        audio_s = args.steps * n_clips * leg.unit_seconds
        line = {
            "metric": "audio sec generated / wall sec (xRT), base 50-step @44.1 kHz",
            "value": round(audio_s / elapsed, 2),
            "unit": "audio_s_per_wall_s",
            "clips_per_s": round(args.steps * n_clips / elapsed, 3) if not args.stream else None,
            "xrt_per_stream": round(leg.unit_seconds / (elapsed / args.steps), 3) if args.stream else None,
            "n_gpus": world,
            "n_ranks_seen": n_ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "ms_per_step_fastest_rank": round(elapsed_min / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # (the disclosure holds whenever the sampler's Linears run on bf16 MFMAs: the launch path's gemm_x6 and both
            #  persistent kernels; streaming chunks (<= 96 token rows) and AFTER_GEMM_X6=0 are fp32 MFMA throughout)
            "dtype": ("f32" if (args.stream or head_arith == 0) else
                      "f32 (inputs, results, LayerNorm, attention, GELU, CFG / Euler tail, encoders and codec fp32 as in the reference; the "
                      "qkv / MLP Linears run on the 16-bit matrix pipe with split operands: " + ARITH[head_arith] + " -- measured error vs fp64 "
                      + ("0.63 - 0.72 x" if head_arith == 2 else "0.80 - 0.85 x") + " that of the exact fp32 MFMA chain on the same data "
                      "(tests/test_gemm_gpu.py::test_h3_two_piece_fp16_products_against_the_fp32_chain), latents within 1e-4 of the fp32 "
                      "reference over 50 steps; AFTER_SEG_SPLIT=bf16 / AFTER_CLIP_SPLIT=bf16 = the three-plane form, AFTER_GEMM_X6=0 = fp32 MFMA "
                      "everywhere)"),
            "data": "synthetic",
            "config": {"workload": leg.workload,
                       "batch_per_gpu": args.batch_per_gpu if not args.global_batch else None,
                       "global_batch": n_clips, "nb_steps": nb_steps,
                       "parallelism": f"clip-sharded x{world}" if not args.stream else f"stream-sharded x{world}",
                       "sampler_path": head_path},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if legs is not None:
            line["legs"] = legs
        if args.bf16_tier:  # a different arithmetic: said in the metric, the dtype and the roofline's peak -- this line is never the headline
            line["metric"] += " -- OPT-IN bf16 tolerance tier (NOT the default arithmetic, not comparable with the fp32 line)"
            line["dtype"] = ("bf16 operands (the top planes of the exact three-way splits: round-to-nearest bf16 of weights and "
                             "activations), fp32 accumulate, ONE MFMA per product block in the qkv / MLP Linears of the persistent "
                             "offline samplers; LayerNorm, attention, RoPE, GELU, CFG / Euler tail, encoders and codec fp32 as in the "
                             "default.  Tolerance tier of BASELINE.md section 4(3): latents <= 5e-2 abs / 1e-2 rel-L2 over 50 steps "
                             "(tests/test_bf16_tier_gpu.py), against 1e-4 for the default")
            if isinstance(roof, dict) and roof.get("unit") == "TFLOP/s" and roof.get("achieved"):
                roof["peak"] = PEAK_BF16_MFMA_TFLOPS
                roof["frac"] = round(roof["achieved"] / PEAK_BF16_MFMA_TFLOPS, 4)
                roof["peak_note"] = "dense bf16 MFMA peak 2500 TFLOP/s, one MFMA per product block in this tier"
                for k in ("mfma_busy", "traffic", "traffic_ratio", "traffic_note", "traffic_ratio_note", "algorithmic_bytes_per_euler_step",
                          "replication_by_construction"):
                    roof.pop(k, None)  # (the committed counter passes are the default arithmetic's)
        if args.stream:
            line["metric"] = "audio sec generated / wall sec (xRT, all streams), base+cycle 100-step streaming @44.1 kHz"
        print(json.dumps(line))
    if world > 1:
        dist.barrier()  # rank 0's roofline pass is done: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
