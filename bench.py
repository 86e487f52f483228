#!/usr/bin/env python3
"""Benchmark of the AFTER latent-diffusion sampling path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic clips per GPU:
conditioning encoders -> 50-step CFG rectified-flow sampler -> AutoEncoder.decode
(BASELINE.json configs[1]: base audio-to-audio, 50 steps, batch 1 per GPU, random
init, synthetic latents; one clip = 524288 samples = 11.889 s @ 44.1 kHz).
Prints ONE JSON line on rank 0."""
import argparse
import json
import os
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
NB_STEPS = 50


def cpu_baseline(diffusion, nb_steps, state_dicts, dcfg, acfg):
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
        tc = oracle.encoder1d_forward(sd_et, zs, dcfg["encoder_time"])
        z = oracle.sample(sd_net, dcfg["net"], x0, cond, tc, nb_steps, 2.0, 1.0)
        y = oracle.ae_decode(sd_ae, z, acfg)
        dt = time.perf_counter() - t0
    assert y.shape[-1] == CLIP_SAMPLES
    return {"value": CLIP_SECONDS / dt, "unit": "audio_s_per_wall_s", "cores": cores, "kind": "port",
            "sample": f"1 clip, full path (encoders + {nb_steps}-step sampler + decode), "
                      f"{dt:.2f} s on {cores} threads, torch {torch.__version__} CPU fp32"}


def main():
    import faulthandler
    faulthandler.dump_traceback_later(900, exit=True)  # never hang a GPU box silently
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--config", default="base")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    # test hook: AFTER_BENCH_SHARE_GPU=1 runs all ranks on cuda:0 over gloo, to exercise the
    # multi-rank flow on a single-GPU box (the numbers of such a run mean nothing)
    share = os.environ.get("AFTER_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from after_amd import parallel, pipeline
    torch.set_grad_enabled(False)
    model, dcfg, acfg = pipeline.build_models(args.config, "baseAE", dev, seed=0)
    if world > 1:  # identical models everywhere: one RCCL broadcast at start-up
        for m in (model.net, model.encoder, model.encoder_time, model.emb_model):
            if m is not None:
                parallel.broadcast_module(m)

    B = args.batch_per_gpu
    n_clips = B * world
    g = torch.Generator(device="cpu").manual_seed(1000)
    zs_all = torch.randn(n_clips, 64, T_FRAMES, generator=g)
    zt_all = torch.randn(n_clips, 64, T_FRAMES, generator=g)
    x0_all = torch.randn(n_clips, 64, T_FRAMES, generator=g)
    lo, hi = parallel.shard_bounds(n_clips, rank, world)
    zs, zt, x0 = (t[lo:hi].to(dev) for t in (zs_all, zt_all, x0_all))
    tcond = None
    if dcfg["encoder_time"] is None:  # midi: synthetic piano roll
        tcond = torch.zeros(hi - lo, dcfg["net"]["tcond_dim"], T_FRAMES, device=dev)
        tcond[:, 60:64, 32:96] = 0.7

    def step(gather=True):
        audio, z = pipeline.generate_from_latents(model, zs, zt, x0, nb_steps=NB_STEPS,
                                                  guidance_timbre=2.0, guidance_structure=1.0,
                                                  time_cond=tcond)
        if world > 1 and gather:
            audio = parallel.gather_clips(audio, n_clips)
        return audio

    for _ in range(args.warmup):
        out = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = te.item()
    assert out.shape == (n_clips, 1, CLIP_SAMPLES) and torch.isfinite(out).all()

    # ---- roofline of the dominant kernel (the fp32 MFMA GEMM of the denoiser), measured
    # live with HIP events on the launch stream in one extra, untimed pass
    roof = None
    if rank == 0:
        model.net.profile(True)
        step(gather=False)  # rank 0 only: no collective in this pass
        torch.cuda.synchronize()
        ms, launches, flops = model.net.gemm_time()
        model.net.profile(False)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r1_pmc_hbm_base_b1.json")
        if args.config == "base" and args.batch_per_gpu == 1 and os.path.exists(pmc):
            # not measurable live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
            # command (scripts/pmc_summary.py: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch,
            # Infinity-Cache hits included), averaged over the GEMM launches
            traffic = json.load(open(pmc)).get("gemm_f32_mean_bytes_per_launch")
        # the same launches without per-launch event bracketing: trains of 100 back-to-back
        # launches of the two dominant shapes (qkv / MLP-up and MLP-down), one event pair per train
        b2b = None
        try:
            from after_amd import diag
            M = 3 * B * T_FRAMES
            E_, ME_ = dcfg["net"]["embed_dim"], dcfg["net"]["embed_dim"] * dcfg["net"]["mlp_multiplier"]
            tot_t, tot_f = 0.0, 0.0
            for (n_, k_) in ((ME_, E_), (E_, ME_)):
                a_ = torch.randn(M, k_, device=dev)
                w_ = torch.randn(n_, k_, device=dev)
                o_ = torch.empty(M, n_, device=dev)
                for _ in range(5):
                    diag.gemm(a_, w_, out=o_)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100):
                    diag.gemm(a_, w_, out=o_)
                e1.record()
                torch.cuda.synchronize()
                tot_t += e0.elapsed_time(e1) * 1e-3 / 100
                tot_f += 2.0 * M * n_ * k_
            b2b = round(tot_f / tot_t / 1e12, 2)
        except Exception:  # diagnostics only
            b2b = None
        if launches:
            ach = flops / (ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "gemm_f32_bal_kernel / gemm_f32_dma_kernel (v_mfma_f32_16x16x4_f32)",
                    "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "traffic_unit": "bytes per launch (PMC profile, see profiles/r1_pmc_hbm_base_b1.json)",
                    "launches": int(launches), "avg_launch_us": round(ms * 1e3 / launches, 2),
                    "achieved_back_to_back": b2b,
                    "note": "achieved = per-launch HIP-event bracketing inside the sampler (launch latency "
                            "included); achieved_back_to_back = same kernels in trains of 100 launches",
                    "flops_per_launch": round(flops / launches)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and dcfg["encoder_time"] is not None:
        sds = tuple({k: v.detach().cpu() for k, v in m.state_dict().items()}
                    for m in (model.net, model.encoder, model.encoder_time, model.emb_model))
        cpu = cpu_baseline(args.config, NB_STEPS, sds, dcfg, acfg)

    if rank == 0:
        audio_s = args.steps * n_clips * CLIP_SECONDS
        line = {
            "metric": "audio sec generated / wall sec (xRT), base 50-step @44.1 kHz",
            "value": round(audio_s / elapsed, 2),
            "unit": "audio_s_per_wall_s",
            "clips_per_s": round(args.steps * n_clips / elapsed, 3),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.config} audio-to-audio from synthetic latents, "
                                   f"{NB_STEPS} Euler steps with 3-way CFG (g_t=2, g_s=1), "
                                   f"T=256 frames = 11.889 s clips, encoders + sampler + AE decode, "
                                   f"random-init weights",
                       "batch_per_gpu": B, "global_batch": n_clips,
                       "parallelism": f"clip-sharded x{world}"},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()  # rank 0's roofline pass is done: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
