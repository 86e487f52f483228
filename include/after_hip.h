/*
 * after_hip.h -- C ABI of libafter_hip.so: the MI355X (gfx950) implementation of
 * AFTER's latent-diffusion sampling path.
 *
 * The reference (acids-ircam/AFTER) has no FFI: its boundary for this path is a
 * Python object surface.  Each entry point below names the
 * reference interface it replaces (file:line, relative to the reference root).
 * The Python host in after_amd/ keeps the reference's signatures and calls these.
 *
 * Conventions
 *   - every pointer argument is a DEVICE pointer to contiguous row-major fp32
 *     owned by the caller, unless the parameter is documented as host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all
 *     work is enqueued on it and nothing synchronises the device;
 *   - functions return 0 on success or a negative AFTER_E_* code and never throw;
 *     after_last_error() returns a thread-local description of the last failure;
 *   - the calls that do the path's work (forward / model_forward / sample / encode /
 *     decode / roll / reset ...) never allocate and never synchronise: everything is
 *     provisioned by the CONFIGURATION calls -- *_create, *_enable_*, *_set_* --
 *     which may allocate and synchronise the device (they also run the persistent
 *     samplers' placement census, see after_denoiser_set_stream_persist);
 *   - a handle is not re-entrant (it owns workspaces and streaming state): use
 *     one handle per stream / per concurrent caller.  Different handles are
 *     independent.
 */
#ifndef AFTER_HIP_H
#define AFTER_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define AFTER_OK 0
#define AFTER_E_INVALID (-1)  /* bad argument / unsupported shape          */
#define AFTER_E_HIP (-2)      /* a HIP runtime call failed                 */
#define AFTER_E_CAPACITY (-3) /* request exceeds what *_create provisioned */
#define AFTER_E_NOMEM (-4)

/* CFG row arrangements of the three shipped samplers */
#define AFTER_CFG_API 0    /* after/diffusion/model.py:730-759, clamp 0.01           */
#define AFTER_CFG_EXPORT 1 /* after_scripts/export.py:364-394, clamp 0.1             */
#define AFTER_CFG_MIDI 2   /* after_scripts/export_midi.py:329-358 (c,tc)/(c,-)/(-,-) */

const char* after_last_error(void);
/* library build id string, e.g. "after_hip gfx950 r1" */
const char* after_version(void);

/* ------------------------------------------------------------------ denoiser
 * DenoiserV2 (after/diffusion/networks/transformerv2.py:460-543) constructor
 * arguments that shape the inference graph. */
typedef struct after_denoiser_cfg {
    int n_channels;           /* latent channels C (IN_SIZE)                    */
    int embed_dim;            /* E, multiple of 64; heads = E/64 (:320)         */
    int cond_dim;             /* ZT (timbre vector), > 0                        */
    int tcond_dim;            /* ZS (structure channels), > 0                   */
    int noise_embed_dims;     /* Fourier features of t (:483-485), even         */
    int n_layers;
    int mlp_multiplier;
    int causal;               /* 1: chunk-wise causal mask (:206-216); 0: no mask (general, slower kernel) */
    int local_attention_size; /* sliding window W >= 0 (the shipped, banded kernel);
                                 < 0 = all previous chunks (:211-213, general kernel, no K/V caches) */
    int attention_chunk_size; /* chunk size, 1..8                               */
} after_denoiser_cfg;

/* Order of the `weights` array of after_denoiser_create (state-dict keys of the
 * reference module; P = "denoiser_trans_block."):
 *   0 embedding.0.weight [E, NE+ZT]     1 embedding.0.bias [E]
 *   2 embedding.2.weight [E, E]         3 embedding.2.bias [E]
 *   4 P patchify_and_embed.1.weight [E, C]        5 ....bias [E]
 *   6 P patchify_and_embed_tcond.1.weight [ZS,ZS] 7 ....bias [ZS]
 *   8 P out_proj.0.weight [C, E]        9 P out_proj.0.bias [C]
 *   then for layer l = 0..L-1, base = 10 + 13*l, B = P "decoder_blocks.<l>.":
 *   +0 B self_attention.qkv_linear.weight [3E, E]
 *   +1 B mlp.mlp.0.weight [ME, E]   +2 B mlp.mlp.0.bias [ME]
 *   +3 B mlp.mlp.2.weight [E, ME]   +4 B mlp.mlp.2.bias [E]
 *   +5 B norm1.weight [E]  +6 B norm1.bias [E]
 *   +7 B norm3.weight [E]  +8 B norm3.bias [E]
 *   +9 B linear.weight [2E, E]      +10 B linear.bias [2E]
 *   +11 B tcond_linear.weight [2E, ZS]  +12 B tcond_linear.bias [2E]
 */
#define AFTER_DENOISER_FIXED_WEIGHTS 10
#define AFTER_DENOISER_LAYER_WEIGHTS 13

typedef struct after_denoiser after_denoiser;

/* Copies and re-lays-out the weights into device memory owned by the handle and
 * provisions workspaces for up to `max_rows` network rows (a CFG sample of B
 * clips uses 3B rows), `max_T` latent frames and `max_steps` Euler steps.
 * Replaces: DenoiserV2.__init__ + load_state_dict (transformerv2.py:463-508). */
int after_denoiser_create(const after_denoiser_cfg* cfg, const float* const* weights,
                          int n_weights, int max_rows, int max_T, int max_steps,
                          after_denoiser** out);
void after_denoiser_destroy(after_denoiser* h);

/* out[b,C,T] = net(x, time, cond, time_cond, cache_index)
 * Replaces: DenoiserV2.forward (transformerv2.py:517-543).  time[b] (the
 * reference's [b,1,1] / [b,1,T] inputs are reduced to [b] by the host exactly as
 * :525-528 does).  cache_index must be 0 unless streaming caches were enabled
 * with after_denoiser_enable_cache. */
int after_denoiser_forward(after_denoiser* h, const float* x, const float* time,
                           const float* cond, const float* time_cond, float* out, int b, int T,
                           int cache_index, void* stream);

/* out[B,C,T] = CFG-combined velocity.  Replaces: RectifiedFlow.model_forward
 * (after/diffusion/model.py:721-761); cfg_mode selects the export variants. */
int after_model_forward(after_denoiser* h, const float* x, const float* time, const float* cond,
                        const float* time_cond, float* out, int B, int T, float guidance_timbre,
                        float guidance_structure, float drop_value, int cfg_mode,
                        int cache_index, void* stream);

/* out[B,C,T] = x0 integrated with nb_steps Euler steps of the rectified flow,
 * t = linspace(0,1,nb_steps+1)[:-1], dt = 1/nb_steps.  Replaces:
 * RectifiedFlow.sample (after/diffusion/model.py:763-785); with streaming caches enabled:
 * Streamer.sample (after_scripts/export.py:398-416).  out may alias x0. */
int after_sample(after_denoiser* h, const float* x0, const float* cond, const float* time_cond,
                 float* out, int B, int T, int nb_steps, float guidance_timbre,
                 float guidance_structure, float drop_value, int cfg_mode, void* stream);

/* after_sample can replay a hipGraph of the whole Euler loop (captured once per
 * (B, T, nb_steps, cfg_mode, drop_value) on a private stream; inputs are staged into
 * handle-owned tensors, guidance scalars live in device memory).  Default: off (plain
 * launches on the caller's stream measured faster on ROCm 7.2);
 * enable = 1 or environment AFTER_GRAPH=1 selects the replay. */
int after_denoiser_set_graph(after_denoiser* h, int enable);

/* Arithmetic path of the qkv / MLP Linears (nn.Linear at transformerv2.py:251, :275-283; every other Linear
 * always runs on the fp32 MFMA kernel).  The result is fp32 either way:
 *   0  fp32 MFMA (v_mfma_f32_16x16x4_f32: an exact fp32 fma chain), every size;
 *   1  (default) the bf16-split kernel -- every fp32 operand carried as three bf16 planes, each product formed
 *      as six exact bf16 MFMAs, fp32 accumulation; error vs fp64 <= the fp32 chain's -- for calls with
 *      >= min_rows token rows (rows * T), fp32 MFMA below (streaming chunks);
 *   2  the bf16-split kernel at every size (parity tests);
 *   3  OPT-IN, never a default: the bf16 TOLERANCE TIER (latents within 5e-2 abs / 1e-2 rel-L2 of
 *      the fp32 result over 50 steps, against 1e-4 for modes 0 - 2).  Path selection as mode 1; the persistent offline
 *      samplers (one clip at base width, batches of >= 5 clips) then issue ONE bf16 MFMA per product block -- the operands'
 *      top bf16 planes, i.e. round-to-nearest bf16 of weights and activations, fp32 accumulate -- and fetch only those
 *      planes; LayerNorm, attention, RoPE, GELU and the sampler tail stay fp32.  Calls that run by launches keep mode 1's
 *      arithmetic (at least as accurate).  after_denoiser_gemm_path reports 3 while the tier is on; any other mode
 *      switches it off (bit-identical results to before).
 * min_rows <= 0 keeps the current threshold.  Environment at create: AFTER_GEMM_X6 (0 - 2), AFTER_GEMM_X6_MINROWS. */
int after_denoiser_set_gemm_path(after_denoiser* h, int mode, int min_rows);
int after_denoiser_gemm_path(after_denoiser* h, int* mode, int* min_rows);

/* The streaming sampler (after_sample on a handle with K/V caches: Streamer.sample, after_scripts/export.py:398-416)
 * runs each cached Euler step as ONE persistent launch -- eight independent XCD-local pipelines, clip c on XCD
 * c / ceil(B / 8), every phase of the network behind an XCD-local barrier inside the kernel -- when the geometry allows
 * it: embed 512 / mlp x 3 / eight heads, finite causal window, <= 8 layers, 256 CUs, ceil(B / 8) * T * 3 <= 32 token rows
 * per XCD, <= 128 Euler steps, gemm path != 2.  Otherwise, or with enable = 0 / AFTER_STREAM_PERSIST=0, as the launch-per-kernel sequence.
 * Same arithmetic (fp32 MFMA, a different but fixed K split): the two paths agree to fp32 round-off.
 *
 * Provisioning.  The persistent samplers' buffers (barrier state, 16 x 16-tiled weight copies: + 54 MB at base width,
 * per-XCD activation slices) are allocated, and a dry placement census (are the 256 workgroups placed 32 per XCD?) is run
 * synchronously, by the configuration calls: after_denoiser_enable_cache and after_denoiser_set_stream_persist(h, 1) for
 * the streaming sampler, after_denoiser_create and after_denoiser_set_sample_persist(h, 1) for the offline one.  If the
 * census fails, or memory is short, the handle silently stays on the launch path.  after_sample itself never allocates
 * and never synchronises.
 *
 * Co-residency.  The kernels spin on barriers among their 256 workgroups x 512 threads (80 - 125 KB of LDS each): all of
 * them must be resident at once, i.e. they need the whole device.  Inside one process the library serialises persistent
 * launches of different streams / handles on a device (the later launch waits, on the device, for the earlier one), and a
 * stream under capture takes the launch path.  Kernels of OTHER processes sharing the GPU are outside that guard and can
 * starve the barriers into their ~2-s spin limit: such deployments select the launch path (AFTER_STREAM_PERSIST=0,
 * AFTER_SAMPLE_PERSIST=0).
 *
 * Failures (a barrier or neighbour flag that timed out, a bad placement seen by a real launch) raise STICKY words on the
 * device: every later persistent launch of the handle returns at entry without touching anything.
 *   - The OFFLINE samplers (no K/V caches: RectifiedFlow.sample, which returns a valid tensor or raises, model.py:763-785)
 *     look at a copy of the words before after_sample returns: the call synchronises `stream` once behind its persistent
 *     launch, and a launch that refused or failed is served by the launch-per-kernel path WITHIN THE SAME CALL (AFTER_OK;
 *     the text is kept for after_last_error; the handle stays on the launch path until after_denoiser_set_sample_persist(h, 1)
 *     prepares it again).  after_sample therefore never returns an untouched tensor.  Cost of the synchronisation: one
 *     event wait per call of 13 - 50 ms (measured: DESIGN.md section 7.3).
 *   - The STREAMING sampler defers: the host looks at the words at the start of the next after_sample whose predecessor's
 *     copy has landed, and in after_denoiser_check (which waits for it: the way to validate the last chunk of a stream):
 *     AFTER_E_HIP once, the handle then runs on the launch path; every result since the failing call is invalid (reset the
 *     streamer / repeat the calls).
 * after_denoiser_set_persist_check(h, mode) overrides both: 1 = every persistent after_sample synchronises and reports its
 * own failure (streaming: AFTER_E_HIP from that very call), 0 = every persistent after_sample defers (no host
 * synchronisation inside after_sample; a failed offline call then returns AFTER_OK with an untouched `out`, reported by the
 * next call or after_denoiser_check), -1 = the default above.
 * after_denoiser_stream_persist: *active = 1 when the handle's last streaming shape takes the persistent path. */
int after_denoiser_set_stream_persist(after_denoiser* h, int enable);
/* after_sample for ONE clip without K/V caches (RectifiedFlow.sample, model.py:763-785) as one persistent launch: the same
 * XCD-local pipelines partitioned over time (XCD g owns segment g -- Tseg = 16 or 32 frames -- of the three CFG rows; the
 * attention's left context crosses XCDs through system-scope stores / loads and per-XCD sequence words, no device-wide
 * barrier), the Linears as bf16 x 3 split MFMAs like gemm_x6.  The DEFAULT where eligible: the shipped widths (embed 512 --
 * base, midi -- or 256 -- tiny; mlp x 3, heads of 64), B = 1, T = up to eight segments of 16 or 32 frames (16 .. 128 in steps
 * of 16, 160, 192, 224, 256: XCDs without a segment leave the kernel at once), whole attention chunks per segment,
 * window - 1 <= Tseg, gemm path != 0, no graph replay; otherwise, or with enable = 0 / AFTER_SAMPLE_PERSIST=0, the launch
 * path runs.  Provisioning, co-residency and failures as above.
 *
 * BATCHES (B >= 5 clips; BASELINE config 3's per-GPU shard, config 4): one persistent launch with ONE CLIP PER XCD (clips 8 .. on
 * the same XCDs, one after the other) -- no cross-XCD word at all; the Linears on LDS-staged bf16 x 3 tiles (192 x 192 /
 * 96 x 128) fed by four loader waves per workgroup; a qkv tile is one head of 192 token rows and attends in place (no qkv
 * rows in memory; window - 1 <= 16 and 16 % chunk == 0, else qkv rows + attention items: also AFTER_CLIP_FUSE=0).
 * Eligible: base width (embed 512), T % 16 == 0 and T <= 1024 frames within
 * the handle's capacity, finite causal window, gemm path != 0, no graph replay; provisioned when the handle is created (or
 * grown) with max_rows >= 15 (five clips), sized from the handle's max_T, not from the clips later sampled: 15.5 MB per XCD
 * = 124 MB at max_T = 256, ~ 4x that (~ 520 MB: activation slices and their bf16 planes) at max_T = 1024, + 4.7 MB of regrouped
 * qkv weight planes per layer.  Handles reserved for fewer than five clips never allocate these.  AFTER_SAMPLE_CLIP=0 keeps batches on the launch path,
 * AFTER_SAMPLE_CLIP_MINB=n moves the threshold.
 * _sample_persist: *active = how the last after_sample ran -- 0 by launches, 1 the one-clip kernel, 2 the batch kernel. */
int after_denoiser_set_sample_persist(after_denoiser* h, int enable);
int after_denoiser_sample_persist(after_denoiser* h, int* active);
/* The arithmetic of the qkv / MLP Linears in the last after_sample -- *form = 0: v_mfma_f32_* (the exact fp32 fma chain: gemm path 0),
 * 1: three bf16 planes per operand, six exact MFMAs per product block (gemm_x6.hip: the launch path, and the persistent samplers under
 * AFTER_SEG_SPLIT=bf16 / AFTER_CLIP_SPLIT=bf16), 2: two fp16 pieces per operand under exact power-of-two scales, three exact MFMAs per
 * block (gemm_h3_pipe.h: the DEFAULT of both persistent offline samplers; measured error vs fp64 0.63 - 0.72 x the fp32 chain's,
 * tests/test_gemm_gpu.py), 3: the opt-in bf16 tolerance tier (gemm path 3).  Results are fp32 in every form. */
int after_denoiser_sample_arith(after_denoiser* h, int* form);
/* Persistent launches the last after_sample took (0: it ran by launches of the per-op kernels).  The one-clip kernel serves up to two clips
 * per launch -- an XCD then owns its time segment of BOTH clips' CFG rows and the weights are streamed once for the pair (segments of 16
 * frames: any supported width; of 32 frames: the shipped width on two-piece fp16 operands); AFTER_SEG_PAIR=0: one clip per launch. */
int after_denoiser_sample_launches(after_denoiser* h, int* n);
int after_denoiser_stream_persist(after_denoiser* h, int* active);
/* mode -1 (default): the offline samplers look at their own launch (never an untouched tensor), the streaming sampler defers;
 * 1: every persistent after_sample synchronises `stream` and reports its own failure; 0: every one defers. */
int after_denoiser_set_persist_check(after_denoiser* h, int mode);
/* Waits for the failure words of the handle's last persistent launch (if not yet looked at) and reports them:
 * AFTER_E_HIP if a persistent sampler failed since the last look, AFTER_OK otherwise (also without persistent launches). */
int after_denoiser_check(after_denoiser* h, void* stream);
/* Diagnostics / bench.py's per-phase roofline.  _set_step_trace(h, 1) (a configuration call: allocates the stamp buffer; also
 * AFTER_STEP_TRACE=1 in the environment when the persistent samplers are provisioned) makes every workgroup of a persistent
 * launch stamp the 100 MHz wall clock around each XCD-local barrier; _step_trace synchronises the device and returns
 * out[workgroup][128] of the LAST Euler step launched -- [0] start, [2r-1] / [2r] arrival at / exit from barrier r, then the
 * end; [127] = XCC. */
int after_denoiser_set_step_trace(after_denoiser* h, int enable);
int after_denoiser_step_trace(after_denoiser* h, unsigned long long* out, int n_workgroups);

/* Streaming KV caches (transformerv2.py:143-204; enabled in the reference by the
 * gin binding at after_scripts/export.py:77-79).  cache_size frames (a multiple of
 * the attention chunk) per layer, per diffusion step, per network row;
 * zero-initialised like the reference buffers.  max_steps / max_rows generalise the
 * reference's max_diffusion_steps = 16 / max_batch_size = 4 (:130-131).  After
 * enabling, either drive after_model_forward / after_denoiser_forward with cache_index = i
 * followed by after_denoiser_roll_cache(size, i) per diffusion step, or call after_sample,
 * which then IS that loop (Streamer.sample, export.py:398-416): step i attends over cache
 * slot i and rolls it by the chunk length T (nb_steps <= max_steps). */
int after_denoiser_enable_cache(after_denoiser* h, int cache_size, int max_steps, int max_rows);
int after_denoiser_reset_cache(after_denoiser* h, void* stream);
/* Replaces: DenoiserV2.roll_cache (transformerv2.py:514-515, :171-188). */
int after_denoiser_roll_cache(after_denoiser* h, int size, int cache_index, void* stream);

/* Per-kernel timing hook for bench.py's roofline leg: when enabled, the GEMM
 * launches of the denoiser are bracketed by HIP events on the launch stream;
 * after a stream sync, after_denoiser_gemm_time_ms returns the accumulated
 * duration and launch count since the last call. */
int after_denoiser_profile(after_denoiser* h, int enable);
int after_denoiser_gemm_time_ms(after_denoiser* h, double* total_ms, long long* launches,
                                double* flops);
/* only launches of at least min_flops are bracketed (the dominant qkv / MLP GEMMs: the small
 * patchify / AdaLN projections share the kernel template but not its roofline) */
int after_denoiser_profile_min_flops(after_denoiser* h, double min_flops);
/* which kernel's launches are bracketed: 0 both GEMM kernels, 1 the bf16-split kernel (gemm_x6.hip) only, 2 the fp32 MFMA
 * kernel (gemm.hip) only -- the two have different rooflines -- 3 the persistent streaming sampler (one launch per
 * after_sample call on a handle with K/V caches; flops = its GEMMs, bytes = every weight once per Euler step).  Classes
 * 0-2 make the streaming sampler take its launch-per-kernel path while profiling is on. */
int after_denoiser_profile_kernel(after_denoiser* h, int which);
/* as after_denoiser_gemm_time_ms, plus the algorithmic bytes (A + W + C, fp32) of those launches:
 * the streaming path's GEMMs (<= 96 tokens) are weight-streaming launches priced against HBM */
int after_denoiser_gemm_time2(after_denoiser* h, double* total_ms, long long* launches, double* flops,
                              double* bytes);

/* --------------------------------------------------------------- autoencoder
 * AutoEncoder (after/autoencoder/networks/SimpleNetsStream.py:831-954): PQMF +
 * dilated-conv encoder / decoder.  The TorchScript export keeps exactly
 * encode / decode (after_scripts/export_autoencoder.py:235-265). */
#define AFTER_AE_MAX_STAGES 8
typedef struct after_ae_cfg {
    int pqmf_bands;      /* 16 (= in_channels of the conv nets)                       */
    int channels;        /* BASE_CHANNELS                                             */
    int z_channels;      /* LATENT_SIZE                                               */
    int n_stages;        /* len(factors)                                              */
    int n_dilations;     /* len(dilations) = ResnetBlock1d per stage                  */
    int kernel_size;     /* 3                                                         */
    int use_norm;        /* GroupNorm(min(C,8)) in every ConvBlock1d                  */
    int use_loudness;    /* decoder emits 2x bands, x * sigmoid(amplitude) (:644-646) */
    int causal;          /* cached_conv.get_padding mode (baseAE.gin:32-33)           */
    int multipliers[AFTER_AE_MAX_STAGES + 1];     /* encoder channel multipliers      */
    int dec_multipliers[AFTER_AE_MAX_STAGES + 1]; /* int(m * decoder_ratio), reversed */
    int factors[AFTER_AE_MAX_STAGES];             /* encoder order                    */
    int dilations[AFTER_AE_MAX_STAGES];
    int encoder_out_channels; /* 0 = z_channels; 2 * z_channels with a VAEBottleneck (SimpleNetsStream.py:864-867):
                               * after_ae_encode then writes [B, encoder_out_channels, T]          */
    int use_noise;            /* Decoder1d's NoiseGenerator branch (SimpleNetsStream.py:499-550, :622-650): three plain
                               * k = 3 stride-2 convs (in -> 128 -> 128 -> 5 * pqmf_bands) on the last decoder stage's
                               * output, filtered uniform noise added to the band signal; whole-clip decoding only,
                               * centred padding only; every decode takes its draws from after_ae_set_noise          */
} after_ae_cfg;

/* Order of the `weights` array (reference state-dict keys).
 * CB(p) = the 7 tensors of a ConvBlock1d under prefix p:
 *     p.net.0.gn.weight, p.net.0.gn.bias (NULL when use_norm = 0), p.net.1.alpha,
 *     p.net.1.beta, p.net.2.weight_g, p.net.2.weight_v, p.net.2.bias
 * WN(p) = p.weight_g, p.weight_v, p.bias ;  SN(p) = p.alpha, p.beta
 *   pqmf.forward_conv.weight, pqmf.inverse_conv.weight
 *   encoder.net.0:  CB(.net.branches.0.0) CB(.net.branches.0.1) [WN(.net.branches.1) iff
 *                   pqmf_bands != channels*multipliers[0]]
 *   encoder.net.{1+i} (stage i): for j < n_dilations: CB(.net.j.net.branches.0.0)
 *                   CB(.net.j.net.branches.0.1); SN(.net.{nd}); WN(.net.{nd+1})
 *   SN(encoder.net.{n+1}) WN(encoder.net.{n+2})
 *   WN(decoder.net.0)
 *   decoder.net.{1+i}: SN(.net.0) WN(.net.1) then for j: CB(.net.{2+j}.net.branches.0.0)
 *                   CB(.net.{2+j}.net.branches.0.1)
 *   CB(decoder.synth.branches.0.net.0) CB(decoder.synth.branches.0.net.1)
 *   use_noise: decoder.noise_module.net.{0,2,4}.weight [Cout, Cin, 3] and .bias, in that order (plain Conv1d, no weight norm)
 */
typedef struct after_ae after_ae;

/* Folds weight-norm, packs the conv weights for the MFMA kernels and provisions
 * workspaces for max_batch clips of max_samples audio samples.
 * Replaces: AutoEncoder.__init__ + load_state_dict (SimpleNetsStream.py:834-896). */
int after_ae_create(const after_ae_cfg* cfg, const float* const* weights, int n_weights,
                    int max_batch, int max_samples, after_ae** out);
void after_ae_destroy(after_ae* h);
/* total stride of the codec: pqmf_bands * prod(factors) (2048 for baseAE) */
int after_ae_ratio(const after_ae* h);

/* z[B, Z, L/ratio] = encode(x[B, 1, L]); L a multiple of the ratio.
 * Replaces: AutoEncoder.encode (SimpleNetsStream.py:918-941; the bottleneck is the
 * identity on z at inference, :753-760). */
int after_ae_encode(after_ae* h, const float* x, float* z, int B, int L, void* stream);
/* Two lanes of streams in ONE streaming encoder (the Streamer's structure and timbre inputs go through the same codec,
 * export.py:161-166 loads it twice): after after_ae_set_stream_lanes(h, n) -- on a freshly enabled / reset stream,
 * 2 n <= max_batch -- the encoder's context rows [0, n) and [n, 2 n) are independent streams groups with their own chunk
 * parity.  after_ae_encode_rows encodes x[B, 1, L] against the context rows [row0, row0 + B): one lane (row0 = 0 or n,
 * B <= n) or both in one pass (row0 = 0, B = 2 n; refused with AFTER_E_INVALID while the lanes are a chunk apart). */
int after_ae_set_stream_lanes(after_ae* h, int lane_rows);
int after_ae_encode_rows(after_ae* h, const float* x, float* z, int B, int L, int row0, void* stream);
/* x[B, 1, T*ratio] = decode(z[B, Z, T]).  Replaces: AutoEncoder.decode (:943-954). */
int after_ae_decode(after_ae* h, const float* z, float* x, int B, int T, void* stream);
/* use_noise codecs: the uniform [0, 1) draws of the NEXT decode, u[B, T * ratio / (8 * pqmf_bands), pqmf_bands, 8] on
 * the device (the reference's torch.rand_like(ir), SimpleNetsStream.py:545; the kernel forms 2 u - 1).  The pointer is
 * consumed by that decode; a decode without it fails with AFTER_E_INVALID. */
int after_ae_set_noise(after_ae* h, const float* u);
/* decode(z, with_multi=True) (:943-954): additionally writes x_multiband[B, M, T*ratio/M], the
 * decoder output after the loudness gate and before the PQMF synthesis bank (:643-646). */
int after_ae_decode_multi(after_ae* h, const float* z, float* x, float* multiband, int B, int T,
                          void* stream);
/* out[0] = mean(ELU(|z| - scale)) + 1 over the n elements of z: the regulariser
 * ReluBottleneck.forward returns beside z (SimpleNetsStream.py:742-760 -> SimpleLatentReg,
 * after/autoencoder/core.py:189-198).  Deterministic (one workgroup, fixed order). */
int after_latent_reg(const float* z, long long n, float scale, float* out, void* stream);
/* TanhBottleneck.forward (SimpleNetsStream.py:719-740): z <- scale * tanh(z), in place (its noise term
 * sigma * randn and its zero regulariser are the caller's). */
int after_bottleneck_tanh(float* z, long long n, float scale, void* stream);
/* VAEBottleneck.forward (SimpleNetsStream.py:763-785) on the encoder output zraw [B, 2Z, T] (mean | scale along
 * the channels): std = softplus(scale) + 1e-2; z = noise * std + mean (noise [B, Z, T] ~ N(0, 1) from the caller;
 * NULL = the mean itself); mean_out [B, Z, T] (may be NULL); kl[0] = mean over (b, t) of the channel sums of
 * mean^2 + var - log var - 1.  Deterministic (one workgroup reduces in a fixed order). */
int after_bottleneck_vae(const float* zraw, const float* noise, float* z, float* mean_out, float* kl, int B, int Z,
                         int T, void* stream);
/* multiband[B, M, L/M] = pqmf(x) / x = pqmf.inverse(multiband) (pqmf.py:286-301) */
int after_ae_pqmf_forward(after_ae* h, const float* x, float* mb, int B, int L, void* stream);
int after_ae_pqmf_inverse(after_ae* h, const float* mb, float* x, int B, int Tm, void* stream);

/* Streaming (real-time) mode of the causal, GroupNorm-free codec: after_ae_encode /
 * after_ae_decode become stateful and process consecutive chunks of the same B streams.
 * Every temporal conv keeps its left context (the last (k-1)*dilation activated input
 * samples), the transposed convs run in their padding-0 overlap-add form, the PQMF banks
 * keep K-1 samples / frames.  This is `cached_conv` (acids-ircam/cached_conv, the
 * dependency behind `cc.use_cached_conv(True)`) applied to the causal model:
 * export_autoencoder.py:293-303, CachedConv1d / CachedConvTranspose1d / CachedPadding1d.
 * Chunked output == the offline causal model with ConvTranspose1d padding 0 on the
 * concatenated stream.  Returns AFTER_E_INVALID for a non-causal or GroupNorm codec.
 * enable = 0 returns to the offline path (state is kept). */
int after_ae_enable_streaming(after_ae* h, int enable);
/* zero every streaming state (start of a new stream) */
int after_ae_reset_state(after_ae* h, void* stream);

/* Streaming twin of a NON-causal codec's encoder: what export_autoencoder.py:305-312 packs into
 * export_stream.ts for a non-causal model -- `model.encoder` rebuilt under cc.use_cached_conv(True)
 * with CachedGroupNorm.stream = True (SimpleNetsStream.py:95-147), while the PQMF, the bottleneck and
 * the decoder stay the offline modules.  after_ae_encode becomes stateful over consecutive chunks of
 * the same B streams:
 *   - every conv keeps its l + r frames of (activated) context and reads only the past
 *     (cached_conv.CachedConv1d on the centred padding (l, r));
 *   - ResnetBlock1d's shortcut input is delayed by block1's r frames (cc.AlignBranches,
 *     SimpleNetsStream.py:236-249); Downsample1d adds (f - (r + cd) % f) % f frames of input delay;
 *   - GroupNorm statistics cover the previous `gn_window_samples` of the stream + the chunk
 *     (the reference's "automatic" window = the length of the first call, 131072 samples in the
 *     export script); ignored for a GroupNorm-free codec;
 *   - the PQMF analysis is the offline, zero-padded one on each chunk, as in the reference.
 * The latents lag the offline encoder by after_ae_encoder_delay() frames; without GroupNorm they are
 * the offline latents of the concatenated multiband stream, that many frames late, for any chunking.
 * after_ae_decode is unaffected.  AFTER_E_INVALID for a causal codec (use after_ae_enable_streaming). */
int after_ae_enable_encoder_streaming(after_ae* h, int enable, int gn_window_samples);
int after_ae_encoder_delay(const after_ae* h);
/* CachedGroupNorm(stream=True) on the decoder (the same export binds it for both twins): every
 * GroupNorm of after_ae_decode normalises over the previous `window_latent_frames` of the stream +
 * the call's frames (the reference: the length of the first decode after construction, 64 frames in
 * the export script).  0 = plain per-call GroupNorm.  after_ae_reset_state zeroes the windows. */
int after_ae_set_decoder_gn_window(after_ae* h, int window_latent_frames);

/* ------------------------------------------------------- conditioning encoders
 * encoder_time: Encoder1D (after/diffusion/networks/encoder.py:116-322), causal
 * padding through the scoped gin binding (after/diffusion/configs/base.gin:55). */
typedef struct after_encoder1d_cfg {
    int in_size;
    int n_blocks;     /* len(channels)                                              */
    int channels[8];
    int ratios[8];    /* [1] + ratios of the gin config: stride of block i's pool   */
    int kernel_size;  /* 5                                                          */
    int causal;
    int use_tanh;
} after_encoder1d_cfg;
/* weights: for block i: BN(.net.i.net.0.net.branches.0.0: weight, bias, running_mean,
 * running_var) WN(...branches.0.2) BN(...branches.0.3) WN(...branches.0.6)
 * WN(.net.i.net.1); then the final V2ConvBlock1D .net.n: BN WN BN WN. */
typedef struct after_encoder1d after_encoder1d;
int after_encoder1d_create(const after_encoder1d_cfg* cfg, const float* const* weights,
                           int n_weights, int max_batch, int max_T, after_encoder1d** out);
void after_encoder1d_destroy(after_encoder1d* h);
/* out[B, channels[-1], T / prod(ratios)] = encoder_time(z[B, in_size, T])
 * Replaces: Encoder1D.forward / forward_stream offline (encoder.py:273-322). */
int after_encoder1d_forward(after_encoder1d* h, const float* z, float* out, int B, int T,
                            void* stream);
/* Streaming twin (`encoder_time.forward_stream` under cc.use_cached_conv(True): export.py:17,
 * 438-441): after_encoder1d_forward becomes stateful over consecutive chunks, each causal
 * conv keeping its left context.  Chunked output == offline causal output of the whole
 * stream.  AFTER_E_INVALID unless cfg.causal. */
int after_encoder1d_enable_streaming(after_encoder1d* h, int enable);
int after_encoder1d_reset_state(after_encoder1d* h, void* stream);

/* encoder: ECAPATDNN (after/diffusion/networks/ecapa_encoder.py:458-666) with
 * pooling + global context (every shipped config). */
typedef struct after_ecapa_cfg {
    int in_size, out_dim;
    int n_blocks;     /* len(channels) = 1 TDNN + (n-2) SE-Res2Net + MFA            */
    int channels[8];
    int kernel_sizes[8];
    int dilations[8];
    int res2net_scale, se_channels, attention_channels;
    int use_tanh;
} after_ecapa_cfg;
/* weights: TD(p) = p.conv.conv.weight, p.conv.conv.bias, p.norm.weight, p.norm.bias,
 * p.norm.running_mean, p.norm.running_var;  CV(p) = p.conv.weight, p.conv.bias
 *   TD(blocks.0); for i in 1..n-2: TD(blocks.i.tdnn1) TD(blocks.i.res2net_block.blocks.j)
 *   for j < scale-1, TD(blocks.i.tdnn2) CV(blocks.i.se_block.conv1) CV(blocks.i.se_block.conv2)
 *   [CV(blocks.i.shortcut) iff channels differ]; TD(mfa); TD(asp.tdnn); CV(asp.conv);
 *   asp_bn.weight, .bias, .running_mean, .running_var; CV(fc). */
typedef struct after_ecapa after_ecapa;
int after_ecapa_create(const after_ecapa_cfg* cfg, const float* const* weights, int n_weights,
                       int max_batch, int max_T, after_ecapa** out);
void after_ecapa_destroy(after_ecapa* h);
/* out[B, out_dim] = encoder(z[B, in_size, T]).  Replaces: ECAPATDNN.forward
 * (ecapa_encoder.py:567-624; regularisation "ac" leaves Z unchanged). */
int after_ecapa_forward(after_ecapa* h, const float* z, float* out, int B, int T, void* stream);

/* ---------------------------------------------------------------- UNET1D denoiser
 * The Conv1d / GroupNorm / SiLU / FiLM alternative to DenoiserV2
 * (after/diffusion/networks/unet1d.py:254-429; no shipped gin config selects it).
 * Built for time_cond_channels > 0, cond_channels > 0; GroupNorm(min(16, C/4), C) must be well defined for
 * every block.  n_attn_layers > 0 (unet1d.py:339, 350, 372): SelfAttention1d (blocks.py:201-243) behind the conv block of
 * down_layers i >= n - n_attn_layers (i >= 1; 4 heads), of up_layers j < n_attn_layers (j < n - 1; 4 heads) and of the
 * middle block (in_c // 32 heads); head sizes 4 .. 64 (powers of two). */
typedef struct after_unet1d_cfg {
    int in_size, out_size;   /* out_size <= 0: = in_size                                  */
    int n_blocks;            /* len(channels)                                             */
    int channels[8];
    int ratios[8];           /* [1] + ratios of the constructor (unet1d.py:283)           */
    int kernel_size;         /* odd                                                       */
    int time_channels, time_cond_in_channels, time_cond_channels, cond_channels;
    int use_res_last;
    int n_attn_layers;       /* 0 .. n_blocks                                             */
} after_unet1d_cfg;
/* weights (reference state_dict keys), CB(p) = p.conv1.{weight,bias} p.gn1.{weight,bias}
 * p.conv2.{weight,bias} p.gn2.{weight,bias} p.time_mlp.{0,2}.{weight,bias}
 * p.cond_mlp.{0,2}.{weight,bias} [p.to_out.{weight,bias} when the block takes a skip]:
 *   cond_emb_time.i.0.{weight,bias}, i = 0..n
 * SA(p) = p.self_attn.norm.{weight,bias} p.self_attn.qkv_proj.{weight,bias} p.self_attn.out_proj.{weight,bias},
 * present only where the block has a self-attention layer:
 *   for i < n: CB(down_layers.i.conv) [SA(down_layers.i)] down_layers.i.pool.{weight,bias}
 *   CB(middle_block.conv) [SA(middle_block)]
 *   for j < n: up_layers.j.up[.1].{weight,bias} (NULL, NULL when `up` is Identity) CB(up_layers.j.conv) [SA(up_layers.j)] */
typedef struct after_unet1d after_unet1d;
int after_unet1d_create(const after_unet1d_cfg* cfg, const float* const* weights, int n_weights,
                        int max_batch, int max_T, after_unet1d** out);
void after_unet1d_destroy(after_unet1d* h);
/* out[B, out_size, T] = net(x[B, in_size, T], time[B], cond[B, cond_channels],
 * time_cond[B, time_cond_in_channels, T]).  Replaces: UNET1D.forward (unet1d.py:374-414). */
int after_unet1d_forward(after_unet1d* h, const float* x, const float* time, const float* cond,
                         const float* time_cond, float* out, int B, int T, void* stream);
/* RectifiedFlow.model_forward / .sample (model.py:721-785) around UNET1D, entirely on the device:
 * the 3x classifier-free-guidance batch, the network on 3 B rows, the guidance combination and the
 * Euler update.  cfg_mode as after_sample.  Needs a handle created with max_batch >= 3 B and
 * out_size == in_size. */
int after_unet1d_model_forward(after_unet1d* h, const float* x, const float* time, const float* cond,
                               const float* time_cond, float* out, int B, int T, float g_timbre,
                               float g_structure, float drop_value, int cfg_mode, void* stream);
int after_unet1d_sample(after_unet1d* h, const float* x0, const float* cond, const float* time_cond,
                        float* out, int B, int T, int nb_steps, float g_timbre, float g_structure,
                        float drop_value, int cfg_mode, void* stream);

/* ------------------------------------------------------------ diagnostics
 * Not part of the reference's surface: the fp32 MFMA GEMM behind every Linear,
 * exposed for unit parity tests and roofline measurements.
 *   C[M,N] = epi(A[M,K] * W[N,K]^T + bias);  epilogue 0 none, 1 GELU(erf), 2 + R[M,N], 3 ReLU, 4 sigmoid
 * force_mt/force_nt = 0 lets the library pick the tile. */
/* diagnostics: when non-NULL, each workgroup of after_gemm_f32's kernel records at
 * dbg[8*wg..]: four s_memtime stamps (start, operands landed, MFMAs done, stores done),
 * the 100 MHz wall clock at start and end, and the hardware CU id. */
void after_gemm_set_debug(unsigned long long* dbg);
int after_gemm_f32(const float* A, int lda, const float* W, int ldw, const float* bias,
                   const float* R, int ldr, float* C, int ldc, int M, int N, int K, int epilogue,
                   int force_mt, int force_nt, void* stream);

/* Diagnostics / tests for after_amd/csrc/gemm_x6.hip, the kernel behind the denoiser's qkv / MLP Linears
 * (reference transformerv2.py:251, :275-283) on the bf16-split path: every fp32 operand is carried as its
 * three bf16 planes (x = h + m + l exactly), products are six v_mfma_f32_16x16x32_bf16 per 32-deep step with
 * fp32 accumulation, the result is fp32 (C) or again planes (C3) for a following GEMM.  Plane storage
 * ("x6 blocks", after_amd/csrc/common.h): for a [R][K] matrix, K % 32 == 0, 1-KB blocks
 * [ceil(R / 16)][K / 32][plane] of 16 rows x 32 columns, i.e. 16 * ceil(R / 16) * 3 * K unsigned shorts;
 * after_gemm_x6_offset gives the element offset of (row, plane, column).  after_gemm_x6_split writes the blocks
 * of an fp32 matrix; tile 0 = chosen by shape (after_gemm_x6_pick_tile), else an id of the tile table in
 * gemm_x6.hip.  Exactly one of C / C3 (C3 needs N % 32 == 0). */
int after_gemm_x6_split(const float* W, int ldw, unsigned short* W3, int N, int K, void* stream);
long long after_gemm_x6_offset(int row, int plane, int col, int K);
void after_gemm_x6_set_debug(unsigned long long* dbg); /* per-workgroup cycle stamps, as after_gemm_set_debug */
int after_gemm_x6_pick_tile(int M, int N, int K);
int after_gemm_x6(const unsigned short* A3, const unsigned short* W3, const float* bias, const float* R, int ldr,
                  float* C, unsigned short* C3, int ldc, int M, int N, int K, int epilogue, int tile, void* stream);

/* Test / measurement entry (after_amd/csrc/split_diag.hip; no product path calls it): C [M][N] = A [M][K] W[N][K]^T in ONE of the
 * arithmetics the denoiser's Linears run in, one wave per 16 x 16 block, operands split on the fly --
 * mode 0: v_mfma_f32_16x16x4_f32, the exact fp32 fma chain (reference: F.linear in fp32, transformerv2.py:251, :275-283);
 * mode 1: three bf16 planes per operand, six bf16 MFMAs per 32-deep block (gemm_x6.hip, the launch path and sample_seg_kernel);
 * mode 2: two fp16 pieces per operand (a 22-bit significand) under exact power-of-two scales derived from the bounds
 *         a_bound >= max|A|, w_bound >= max|W|, three f16 MFMAs per block (gemm_h3_pipe.h, the batch sampler's Linears).
 * M, N multiples of 16, K of 32.  tests/test_gemm_gpu.py holds the split forms' error vs fp64 to the fp32 chain's. */
int after_diag_split_gemm(const float* A, const float* W, float* C, int M, int N, int K, int mode, float a_bound, float w_bound,
                          void* stream);

/* One Conv1d layer on the time-major conv path (act(x) into the zero-haloed [B][T][C] buffer, then
 * the conv as a balanced LDS-DMA GEMM) for parity tests against a plain fp32 conv and for the
 * per-layer tile sweeps (scripts/bench_conv.py).  w [Cout, Cin, k] (torch.nn.Conv1d layout),
 * y[b, co, n] = bias[co] + sum_{t, ci} w[co, ci, t] act(x)[b, ci, n*stride + t*dil - left_pad] (0 outside).
 * act: 0 none, 1 snake(alpha = beta = 1), 2 SiLU, 3 ReLU, 4 tanh.
 * run mode bits: 1 activate + halo (x [B][Cin][T] or NULL = internal buffer), 2 conv (y [B][Cout][Tout]
 * or NULL = internal time-major buffer), 4 fused GroupNorm statistics, 8 residual add, 16 the bf16-pipe form of the
 * layer (conv_x6.hip: stride 1, <= 3 taps, Cout % 4 == 0; refused otherwise). */
typedef struct after_convtm after_convtm;
int after_convtm_create(const float* w, const float* bias, int B, int Cin, int Cout, int T, int Tout,
                        int k, int dil, int stride, int left_pad, int act, after_convtm** out);
int after_convtm_run(after_convtm* h, const float* x, float* y, int mode, void* stream);
void after_convtm_destroy(after_convtm* h);
/* 0 = heuristic; 1..11 pin a tile configuration (conv_tm.hip: launch_tm_id) */
void after_convtm_set_tile(int id);
/* tile of the bf16-pipe form of the layer (after_convtm_run mode bit 4; conv_x6.hip): 0 = by shape, 1..8 pin a tile
   (conv_x6.hip: launch_conv_x6) */
void after_convtm_set_x6_tile(int id);
/* number of conv launches this process has sent down the bf16-pipe path so far (diagnostic: the tests check that the
   decoder's MFMA-bound convs take it by default and that AFTER_CONV_X6=0 keeps them off it) */
long long after_conv_x6_launches(void);
/* ... of them on TWO fp16 pieces per operand (conv_x6.hip's SPLIT tiles, round 6): the launches whose input a whole-clip GroupNorm
   bounds -- |snake(GroupNorm(x))| <= sqrt(n) max|gamma| + max|beta| + max(1 / snake beta) -- so that an exact power-of-two scale keeps
   the pieces inside fp16's range (gemm_h3_pipe.h; three MFMAs per product block instead of six); AFTER_CONV_H3=0 keeps them on three
   bf16 planes (A/B switch).  Snake-only inputs and the GroupNorm-free causal codec have no bound and stay on three planes. */
long long after_conv_h3_launches(void);
/* number of GroupNorm -> Snake -> Conv1d(k = 1) blocks (the second conv of a ResnetBlock1d, SimpleNetsStream.py:196-254) this
   process has run as ONE launch (conv_tm.hip: conv1_act_kernel -- no activated tensor in memory) instead of act_pad + conv;
   AFTER_AE_FUSE_K1=0 keeps them on the two launches (A/B switch) */
long long after_conv1_act_launches(void);

#ifdef __cplusplus
}
#endif
#endif /* AFTER_HIP_H */
