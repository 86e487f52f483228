// Third-generation conv path of the codec and the conditioning encoders: TIME-MAJOR
// activations ([B][T][C], channels contiguous) so that a 1-D convolution IS a GEMM with
// K-contiguous operands on both sides and runs on the balanced LDS-DMA GEMM machinery of
// gemm.hip (gemm_pipe.h):
//
//   y[b, n * ostride + ooff, co] = bias[co] + res[...] +
//        sum_{tap, ci} xp[b, HALO + n * istride + toff0 + tap * dil, ci] * w[co, tap * Cp + ci]
//
//   M = output positions of one clip (and phase), N = Cout, K = taps * Cp.  A row n of the
//   implicit A matrix starts at xp row (HALO + n*istride + toff0); with dil == 1 its K = taps*Cp
//   floats are CONTIGUOUS in memory (a plain GEMM with lda = istride * Cp: the strided
//   Downsample1d, the two-tap phases of the transposed Upsample1d, every k = 1 conv and the
//   undilated k = 3 / k = 5 convs); with dil > 1 the 32-deep slab s of K lives at
//   s*32 + tap(s) * (dil-1) * Cp, one scalar multiply-shift per DMA piece.
//
// What this buys over a [B][C][T] layout with time along N (round 1's conv kernels): the X
// operand is read from LDS as ds_read_b128 k-quads like the weights (was 4 x ds_read_b32 per
// 16x16 block and k-step), the DMA pieces need no per-stage address arithmetic, the XCD tile map
// keeps an XCD on a compact range of output positions so the haloed input is fetched once and the
// weights once per XCD (round 1: every XCD re-read everything, 6.5x the algorithmic bytes), and
// small-T layers (conditioning encoders, first decoder stage) get the split-K tiles that fill
// 256 CUs.
//
// act_pad_tm_kernel materialises act(GroupNorm / BatchNorm-affine(x)) once per conv into the
// zero-haloed buffer (conv padding = reading the halo = the reference's pad-after-activation,
// cached_conv.Conv1d); the conv epilogue accumulates the NEXT GroupNorm's statistics.
#include <cstdio>
#include <cstdlib>
#include <new>

#include "conv.h"
#include "gemm_pipe.h"
#include "gemm_h3_pipe.h"

namespace after {
namespace {

constexpr int HALO = 32;  // zero rows on both sides of every clip
static_assert(HALO == kConvTmHalo, "conv.h");
// The statistics of one (clip, group) are spread over kStatSub accumulator pairs (workgroup id
// mod kStatSub) so that the fp64 atomics of several hundred workgroups do not serialise on 16
// addresses; the consumer adds the sub-slots in a fixed order.
constexpr int kStatSub = 8;

// sin(x) on the hardware sine unit (v_sin_f32 takes revolutions): the argument is reduced to
// [-0.5, 0.5] revolutions with 1 / 2pi split in two floats and FMAs, so its error does not grow with
// |x|.  Measured against fp64 on 4 M uniform samples each of |x| < 4, 50, 400: max abs error
// 2.5e-7 (libm sinf: 6.9e-8, __sinf: 3.9e-7 / 3.4e-6 / 2.8e-5) at 2.8x the throughput of sinf.
__device__ __forceinline__ float sin_rev(float x) {
    constexpr float kInv2Pi = 0.15915494309189535f;
    constexpr float kInv2PiLo = (float)(0.15915494309189535 - (double)kInv2Pi);
    const float k = rintf(x * kInv2Pi);
    float f = fmaf(x, kInv2Pi, -k);
    f = fmaf(x, kInv2PiLo, f);
    return __builtin_amdgcn_sinf(f);
}

constexpr int ACT_SNAKE_LIBM = 17;  // SnakeBeta with libm's sinf (AFTER_SNAKE_LIBM=1: A/B switch)

__device__ __forceinline__ float act_apply(float v, int act, float pa, float pb) {
    switch (act) {
        case ACT_SNAKE: {
            const float s = sin_rev(v * pa);
            return v + pb * (s * s);
        }
        case ACT_SNAKE_LIBM: {
            const float s = sinf(v * pa);
            return v + pb * (s * s);
        }
        case ACT_SILU:
            return v / (1.0f + expf(-v));
        case ACT_RELU:
            return fmaxf(v, 0.f);
        case ACT_TANH:
            return tanhf(v);
        case ACT_LRELU:
            return v > 0.f ? v : 0.2f * v;
        default:
            return v;
    }
}

// GroupNorm statistics -> (mean, rstd) per group, once per block.  The words of the (sub-slot, group) pairs -- 64
// contiguous bytes each -- are fetched ONCE per block, 16 bytes per thread, into LDS; one lane per group then adds the
// sub-slots (integers: exact) and does the fp64 arithmetic.  (Round 4's first form had every group lane issue its 64
// eight-byte loads itself: 512 line requests per block on the same few lines of the same L2 channels -- with several
// hundred blocks per launch those channels were the bottleneck: +3-6 us per 12-MB launch.)  Ends with a barrier.
__device__ __forceinline__ void gn_mean_rstd(const double* stats, int b, int G, int sub_stride, int C, int stat_T, float eps,
                                             long long* swl /* [16 * kStatSub * kStatWords] LDS */, float* gmean,
                                             float* grstd, int nthreads) {
    static_assert(kStatWords % 2 == 0, "16-byte pieces");
    constexpr int PPG = kStatWords / 2;  // 16-byte pieces per (sub-slot, group)
    const long long* sbase = reinterpret_cast<const long long*>(stats) + (size_t)b * G * kStatWords;
    for (int i = threadIdx.x; i < G * kStatSub * PPG; i += nthreads) {
        const int piece = i % PPG, u = (i / PPG) % kStatSub, gI = i / (PPG * kStatSub);
        const longlong2 w2 = *reinterpret_cast<const longlong2*>(sbase + (size_t)u * sub_stride + gI * kStatWords + 2 * piece);
        *reinterpret_cast<longlong2*>(&swl[(gI * kStatSub + u) * kStatWords + 2 * piece]) = w2;
    }
    __syncthreads();
    if ((int)threadIdx.x < G) {
        const int gI = threadIdx.x;
        const double n = (double)(C / G) * stat_T;
        // the sub-slots' words add as integers (exact), then each quantity folds into one fp64 (conv.h: stat_bins)
        long long ws[kStatBins], wq[kStatBins];
#pragma unroll
        for (int k = 0; k < kStatBins; ++k) ws[k] = wq[k] = 0;
#pragma unroll
        for (int u = 0; u < kStatSub; ++u)
#pragma unroll
            for (int k = 0; k < kStatBins; ++k) {
                ws[k] += swl[(gI * kStatSub + u) * kStatWords + k];
                wq[k] += swl[(gI * kStatSub + u) * kStatWords + kStatBins + k];
            }
        const double sm = stat_bins_total(ws), qq = stat_bins_total(wq);
        const double mean = sm / n;
        double var = qq / n - mean * mean;
        var = var < 0 ? 0 : var;
        const bool blown = !(fabs(sm) < kStatBlown) || !(qq < kStatBlown);  // a non-finite (saturated) partial sum went in: conv.h
        gmean[gI] = blown ? __builtin_nanf("") : (float)mean;
        grstd[gI] = blown ? __builtin_nanf("") : (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
}

// ------------------------------------------------------------------ activate + halo
struct ActTmArgs {
    const float* x;       // [B][T][ldx] time-major, or [B][C][T] when x_cm
    float* y;             // [B][Tp][Cp]
    unsigned short* y3;   // instead of y: bf16 planes, x6 blocks of [B x rows16][Cp] (conv_x6.hip)
    float hscale;         // != 0: two fp16 pieces of y x hscale in h3 blocks instead (gemm_h3_pipe.h)
    int rows16;
    const double* stats;  // [sub-slot][B][G][2][kStatBins] 64-bit words (conv.h: stat_bins) or nullptr
    const float* gamma;   // with stats: GroupNorm weight; without: per-channel scale (or nullptr)
    const float* beta;
    const float* act_a;
    const float* act_b;
    const float* state;   // streaming: [B][HALO][Cp] activated rows preceding this chunk, or nullptr
    float* state_out;     // streaming: the context of the NEXT chunk = rows [T, T + HALO) of the padded tensor (another buffer than
                          // `state`: other workgroups still read that one), or nullptr
    const float* scale_b; // optional per-(b, channel) affine [B][C] applied instead of gamma / beta
    const float* shift_b;
    const float* x2;      // optional second time-major input added to x first (Res2Net: x_i + y_{i-1})
    int ldx2;
    int act, C, Cp, T, Tp, G, x_cm, ldx, rows_per_block, pad_reflect, sub_stride, xcd_rows, stat_T;
    float eps;
};

// One thread = one 4-channel quad; a block walks `rows_per_block` consecutive rows of the
// padded buffer, U rows in flight per thread.  Everything the kernel needs -- the first rows, the
// per-channel parameters, the statistics sub-slots -- is requested up front so that the block pays
// ONE exposed memory latency (these launches are 6-25 MB: latency, not bandwidth, is the cost).
__global__ __launch_bounds__(256) void act_pad_tm_kernel(ActTmArgs a) {
    const int Q = a.Cp >> 2;
    const int R = 256 / Q;  // rows per pass (Q <= 256)
    const int r = threadIdx.x / Q, q = threadIdx.x - r * Q;
    const int b = blockIdx.y, c0 = 4 * q;
    const bool active = r < R;
    __shared__ float gmean[16], grstd[16];
    constexpr int U = 4;
    // Block id i runs on XCD i % 8 (private L2s).  With xcd_rows the row blocks are dealt so that XCD j
    // owns the j-th eighth of the rows -- the same rows XCD j wrote as the previous conv's output and
    // will read as the next conv's input (conv_tm's pm = 8 tile map): producer -> consumer traffic
    // stays inside one XCD's L2 instead of crossing the fabric.
    int rb = blockIdx.x;
    if (a.xcd_rows) {
        const int per = gridDim.x >> 3;  // gridDim.x is a multiple of 8
        rb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    }
    const int p_lo = rb * a.rows_per_block;
    if (p_lo >= a.Tp) return;
    const int p_hi = min(p_lo + a.rows_per_block, a.Tp);
    float* yb = a.y + (size_t)b * a.Tp * a.Cp;
    const bool vec = !a.x_cm && c0 + 3 < a.C;
    f32x4 v[U];
    int tt[U];
    auto load_rows = [&](int pb0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = pb0 + u * R;
            int t = p - HALO;
            if (a.pad_reflect && (t < 0 || t >= a.T) && t > -a.T && t < 2 * a.T - 1)
                t = t < 0 ? -t : 2 * (a.T - 1) - t;  // 'reflect' padding (TDNNBlock, ecapa_encoder.py:85-139)
            tt[u] = t;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p >= p_hi) continue;
            if (t >= 0 && t < a.T) {
                if (vec) {
                    v[u] = *reinterpret_cast<const f32x4*>(a.x + ((size_t)b * a.T + t) * a.ldx + c0);
                    if (a.x2) v[u] += *reinterpret_cast<const f32x4*>(a.x2 + ((size_t)b * a.T + t) * a.ldx2 + c0);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c0 + k < a.C)
                            v[u][k] = a.x_cm ? a.x[((size_t)b * a.C + c0 + k) * a.T + t]
                                             : a.x[((size_t)b * a.T + t) * a.ldx + c0 + k] +
                                                   (a.x2 ? a.x2[((size_t)b * a.T + t) * a.ldx2 + c0 + k] : 0.f);
                }
            } else if (t < 0 && a.state) {
                v[u] = *reinterpret_cast<const f32x4*>(a.state + ((size_t)b * HALO + (HALO + t)) * a.Cp + c0);
            }
        }
    };
    if (active) load_rows(p_lo + r);
    // raw per-channel parameters (independent of the statistics)
    float ga[4], be[4], pa[4], pb[4];
    if (active && c0 + 3 < a.C && (a.C & 3) == 0) {  // four whole channels: one 16-byte request per parameter instead of four scalar ones
        auto ld4 = [&](const float* p, float (&o)[4], float dflt) {
            const f32x4 v4 = p ? *reinterpret_cast<const f32x4*>(p + c0) : f32x4{dflt, dflt, dflt, dflt};
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = v4[k];
        };
        const bool per_clip = a.scale_b && !a.stats;
        ld4(per_clip ? a.scale_b + (size_t)b * a.C : a.gamma, ga, 1.f);
        ld4(per_clip ? a.shift_b + (size_t)b * a.C : (a.gamma ? a.beta : nullptr), be, 0.f);
        ld4(a.act_a, pa, 0.f);
        ld4(a.act_b, pb, 0.f);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + k;
            ga[k] = 1.f;
            be[k] = 0.f;
            pa[k] = pb[k] = 0.f;
            if (!active || c >= a.C) continue;
            if (a.scale_b && !a.stats) {
                ga[k] = a.scale_b[(size_t)b * a.C + c];
                be[k] = a.shift_b[(size_t)b * a.C + c];
            } else if (a.gamma) {
                ga[k] = a.gamma[c];
                be[k] = a.beta[c];
            }
            if (a.act_a) pa[k] = a.act_a[c];
            if (a.act_b) pb[k] = a.act_b[c];
        }
    }
    if (a.stats) {
        __shared__ __attribute__((aligned(16))) long long swl[16 * kStatSub * kStatWords];
        gn_mean_rstd(a.stats, b, a.G, a.sub_stride, a.C, a.stat_T, a.eps, swl, gmean, grstd, 256);
    }
    if (!active) return;
    float sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc[k] = ga[k];
        sh[k] = be[k];
        if (a.stats && c0 + k < a.C) {
            const int g = (c0 + k) / (a.C / a.G);
            sc[k] = grstd[g] * ga[k];
            sh[k] = be[k] - gmean[g] * sc[k];
        }
    }
    for (int pb0 = p_lo + r; pb0 < p_hi; pb0 += U * R) {
        f32x4 o[U];
        int to[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            o[u] = v[u];
            to[u] = tt[u];
        }
        if (pb0 + U * R < p_hi) load_rows(pb0 + U * R);  // next rows in flight behind this batch's math
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = pb0 + u * R;
            if (p >= p_hi) continue;
            if (to[u] >= 0 && to[u] < a.T) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o[u][k] = c0 + k < a.C ? act_apply(o[u][k] * sc[k] + sh[k], a.act, pa[k], pb[k]) : 0.f;
            }
            *reinterpret_cast<f32x4*>(yb + (size_t)p * a.Cp + c0) = o[u];
            if (a.state_out && p >= a.T && p < a.T + HALO)
                *reinterpret_cast<f32x4*>(a.state_out + ((size_t)b * HALO + (p - a.T)) * a.Cp + c0) = o[u];
        }
    }
}

// streaming: state[b][j][:] = ypad[b][T + j][:], j < HALO (the last HALO activated rows; the old
// context included when the chunk is shorter than the halo)
__global__ void state_update_tm_kernel(const float* __restrict__ yp, float* __restrict__ state, int Cp,
                                       int T, int Tp, int total4) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int per = HALO * Cp / 4;
    const int b = idx / per, e = idx - b * per;
    reinterpret_cast<f32x4*>(state)[idx] =
        reinterpret_cast<const f32x4*>(yp + ((size_t)b * Tp + T) * Cp)[e];
}

// The same pass with the output as bf16 planes in x6 blocks (common.h) of [B x rows16][Cp] -- the A operand of
// conv_x6.hip.  One wave = one (16-row, 32-channel) block at a time: lane l holds row l / 4, channels 8 (l % 4) .. + 7
// of it, so each of the three planes of the block leaves as ONE 1-KB contiguous store instruction (16 bytes per lane at
// the block's own chunk permutation); a wave walks RB consecutive row blocks of its channel block, all of their rows
// requested up front.  Whole-clip passes of time-major tensors only (no streaming context, no second input).
__global__ __launch_bounds__(256) void act_pad_x6_kernel(ActTmArgs a, int nkb, int RB) {
    __shared__ float gmean[16], grstd[16];
    __shared__ __attribute__((aligned(16))) long long swl[16 * kStatSub * kStatWords];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // waves are dealt over (row group, channel block) pairs, channel blocks fastest: no idle waves whatever Cp / 32 is
    const int nrb = a.rows16 >> 4;
    // block id i runs on XCD i % 8: with xcd_rows XCD j takes the j-th eighth of the (row group, channel block) pairs --
    // the rows XCD j's conv tiles wrote and will read (act_pad_tm_kernel, conv_x6's pm = 8 tile map)
    int bid = blockIdx.x;
    if (a.xcd_rows) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int wv = bid * 4 + w;
    const int kb = wv % nkb, rg = wv / nkb;
    const int r = lane >> 2, c0 = kb * 32 + 8 * (lane & 3);
    constexpr int MAXRB = 4;
    f32x4 v[MAXRB][2];
    const bool kvalid = rg * RB < nrb;
#pragma unroll
    for (int u = 0; u < MAXRB; ++u) {
        v[u][0] = v[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int rbI = rg * RB + u, t = rbI * 16 + r - HALO;
        if (u < RB && kvalid && rbI < nrb && t >= 0 && t < a.T) {
            const float* xr = a.x + ((size_t)b * a.T + t) * a.ldx + c0;
            if (c0 + 7 < a.C) {
                v[u][0] = *reinterpret_cast<const f32x4*>(xr);
                v[u][1] = *reinterpret_cast<const f32x4*>(xr + 4);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c0 + k < a.C) v[u][k >> 2][k & 3] = xr[k];
            }
        }
    }
    float ga[8], be[8], pa[8], pb[8];
    if (kvalid && c0 + 7 < a.C && ((a.C | c0) & 3) == 0) {  // eight whole channels: four 32-byte requests instead of 32 scalar ones
        auto ld8 = [&](const float* p, float (&o)[8], float dflt) {
            if (p) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + c0), v1 = *reinterpret_cast<const f32x4*>(p + c0 + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[k] = v0[k];
                    o[4 + k] = v1[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = dflt;
            }
        };
        ld8(a.gamma, ga, 1.f);
        ld8(a.gamma ? a.beta : nullptr, be, 0.f);
        ld8(a.act_a, pa, 0.f);
        ld8(a.act_b, pb, 0.f);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k;
            ga[k] = 1.f;
            be[k] = 0.f;
            pa[k] = pb[k] = 0.f;
            if (!kvalid || c >= a.C) continue;
            if (a.gamma) {
                ga[k] = a.gamma[c];
                be[k] = a.beta[c];
            }
            if (a.act_a) pa[k] = a.act_a[c];
            if (a.act_b) pb[k] = a.act_b[c];
        }
    }
    if (a.stats) gn_mean_rstd(a.stats, b, a.G, a.sub_stride, a.C, a.stat_T, a.eps, swl, gmean, grstd, 256);
    if (!kvalid) return;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = ga[k];
        sh[k] = be[k];
        if (a.stats && c0 + k < a.C) {
            const int g = (c0 + k) / (a.C / a.G);
            sc[k] = grstd[g] * ga[k];
            sh[k] = be[k] - gmean[g] * sc[k];
        }
    }
    const int slot = (lane & 3) ^ ((0x78 >> (2 * ((r >> 2) & 3))) & 3);  // x6 chunk permutation (common.h: x6_offset)
#pragma unroll
    for (int u = 0; u < MAXRB; ++u) {
        const int rbI = rg * RB + u, t = rbI * 16 + r - HALO;
        if (u >= RB || rbI >= nrb) continue;
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xv = v[u][k >> 2][k & 3];
            o[k] = (t >= 0 && t < a.T && c0 + k < a.C) ? act_apply(xv * sc[k] + sh[k], a.act, pa[k], pb[k]) : 0.f;
        }
        if (a.hscale != 0.f) {  // (wave-uniform) two fp16 pieces, two planes per block
            uint2 h0, l0, h1, l1;
            const float hs = a.hscale;
            h3_split4(o[0] * hs, o[1] * hs, o[2] * hs, o[3] * hs, h0, l0);
            h3_split4(o[4] * hs, o[5] * hs, o[6] * hs, o[7] * hs, h1, l1);
            unsigned short* bp = a.y3 + ((((size_t)b * nrb + rbI) * nkb + kb) * 2) * 512 + r * 32 + slot * 8;
            *reinterpret_cast<uint4*>(bp) = uint4{h0.x, h0.y, h1.x, h1.y};
            *reinterpret_cast<uint4*>(bp + 512) = uint4{l0.x, l0.y, l1.x, l1.y};
            continue;
        }
        uint2 h0, m0, l0, h1, m1, l1;
        x6_split4(o[0], o[1], o[2], o[3], h0, m0, l0);
        x6_split4(o[4], o[5], o[6], o[7], h1, m1, l1);
        unsigned short* bp = a.y3 + ((((size_t)b * nrb + rbI) * nkb + kb) * 3) * 512 + r * 32 + slot * 8;
        *reinterpret_cast<uint4*>(bp) = uint4{h0.x, h0.y, h1.x, h1.y};
        *reinterpret_cast<uint4*>(bp + 512) = uint4{m0.x, m0.y, m1.x, m1.y};
        *reinterpret_cast<uint4*>(bp + 1024) = uint4{l0.x, l0.y, l1.x, l1.y};
    }
}

// The same pass for LARGE tensors (a batch of clips): a wave walks RBL row blocks of its channel block two at a time, the next pair's rows
// requested before the current pair is activated, split and stored.  In act_pad_x6_kernel every wave of the launch requests its rows,
// then activates (SnakeBeta + the three-way split: ~40 VALU operations per element), then stores -- all at the same time: the memory
// system idles while the VALUs work and vice versa (384 channels x 4096 frames x 8 clips: 126 MB in 76 us = 1.7 TB/s).  Waves that
// loop drift apart, and loads, arithmetic and stores of different waves overlap.
__global__ __launch_bounds__(256) void act_pad_x6_loop_kernel(ActTmArgs a, int nkb, int RBL) {
    __shared__ float gmean[16], grstd[16];
    __shared__ __attribute__((aligned(16))) long long swl[16 * kStatSub * kStatWords];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nrb = a.rows16 >> 4;
    const int wv = blockIdx.x * 4 + w;
    const int kb = wv % nkb, rg = wv / nkb;
    const int r = lane >> 2, c0 = kb * 32 + 8 * (lane & 3);
    const int rb_lo = rg * RBL, rb_hi = min(rb_lo + RBL, nrb);
    const bool kvalid = rb_lo < nrb;
    const bool whole = c0 + 7 < a.C;
    auto load2 = [&](f32x4 (&v)[2][2], int rb0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            v[u][0] = v[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int rbI = rb0 + u, t = rbI * 16 + r - HALO;
            if (rbI < rb_hi && t >= 0 && t < a.T) {
                const float* xr = a.x + ((size_t)b * a.T + t) * a.ldx + c0;
                if (whole) {
                    v[u][0] = *reinterpret_cast<const f32x4*>(xr);
                    v[u][1] = *reinterpret_cast<const f32x4*>(xr + 4);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (c0 + k < a.C) v[u][k >> 2][k & 3] = xr[k];
                }
            }
        }
    };
    f32x4 va[2][2], vb[2][2];
    if (kvalid) load2(va, rb_lo);
    float ga[8], be[8], pa[8], pb[8];
    if (kvalid && whole && ((a.C | c0) & 3) == 0) {
        auto ld8 = [&](const float* p, float (&o)[8], float dflt) {
            if (p) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + c0), v1 = *reinterpret_cast<const f32x4*>(p + c0 + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[k] = v0[k];
                    o[4 + k] = v1[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = dflt;
            }
        };
        ld8(a.gamma, ga, 1.f);
        ld8(a.gamma ? a.beta : nullptr, be, 0.f);
        ld8(a.act_a, pa, 0.f);
        ld8(a.act_b, pb, 0.f);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k;
            ga[k] = 1.f;
            be[k] = 0.f;
            pa[k] = pb[k] = 0.f;
            if (!kvalid || c >= a.C) continue;
            if (a.gamma) {
                ga[k] = a.gamma[c];
                be[k] = a.beta[c];
            }
            if (a.act_a) pa[k] = a.act_a[c];
            if (a.act_b) pb[k] = a.act_b[c];
        }
    }
    if (a.stats) gn_mean_rstd(a.stats, b, a.G, a.sub_stride, a.C, a.stat_T, a.eps, swl, gmean, grstd, 256);
    if (!kvalid) return;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = ga[k];
        sh[k] = be[k];
        if (a.stats && c0 + k < a.C) {
            const int g = (c0 + k) / (a.C / a.G);
            sc[k] = grstd[g] * ga[k];
            sh[k] = be[k] - gmean[g] * sc[k];
        }
    }
    const int slot = (lane & 3) ^ ((0x78 >> (2 * ((r >> 2) & 3))) & 3);  // x6 chunk permutation (common.h: x6_offset)
    auto emit2 = [&](const f32x4 (&v)[2][2], int rb0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rbI = rb0 + u, t = rbI * 16 + r - HALO;
            if (rbI >= rb_hi) continue;
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xv = v[u][k >> 2][k & 3];
                o[k] = (t >= 0 && t < a.T && c0 + k < a.C) ? act_apply(xv * sc[k] + sh[k], a.act, pa[k], pb[k]) : 0.f;
            }
            if (a.hscale != 0.f) {  // (wave-uniform) two fp16 pieces, two planes per block
                uint2 h0, l0, h1, l1;
                const float hs = a.hscale;
                h3_split4(o[0] * hs, o[1] * hs, o[2] * hs, o[3] * hs, h0, l0);
                h3_split4(o[4] * hs, o[5] * hs, o[6] * hs, o[7] * hs, h1, l1);
                unsigned short* bp = a.y3 + ((((size_t)b * nrb + rbI) * nkb + kb) * 2) * 512 + r * 32 + slot * 8;
                *reinterpret_cast<uint4*>(bp) = uint4{h0.x, h0.y, h1.x, h1.y};
                *reinterpret_cast<uint4*>(bp + 512) = uint4{l0.x, l0.y, l1.x, l1.y};
                continue;
            }
            uint2 h0, m0, l0, h1, m1, l1;
            x6_split4(o[0], o[1], o[2], o[3], h0, m0, l0);
            x6_split4(o[4], o[5], o[6], o[7], h1, m1, l1);
            unsigned short* bp = a.y3 + ((((size_t)b * nrb + rbI) * nkb + kb) * 3) * 512 + r * 32 + slot * 8;
            *reinterpret_cast<uint4*>(bp) = uint4{h0.x, h0.y, h1.x, h1.y};
            *reinterpret_cast<uint4*>(bp + 512) = uint4{m0.x, m0.y, m1.x, m1.y};
            *reinterpret_cast<uint4*>(bp + 1024) = uint4{l0.x, l0.y, l1.x, l1.y};
        }
    };
    for (int rb0 = rb_lo; rb0 < rb_hi; rb0 += 4) {
        if (rb0 + 2 < rb_hi) load2(vb, rb0 + 2);
        emit2(va, rb0);
        if (rb0 + 2 >= rb_hi) break;
        if (rb0 + 4 < rb_hi) load2(va, rb0 + 4);
        emit2(vb, rb0 + 2);
    }
}

// ------------------------------------------------------------------ the conv GEMM
struct ConvTmArgs {
    const float* xp;    // [B][Tp][Cp]
    const float* w;     // [phases][Cout][K]
    const float* bias;  // [Cout] or nullptr
    const float* res;   // [B][Tout][Cout] or nullptr
    float* y;           // [B][Tout][Cout]  (y_cm: [B][Cout][Tout])
    double* stats;      // [kStatSub][sub_stride] words with [B][G][2][kStatBins] inside: accumulators of y, or nullptr
    int Cp, Cout, Tp, Tout, K, phases, lda, ostride, Nn, G, y_cm, sub_stride;
    int magic, extra;   // tap(s) = (s * magic) >> 16,  extra = (dil - 1) * Cp
    int abase[kMaxPhases];  // (HALO + toff[ph][0]) * Cp
    int ooff[kMaxPhases];
    int gs_off;         // float offset of the statistics scratch inside the dynamic LDS
    int gs_in_ring;     // 1: the scratch overlays the (then idle) ring
    // post-activation epilogue of the TDNN / FiLM layers: y = out_act(acc + bias) * post_scale + post_shift
    const float* post_scale;
    const float* post_shift;
    int out_act, post_bstride, bias_bstride;
    // generalised addressing (channel-sliced views of wider tensors, the reference's [B][C][T] at
    // the API edges): element (b, row, col) of the input sits at xp[b * x_bs + row * x_ld + col]
    long long x_bs, y_bs, res_bs;
    int x_ld, y_ld, y_coff, res_ld, res_coff, res_cm;
    // second output: the NEXT conv's activated, haloed input written by this epilogue (no act_pad
    // launch in between) -- y2[b][HALO + row][y2_coff + col - y2_clo] = act2((v + add) * sc + sh) for
    // the output channels col in [y2_clo, y2_chi).  Halo: zeros, or mirrored rows (TDNN 'reflect').
    float* y2;
    const float* y2_scale;  // [Cout] or nullptr
    const float* y2_shift;
    const float* y2_pa;     // snake alpha / 1 / (beta + eps), indexed by output channel
    const float* y2_pb;
    const float* y2_add;    // [B][Tout][y2_add_ld] tensor whose columns y2_add_coff + (col - y2_clo) are added first
    long long y2_bs, y2_add_bs;
    int y2_ld, y2_coff, y2_clo, y2_chi, y2_act, y2_add_ld, y2_add_coff, y2_reflect, y2_zero_halo, tiles_m_last;
};

template <int MB, int NB, int KS, int NS, int RS, bool DIL, bool Y2>
__global__ __launch_bounds__(128 * KS * RS) void conv_tm_kernel(ConvTmArgs g, int tiles_m, int tiles_n,
                                                                int xcd_pm, int ny) {
    constexpr int BK = 32, CPR = 8, RPP = 8, KK = 2;
    static_assert(MB % RS == 0, "row parts must divide the tile's block rows");
    constexpr int MT = MB / RS, NT = NB;
    constexpr int BM = 16 * MB, BN = 32 * NB;
    constexpr int NW = 2 * KS * RS;
    constexpr int ROWS = KS * (BM + BN);
    constexpr int LPS = ROWS / RPP / NW;
    static_assert(ROWS % (RPP * NW) == 0, "ring slot must split evenly over the waves");
    constexpr int STAGE = ROWS * BK;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // ---- workgroup -> (clip / phase, tile).  Block id b runs on XCD b % 8 (private L2s): with
    // nwg % 8 == 0 the XCDs are laid out as a pm x pn grid over each clip's tile matrix and walk
    // the clips in the same order, otherwise XCDs take contiguous ranges of the linear order.
    const int nwg = tiles_m * tiles_n;
    int tm, tn, yi;
    if (xcd_pm > 0) {
        const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
        const int per = nwg >> 3;
        yi = li / per;
        const int l2 = li - yi * per;
        const int pn = 8 / xcd_pm;
        const int cm = tiles_m / xcd_pm, cn = tiles_n / pn;
        const int xi = xcd % xcd_pm, xj = xcd / xcd_pm;
        (void)cn;
        tm = xi * cm + l2 % cm;
        tn = xj * cn + l2 / cm;
    } else {
        const int total = nwg * ny;
        int bid = blockIdx.x;
        const int xcd = bid & 7, q = total >> 3, r = total & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        yi = bid / nwg;
        const int t = bid - yi * nwg;
        tn = t / tiles_m;
        tm = t - tn * tiles_m;
    }
    const int b = yi / g.phases, ph = yi - b * g.phases;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wid % KS, rp = (wid / KS) % RS, part = wid / (KS * RS);
    const int M = g.Nn, N = g.Cout, Kh = g.K / KS;
    const int nk = Kh / BK;

    // ---- DMA addressing: wave-uniform 64-bit base + per-lane 32-bit byte offset (+ a scalar
    // slab offset).  A pieces: slab S = half * nk + s of the flattened (tap, channel) axis.
    const float* xb = g.xp + (size_t)b * g.x_bs + g.abase[ph];
    const float* wb = g.w + (size_t)ph * g.Cout * g.K;
    const int rsub = lane / CPR, pos = lane % CPR;
    unsigned voff[LPS];
    const float* sbase[LPS];
    int kb[LPS], ex[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const int row0 = (wid * LPS + i) * RPP;
        const int row = row0 + rsub;
        if (row0 < KS * BM) {
            const int half = row0 / BM;
            const int gm = min(m0 + row - half * BM, M - 1);
            sbase[i] = xb;
            kb[i] = half * nk;
            ex[i] = g.extra;
            voff[i] = ((unsigned)gm * (unsigned)g.lda + (unsigned)((pos ^ (row & (CPR - 1))) * 4)) * 4u;
        } else {
            const int rr0 = row0 - KS * BM;
            const int half = rr0 / BN;
            const int gn = min(n0 + (row - KS * BM) - half * BN, N - 1);
            sbase[i] = wb + half * Kh;
            kb[i] = 0;
            ex[i] = 0;
            voff[i] = ((unsigned)gn * (unsigned)g.K + (unsigned)((pos ^ (row & (CPR - 1))) * 4)) * 4u;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int magic = g.magic;
#define AFTER_BAL_DMA(w_, slab_, slot_)                                                              \
    {                                                                                                \
        const int S__ = kb[w_] + (slab_);                                                            \
        int o__ = S__ * BK;                                                                          \
        if constexpr (DIL) o__ += ((S__ * magic) >> 16) * ex[w_];                                    \
        lds_dma16(lds0 + (unsigned)(((slot_) * STAGE + (wid * LPS + (w_)) * RPP * BK) * 4), voff[w_],  \
                  (unsigned long long)(uintptr_t)(sbase[w_] + o__)); /* (M0 is the compiler's: gemm_pipe.h) */ \
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // statistics scratch behind the ring: zeroed here, long before the epilogue (the main loop's
    // barriers order it); inside the ring (gs_in_ring) it is zeroed after the last ring read
    // [2][NW][NT][SL] partial sums (SL = 4 channel quads of a 16-column block, or its 16 columns when a quad may
    // straddle two groups): every (wave, column block) owns its slots, tid < 16 adds them per group in a fixed
    // order -- no LDS atomics, so the tile's statistics do not depend on the order the waves finish in
    // round 4: the per-wave float slots and the one-lane-per-group serial loop over them (96 dependent LDS reads:
    // 6-11 us per tile, measured on conv_x6's copy of it) became binned INTEGER accumulators in LDS -- every lane quad's
    // partial sum goes in with ds_add_u64 (exact, order-independent: conv.h stat_bins_add), the non-zero words go on to
    // the global accumulators
    long long* lbins = reinterpret_cast<long long*>(smem + g.gs_off);  // [groups of this tile <= 16][kStatWords]
    const int Cg_ = g.stats ? g.Cout / g.G : 1;
    const int sg0 = g.stats ? n0 / Cg_ : 0;
    const int sng = g.stats ? (min(n0 + BN, g.Cout) - 1) / Cg_ - sg0 + 1 : 0;
    if (g.stats && !g.gs_in_ring)
        for (int i = tid; i < sng * kStatWords; i += 128 * KS * RS) lbins[i] = 0;

    unsigned long long ph_fence = 0, ph_vm = 0, ph_bar = 0;  // cycle-stamp sinks of the pipeline macros
    (void)ph_fence;
    (void)ph_vm;
    (void)ph_bar;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < nk) {
#pragma unroll
            for (int i = 0; i < LPS; ++i) AFTER_BAL_DMA(i, s, s)
        }

    const int frow = lane & 15, kq = lane >> 4, sw = frow & (CPR - 1);
    const int aoff = (kh * BM + rp * (BM / RS) + frow) * BK;
    const int woff = (KS * BM + kh * BN + part * 16 * NB + frow) * BK;
    f32x4 fa[KK][2][MT], fb[KK][2][NT];
    unsigned a_c[KK], w_c[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const int c = ((kq + 4 * kk) ^ sw) * 4;
        a_c[kk] = lds0 + (aoff + c) * 4;
        w_c[kk] = lds0 + (woff + c) * 4;
    }
    // the pipeline macros test `g.dbg` for their optional cycle stamps (gemm.hip's timeline
    // diagnostics): inside the main loop `g` names a constant-null stand-in, so they fold away
    struct DbgNull {
        unsigned long long* dbg = nullptr;
    };
    const DbgNull conv_tm_dbg_null__{};
#define g conv_tm_dbg_null__
    AFTER_GEMM_WAIT_SLAB(0, NS - 1)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    AFTER_GEMM_LOAD_FRAGS(0, 0)
    int kt = 0;
    for (; kt + 1 + NS < nk; kt += 2) {
        AFTER_GEMM_STEP_IL(0, 1, kt, true)
        AFTER_GEMM_STEP_IL(1, 0, kt + 1, true)
    }
    for (; kt < nk; kt += 2) {
        AFTER_GEMM_STEP_IL(0, 1, kt, false)
        if (kt + 1 < nk) AFTER_GEMM_STEP_IL(1, 0, kt + 1, false)
    }
#undef g

    // ---- split-K reduction through LDS in k-part order (bit-deterministic), as in gemm.hip
    float* red = smem;
    if (KS > 1 || (g.stats && g.gs_in_ring)) __syncthreads();  // every wave is past its last ring read
    if (g.stats && g.gs_in_ring)
        for (int i = tid; i < sng * kStatWords; i += 128 * KS * RS) lbins[i] = 0;
    if constexpr (KS > 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                *reinterpret_cast<f32x4*>(red + ((wid * MT * NT + i * NT + j) * 64 + lane) * 4) = acc[i][j];
    }
    if (KS > 1 || (g.stats && g.gs_in_ring)) __syncthreads();

    // accumulator layout (W fragment as srcA): lane l holds C[row = l & 15][col = 4 (l >> 4) + r]
    const int crow = lane & 15, ccol0 = 4 * (lane >> 4);
    const int Cg = g.stats ? g.Cout / g.G : 1;
    const bool quad = (Cg & 3) == 0;  // a lane's four channels sit in one group (every shipped width)
    float* yb = g.y ? g.y + (size_t)b * g.y_bs + g.y_coff : nullptr;
    const float* rb = g.res ? g.res + (size_t)b * g.res_bs + g.res_coff : nullptr;
    float* y2b = (Y2 && g.y2) ? g.y2 + (size_t)b * g.y2_bs + g.y2_coff : nullptr;
    const float* a2b = (Y2 && g.y2_add) ? g.y2_add + (size_t)b * g.y2_add_bs + g.y2_add_coff : nullptr;
    const bool vec_ok = !g.y_cm && !g.res_cm && ((g.y_ld | g.y_coff) & 3) == 0 &&
                        (!g.res || ((g.res_ld | g.res_coff) & 3) == 0);
    const bool vec2_ok = ((g.y2_ld | g.y2_coff | g.y2_clo | g.y2_add_ld | g.y2_add_coff) & 3) == 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        // keep the column blocks' parameter loads from being hoisted together: the epilogue must not
        // need more registers than the main loop (64 x 96 tiles: 128 VGPRs = 4 waves per SIMD)
        __builtin_amdgcn_sched_barrier(0);
        const int gn = n0 + part * 16 * NB + j * 16 + ccol0;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, ps = {1.f, 1.f, 1.f, 1.f}, pt = {0.f, 0.f, 0.f, 0.f};
        f32x4 s2 = {1.f, 1.f, 1.f, 1.f}, t2 = {0.f, 0.f, 0.f, 0.f}, pa2 = t2, pb2 = t2;
        const bool in2 = Y2 && y2b && gn >= g.y2_clo && gn < g.y2_chi;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (gn + r < N) {
                if (g.bias) bv[r] = g.bias[(size_t)b * g.bias_bstride + gn + r];
                if (g.post_scale) {
                    ps[r] = g.post_scale[(size_t)b * g.post_bstride + gn + r];
                    pt[r] = g.post_shift[(size_t)b * g.post_bstride + gn + r];
                }
                if (in2) {
                    if (g.y2_scale) {
                        s2[r] = g.y2_scale[gn + r];
                        t2[r] = g.y2_shift[gn + r];
                    }
                    if (g.y2_pa) pa2[r] = g.y2_pa[gn + r];
                    if (g.y2_pb) pb2[r] = g.y2_pb[gn + r];
                }
            }
        f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, qsum = {0.f, 0.f, 0.f, 0.f};  // per column of the lane's quad
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if ((i * NT + j) % KS != kh) continue;
            f32x4 o = acc[i][j];
            if constexpr (KS > 1) {
                const int w0 = (part * RS + rp) * KS;
                o = *reinterpret_cast<const f32x4*>(red + ((w0 * MT * NT + i * NT + j) * 64 + lane) * 4);
#pragma unroll
                for (int q = 1; q < KS; ++q)
                    o += *reinterpret_cast<const f32x4*>(red + (((w0 + q) * MT * NT + i * NT + j) * 64 + lane) * 4);
            }
            const int gm = m0 + rp * (BM / RS) + i * 16 + crow;
            if (gm >= M || gn >= N) continue;
            o += bv;
            if (g.out_act == ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
            } else if (g.out_act == ACT_TANH) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = tanhf(o[r]);
            } else if (g.out_act == ACT_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = o[r] / (1.0f + expf(-o[r]));
            }
            if (g.post_scale) o = o * ps + pt;
            const int trow = gm * g.ostride + g.ooff[ph];
            if (trow >= g.Tout) continue;
            const bool full4 = gn + 3 < N;
            // residual
            if (rb) {
                if (vec_ok && full4) {
                    o += *reinterpret_cast<const f32x4*>(rb + (size_t)trow * g.res_ld + gn);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (gn + r < N)
                            o[r] += g.res_cm ? rb[(size_t)(gn + r) * g.Tout + trow] : rb[(size_t)trow * g.res_ld + gn + r];
                }
            }
            // primary output
            if (yb) {
                if (vec_ok && full4) {
                    *reinterpret_cast<f32x4*>(yb + (size_t)trow * g.y_ld + gn) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (gn + r < N) {
                            if (g.y_cm) yb[(size_t)(gn + r) * g.Tout + trow] = o[r];
                            else yb[(size_t)trow * g.y_ld + gn + r] = o[r];
                        }
                }
            }
            // statistics of the primary output
            if (g.stats) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (gn + r < N) {
                        ssum[r] += o[r];
                        qsum[r] += o[r] * o[r];
                    }
            }
            // the next conv's activated, haloed input
            if (in2) {
                f32x4 v2 = o;
                const int c2 = gn - g.y2_clo;
                if (a2b) {
                    if (vec2_ok && full4) v2 += *reinterpret_cast<const f32x4*>(a2b + (size_t)trow * g.y2_add_ld + c2);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (gn + r < g.y2_chi) v2[r] += a2b[(size_t)trow * g.y2_add_ld + c2 + r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v2[r] = act_apply(v2[r] * s2[r] + t2[r], g.y2_act, pa2[r], pb2[r]);
                constexpr int kNone = -(1 << 30);
                int rows[3] = {trow, kNone, kNone};
                if (g.y2_reflect) {  // mirrored halo rows (no edge repeat): -j <- j, T-1+j <- T-1-j
                    if (trow >= 1 && trow <= g.y2_reflect) rows[1] = -trow;
                    if (trow <= g.Tout - 2 && trow >= g.Tout - 1 - g.y2_reflect) rows[2] = 2 * (g.Tout - 1) - trow;
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (q && rows[q] == kNone) continue;
                    float* dp = y2b + (size_t)(HALO + rows[q]) * g.y2_ld + c2;
                    if (vec2_ok && gn + 3 < g.y2_chi) *reinterpret_cast<f32x4*>(dp) = v2;
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (gn + r < g.y2_chi) dp[r] = v2[r];
                    }
                }
            }
        }
        if (g.stats) {
            if (quad) {  // Cg % 4 == 0: the lane's four columns sit in one group (every shipped width)
                float s1 = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]), q1 = (qsum[0] + qsum[1]) + (qsum[2] + qsum[3]);
                // the 16 lanes l & 15 of a quad share the channel quad: butterfly over the rows
#pragma unroll
                for (int o2 = 1; o2 < 16; o2 <<= 1) {
                    s1 += __shfl_xor(s1, o2, 64);
                    q1 += __shfl_xor(q1, o2, 64);
                }
                if (crow == 0 && gn < N) {
                    long long* bp = lbins + (gn / Cg - sg0) * kStatWords;
                    stat_bins_add(bp, s1);
                    stat_bins_add(bp + kStatBins, q1);
                }
            } else {  // narrow test configurations: per column
#pragma unroll
                for (int o2 = 1; o2 < 16; o2 <<= 1)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ssum[r] += __shfl_xor(ssum[r], o2, 64);
                        qsum[r] += __shfl_xor(qsum[r], o2, 64);
                    }
                if (crow == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (gn + r < N) {
                            long long* bp = lbins + ((gn + r) / Cg - sg0) * kStatWords;
                            stat_bins_add(bp, ssum[r]);
                            stat_bins_add(bp + kStatBins, qsum[r]);
                        }
                }
            }
        }
    }
    // zero halo of the second output: first / last row tile of phase 0, each for its own columns
    if (Y2 && y2b && g.y2_zero_halo && ph == 0 && (tm == 0 || tm == g.tiles_m_last)) {
        const int c_lo = max(n0, g.y2_clo), c_hi = min(min(n0 + BN, N), g.y2_chi);
        const int w4 = (c_hi - c_lo) > 0 ? (c_hi - c_lo + 3) / 4 : 0;
        for (int side = 0; side < 2; ++side) {
            if ((side == 0 && tm != 0) || (side == 1 && tm != g.tiles_m_last)) continue;
            const int row0 = side == 0 ? 0 : HALO + g.Tout;
            for (int idx = tid; idx < HALO * w4; idx += 128 * KS * RS) {
                const int rr = idx / w4, c = c_lo + 4 * (idx - rr * w4);
                float* dp = y2b + (size_t)(row0 + rr) * g.y2_ld + (c - g.y2_clo);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c + r < c_hi) dp[r] = 0.f;
            }
        }
    }
    if (g.stats) {
        __syncthreads();
        if (tid < sng * kStatWords) {
            const long long v = lbins[tid];
            if (v) {
                const int grp = sg0 + tid / kStatWords, k = tid % kStatWords;
                long long* sp = reinterpret_cast<long long*>(g.stats) + (size_t)(blockIdx.x % kStatSub) * g.sub_stride +
                                ((size_t)b * g.G + grp) * kStatWords + k;
                atomicAdd(reinterpret_cast<unsigned long long*>(sp), (unsigned long long)v);
            }
        }
    }
}
#undef AFTER_BAL_DMA

// w_out[ph][co][tap * Cp + ci] = packed[ph][co][tap][ci] (packed: conv.hip's weight-norm-folded
// [phase][Cout][taps][pad16(Cin)]), zero past Cin
__global__ void repack_tm_kernel(const float* __restrict__ w, float* __restrict__ out, int phases, int Cout,
                                 int taps, int Cin, int Cin_pad, int Cp) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)phases * Cout * taps * Cp;
    if (idx >= total) return;
    const int ci = idx % Cp;
    const int tap = (idx / Cp) % taps;
    const size_t pc = idx / ((size_t)Cp * taps);  // ph * Cout + co
    out[idx] = ci < Cin ? w[(pc * taps + tap) * Cin_pad + ci] : 0.f;
}

// stats[b][g] += (sum, sum of squares) of x[b, :, group g] (time-major, row pitch ld) for producers
// that are not convs (the PQMF analysis bank) and for concatenated tensors (UNET1D).  A thread keeps
// its channels (tid, tid + 256, ...) across the block's rows and adds its sums to LDS once.
__global__ __launch_bounds__(256) void stats_accum_tm_kernel(const float* __restrict__ x, int ld,
                                                             double* __restrict__ stats, int C, int T,
                                                             int G, int rows_per_block, int sub_stride) {
    // per-thread partial sums land in their own LDS slots and thread g < G adds group g's slots in a fixed order
    // (no LDS atomics: the block's contribution does not depend on thread scheduling)
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [2][max(256, C)]
    const int b = blockIdx.y;
    const int Cg = C / G;
    const int lo = blockIdx.x * rows_per_block, hi = min(lo + rows_per_block, T);
    const int W = C <= 128 ? 256 : C;
    if (C <= 128) {  // several rows per pass: slot = thread
        const int R = 256 / C;
        const int r = threadIdx.x / C, c = threadIdx.x - r * C;
        float s = 0.f, q = 0.f;
        if (r < R)
            for (int t = lo + r; t < hi; t += R) {
                const float v = x[((size_t)b * T + t) * ld + c];
                s += v;
                q += v * v;
            }
        sh[threadIdx.x] = s;
        sh[W + threadIdx.x] = q;
    } else {  // slot = channel
        for (int c = threadIdx.x; c < C; c += 256) {
            float s = 0.f, q = 0.f;
            for (int t = lo; t < hi; ++t) {
                const float v = x[((size_t)b * T + t) * ld + c];
                s += v;
                q += v * v;
            }
            sh[c] = s;
            sh[W + c] = q;
        }
    }
    __syncthreads();
    if (threadIdx.x < G) {
        float s = 0.f, q = 0.f;
        const int c0 = threadIdx.x * Cg;
        if (C <= 128) {
            const int R = 256 / C;
            for (int r = 0; r < R; ++r)
                for (int c = c0; c < c0 + Cg; ++c) {
                    s += sh[r * C + c];
                    q += sh[W + r * C + c];
                }
        } else {
            for (int c = c0; c < c0 + Cg; ++c) {
                s += sh[c];
                q += sh[W + c];
            }
        }
        // sub_stride > 0: the blocks spread over the sub-slots the consumer adds anyway (several hundred blocks'
        // atomics on the 16 words of one clip serialise otherwise: 43 us for the codec's 2-MB input)
        long long* sp = reinterpret_cast<long long*>(stats) + (size_t)(blockIdx.x % kStatSub) * sub_stride +
                        ((size_t)b * G + threadIdx.x) * kStatWords;
        stat_bins_add(sp, s);
        stat_bins_add(sp + kStatBins, q);
    }
}

// dst[b][t][coff + c] = src[b][t][c]  (row-pitched views; one thread per 4 channels when aligned)
__global__ __launch_bounds__(256) void copy_cols_tm_kernel(const float* __restrict__ src, int lds_,
                                                           float* __restrict__ dst, int ldd, int C, size_t rows,
                                                           int vec) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        const int q = C >> 2;
        if (idx >= rows * q) return;
        const size_t r = idx / q;
        const int c = 4 * (int)(idx - r * q);
        *reinterpret_cast<f32x4*>(dst + r * ldd + c) = *reinterpret_cast<const f32x4*>(src + r * lds_ + c);
    } else {
        if (idx >= rows * C) return;
        const size_t r = idx / C;
        const int c = (int)(idx - r * C);
        dst[r * ldd + c] = src[r * lds_ + c];
    }
}

// [B][C][T] -> [B][T][ld] (columns coff ..): 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void cm_to_tm_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                       int T, int ld) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        tile[i][tx] = (c < C && t < T) ? x[((size_t)b * C + c) * T + t] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        if (t < T && c < C) y[((size_t)b * T + t) * ld + c] = tile[tx][i];
    }
}

// y[b][c][t] = x[b][t][c] (diagnostic entry only: the bf16-pipe conv writes time-major)
__global__ __launch_bounds__(256) void tm_to_cm_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int T,
                                                       size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const size_t b = idx / ((size_t)C * T), rem = idx - b * (size_t)C * T;
    const int cc = (int)(rem / T), t = (int)(rem - (size_t)cc * T);
    y[idx] = x[(b * T + t) * C + cc];
}

// nn.Upsample(mode='nearest', scale_factor=r) on time-major rows: y[b][t][:] = x[b][t / r][:]
__global__ __launch_bounds__(256) void upsample_rows_tm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int C, int T, int r, size_t total4) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int q = C >> 2;
    const size_t row = idx / q;  // b * T * r + t_out
    const int c = 4 * (int)(idx - row * q);
    const size_t b = row / ((size_t)T * r);
    const size_t to = row - b * (size_t)T * r;
    *reinterpret_cast<f32x4*>(y + row * C + c) = *reinterpret_cast<const f32x4*>(x + (b * T + to / r) * C + c);
}

template <int MB, int NB, int KS, int NS, int RS>
int launch_tm_cfg(ConvTmArgs a, int B, bool dil, hipStream_t s) {
    constexpr int BM = 16 * MB, BN = 32 * NB;
    const int tiles_m = cdiv(a.Nn, BM), tiles_n = cdiv(a.Cout, BN);
    const size_t ring = size_t(NS) * KS * (BM + BN) * 32 * sizeof(float);
    const size_t red = KS > 1 ? size_t(2 * KS * RS) * (MB / RS) * NB * 256 * sizeof(float) : 0;
    // the statistics scratch ([2][waves][NB][4 or 16] floats, conv_tm_kernel) lives behind the reduction slabs
    // inside the ring when it fits: extra bytes would cost the 80-KiB split-K-4 ring its second workgroup per CU
    const size_t gsb = a.stats ? (size_t)16 * kStatWords * sizeof(long long) : 0;  // binned accumulators of <= 16 groups
    size_t lds = ring > red ? ring : red;
    if (KS > 1) {  // the split-K epilogue synchronises anyway; rings of 80 / 120 KiB must not grow
        a.gs_off = (int)(red / sizeof(float));
        a.gs_in_ring = 1;
        if (red + gsb > lds) lds = red + gsb;
    } else {
        a.gs_off = (int)(lds / sizeof(float));
        a.gs_in_ring = 0;
        lds += gsb;
    }
    static_assert(size_t(NS) * KS * (BM + BN) * 32 * sizeof(float) <= 160 * 1024, "ring exceeds the LDS");
    const int nwg = tiles_m * tiles_n, ny = B * a.phases;
    a.tiles_m_last = tiles_m - 1;
    // XCD grid pm x (8 / pm) over the tile matrix: an XCD fetches 1/pm of the input rows (each
    // Cp floats, shared by the taps) and 1/pn of the weights (K = taps * Cp floats per row)
    int pm = 0;
    if ((nwg & 7) == 0) {
        double best = 0;
        for (int c = 1; c <= 8; c *= 2) {
            if (tiles_m % c || tiles_n % (8 / c)) continue;
            const double cost = (double)a.Nn * a.Cp / c + (double)a.Cout * a.K / (8 / c);
            if (pm == 0 || cost < best) {
                pm = c;
                best = cost;
            }
        }
    }
    static LdsAttr attr[4];  // per (DIL, Y2) variant of this configuration
    auto go = [&](auto kern, int v) -> int {
        AFTER_TRY(ensure_lds_attr(attr[v], reinterpret_cast<const void*>(kern), lds));
        hipLaunchKernelGGL(kern, dim3(nwg * ny), dim3(128 * KS * RS), lds, s, a, tiles_m, tiles_n, pm, ny);
        AFTER_HIP_CHECK(hipGetLastError());
        return AFTER_OK;
    };
    if (a.y2) {
        if (dil) return go(conv_tm_kernel<MB, NB, KS, NS, RS, true, true>, 3);
        return go(conv_tm_kernel<MB, NB, KS, NS, RS, false, true>, 2);
    }
    if (dil) return go(conv_tm_kernel<MB, NB, KS, NS, RS, true, false>, 1);
    return go(conv_tm_kernel<MB, NB, KS, NS, RS, false, false>, 0);
}

int g_tm_force = -1;  // AFTER_CONV_TM_TILE / after_convtm_set_tile: tile configuration id (0 = heuristic)

// tile configurations by id (scripts/bench_conv.py sweeps them per layer shape)
int launch_tm_id(int id, const ConvTmArgs& a, int B, bool dil, hipStream_t s) {
    switch (id) {
        case 1: return launch_tm_cfg<4, 3, 1, 2, 2>(a, B, dil, s);   // 64 x 96, rows split over the waves
        case 2: return launch_tm_cfg<4, 2, 1, 2, 2>(a, B, dil, s);   // 64 x 64
        case 3: return launch_tm_cfg<4, 1, 1, 2, 2>(a, B, dil, s);   // 64 x 32
        case 4: return launch_tm_cfg<3, 1, 2, 2, 1>(a, B, dil, s);   // 48 x 32, 2 k-parts
        case 5: return launch_tm_cfg<3, 1, 4, 2, 1>(a, B, dil, s);   // 48 x 32, 4 k-parts (8 waves)
        case 6: return launch_tm_cfg<2, 3, 2, 2, 1>(a, B, dil, s);   // 32 x 96, 2 k-parts
        case 9: return launch_tm_cfg<2, 2, 2, 2, 1>(a, B, dil, s);   // 32 x 64, 2 k-parts
        // (measured and dropped, profiles/r2_bench_conv_*.jsonl: 48 x 96 and 64 x 96 split-K, 96 x 96,
        //  64 x 64 split-K, 32 x 32 split-K 4, and 3- / 4-deep rings of the split-K tiles -- a deeper
        //  ring does not help launches of one workgroup per CU)
        // 16-row tiles for the conditioning encoders (T = 128 / 256 positions): the only way to put
        // a 256 x 512 output on all 256 CUs
        case 15: return launch_tm_cfg<1, 1, 4, 2, 1>(a, B, dil, s);  // 16 x 32, 4 k-parts (8 waves)
        case 16: return launch_tm_cfg<1, 1, 2, 2, 1>(a, B, dil, s);  // 16 x 32, 2 k-parts
        default: break;
    }
    set_error("conv_tm: no tile configuration %d", id);
    return AFTER_E_INVALID;
}


// =====================================================================================================
// ConvBlock1d with a k = 1 conv in ONE launch (the second conv of every ResnetBlock1d: GroupNorm -> SnakeBeta -> Conv1d(k = 1)
// + residual, SimpleNetsStream.py:150-194, :196-254): the activated tensor never exists in memory.  A k = 1 conv needs no
// halo and no neighbour rows, so a workgroup that owns BM rows of ALL output channels activates its input rows exactly once:
// raw rows -> registers (requested first, beside the first weight fragments), the producer's GroupNorm statistics -> per-channel
// scale / shift (gn_mean_rstd), act(x * sc + sh) -> an fp32 tile in LDS, then Y^T = W A^T as v_mfma_f32_16x16x4_f32 (the exact
// fp32 fma chain of conv_tm's own MFMAs) with the weight fragments streamed from the L2 through a four-deep register ring -- wave w
// owns the column blocks [w NBW, (w + 1) NBW) for all BM rows -- and conv_x6's epilogue: bias, residual, fp32 rows, the NEXT
// GroupNorm's statistics through binned integer accumulators.  Against act_pad + conv it saves a launch, the activated tensor's
// write (fp32, or 1.5 x that as bf16 planes) and its read.
struct Conv1ActArgs {
    const float* x;        // [B * T][ldx] time-major raw input
    const double* stats_in;  // accumulators of x (GroupNorm) or nullptr
    const float *gamma, *beta, *act_a, *act_b;
    const float* w;        // [C][ldw]: conv_tm's GEMM operand of a k = 1 conv
    const float* bias;
    const float* res;      // [B * T][res_ld] or nullptr (may alias y)
    float* y;              // [B * T][y_ld]
    double* stats_out;     // accumulators of y or nullptr
    int ldx, ldw, res_ld, y_ld, T, C, G, sub_stride, act;
    float eps;
};

template <int NW, int NBW, int MB>  // waves, 16-column blocks per wave (C = 16 NBW NW), 16-row blocks per workgroup
__global__ __launch_bounds__(64 * NW) void conv1_act_kernel(Conv1ActArgs a) {
    constexpr int NT = 64 * NW, BM = 16 * MB, C = 16 * NBW * NW, LDA = C + 4, Q = C / 4, PD = 4;
    constexpr int NLD = BM * Q / NT;  // 16-byte pieces of the raw tile per thread
    static_assert((BM * Q) % NT == 0, "raw tile pieces per thread");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const At = smem;                 // [BM][LDA]
    float* const prm = smem + BM * LDA;     // sc | sh | pa | pb, [4][C]
    __shared__ float gmean[16], grstd[16];
    __shared__ __attribute__((aligned(16))) long long swl[16 * kStatSub * kStatWords];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM, b = m0 / a.T;
    const int n0 = w * NBW * 16, fr = lane & 15, fq = lane >> 4;
    // ---- requests first: the raw rows and the first weight fragments (one exposed memory latency for the whole prologue)
    f32x4 xv[NLD];
#pragma clang loop unroll(full)
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + NT * i, r = idx / Q, q = idx - r * Q;
        xv[i] = *reinterpret_cast<const f32x4*>(a.x + (size_t)(m0 + r) * a.ldx + 4 * q);
    }
    f32x4 wf[PD][NBW];
    const float* wl = a.w + (size_t)(n0 + fr) * a.ldw + 4 * fq;  // lane (fr, fq): row n0 + 16 j + fr, k = 16 kb + 4 fq .. + 3
    constexpr int NKB = C / 16;
#pragma unroll
    for (int u = 0; u < PD - 1; ++u)
#pragma unroll
        for (int j = 0; j < NBW; ++j)
            wf[u][j] = u < NKB ? *reinterpret_cast<const f32x4*>(wl + (size_t)16 * j * a.ldw + 16 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- per-channel scale / shift of the GroupNorm (or none), Snake parameters -> LDS.  Every parameter of this thread's
    //      channels is requested before the statistics are waited for: one memory latency, not one per parameter
    constexpr int PK = (C + NT - 1) / NT;
    float pg[PK], pbe[PK], ppa[PK], ppb[PK];
#pragma unroll
    for (int k = 0; k < PK; ++k) {
        const int c = min(tid + NT * k, C - 1);
        pg[k] = a.gamma ? a.gamma[c] : 1.f;
        pbe[k] = a.gamma ? a.beta[c] : 0.f;
        ppa[k] = a.act_a ? a.act_a[c] : 0.f;
        ppb[k] = a.act_b ? a.act_b[c] : 0.f;
    }
    if (a.stats_in) gn_mean_rstd(a.stats_in, b, a.G, a.sub_stride, C, a.T, a.eps, swl, gmean, grstd, NT);
#pragma unroll
    for (int k = 0; k < PK; ++k) {
        const int c = tid + NT * k;
        if (c < C) {
            float sc = pg[k], sh = pbe[k];
            if (a.stats_in) {
                const int g = c / (C / a.G);
                sc = grstd[g] * sc;
                sh = sh - gmean[g] * sc;
            }
            prm[c] = sc;
            prm[C + c] = sh;
            prm[2 * C + c] = ppa[k];
            prm[3 * C + c] = ppb[k];
        }
    }
    __syncthreads();
    // ---- activate the tile into LDS
#pragma clang loop unroll(full)
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + NT * i, r = idx / Q, q = idx - r * Q;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(prm + 4 * q), sh = *reinterpret_cast<const f32x4*>(prm + C + 4 * q);
        const f32x4 pa = *reinterpret_cast<const f32x4*>(prm + 2 * C + 4 * q), pb = *reinterpret_cast<const f32x4*>(prm + 3 * C + 4 * q);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = act_apply(xv[i][k] * sc[k] + sh[k], a.act, pa[k], pb[k]);
        *reinterpret_cast<f32x4*>(At + r * LDA + 4 * q) = o;
    }
    __syncthreads();
    // ---- Y^T = W A^T: the W fragment as srcA (the accumulator holds four consecutive columns of one row per lane)
    f32x4 acc[MB][NBW];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (the epilogue's operands -- bias, the residual tile -- are requested in front of the K loop and land under its MFMAs)
    f32x4 bvv[NBW], rvv[MB][NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int gn = n0 + 16 * j + 4 * fq;
        bvv[j] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MB; ++i)
            rvv[i][j] = a.res ? *reinterpret_cast<const f32x4*>(a.res + (size_t)(m0 + 16 * i + fr) * a.res_ld + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float* al = At + fr * LDA + 4 * fq;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {  // (fully unrolled: the ring slot of a k-block is a compile-time register set)
        if (kb + PD - 1 < NKB) {
#pragma unroll
            for (int j = 0; j < NBW; ++j)
                wf[(kb + PD - 1) % PD][j] = *reinterpret_cast<const f32x4*>(wl + (size_t)16 * j * a.ldw + 16 * (kb + PD - 1));
        }
        f32x4 af[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) af[i] = *reinterpret_cast<const f32x4*>(al + 16 * i * LDA + 16 * kb);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NBW; ++j)
#pragma unroll
                for (int i = 0; i < MB; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kb % PD][j][r], af[i][r], acc[i][j], 0, 0, 0);
    }
    // ---- epilogue: bias, residual, rows; statistics of the next GroupNorm (conv_x6's: butterfly over the 16 rows of a lane
    //      quad, binned integer accumulators in LDS, the non-zero words on to the global ones)
    long long* const lbins = swl;  // [G][kStatWords] (the input statistics' staging area is done with)
    if (a.stats_out) {
        __syncthreads();
        for (int i = tid; i < a.G * kStatWords; i += NT) lbins[i] = 0;
    }
    const int Cg = C / (a.G > 0 ? a.G : 1);
    float ssum[NBW], qsum[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int gn = n0 + 16 * j + 4 * fq;
        ssum[j] = qsum[j] = 0.f;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const size_t row = (size_t)(m0 + 16 * i + fr);
            const f32x4 o = (acc[i][j] + bvv[j]) + rvv[i][j];
            *reinterpret_cast<f32x4*>(a.y + row * a.y_ld + gn) = o;
            ssum[j] += (o[0] + o[1]) + (o[2] + o[3]);
            qsum[j] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
        }
    }
    if (a.stats_out) {
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
            for (int o2 = 1; o2 < 16; o2 <<= 1) {
                ssum[j] += __shfl_xor(ssum[j], o2, 64);
                qsum[j] += __shfl_xor(qsum[j], o2, 64);
            }
        __syncthreads();  // the bins are zero
        if (fr == 0) {
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                long long* bp = lbins + ((n0 + 16 * j + 4 * fq) / Cg) * kStatWords;
                stat_bins_add(bp, ssum[j]);
                stat_bins_add(bp + kStatBins, qsum[j]);
            }
        }
        __syncthreads();
        if (tid < a.G * kStatWords) {
            const long long v = lbins[tid];
            if (v) {
                long long* sp = reinterpret_cast<long long*>(a.stats_out) + (size_t)(blockIdx.x % kStatSub) * a.sub_stride +
                                (size_t)b * a.G * kStatWords + tid;
                atomicAdd(reinterpret_cast<unsigned long long*>(sp), (unsigned long long)v);
            }
        }
    }
}

template <int NW, int NBW, int MB>
int launch_conv1_act_cfg(const Conv1ActArgs& a, long long rows, hipStream_t s) {
    constexpr int BM = 16 * MB, C = 16 * NBW * NW;
    const size_t lds = ((size_t)BM * (C + 4) + 4 * C) * sizeof(float);
    static LdsAttr attr;
    AFTER_TRY(ensure_lds_attr(attr, reinterpret_cast<const void*>(conv1_act_kernel<NW, NBW, MB>), lds));
    hipLaunchKernelGGL((conv1_act_kernel<NW, NBW, MB>), dim3((unsigned)(rows / BM)), dim3(64 * NW), lds, s, a);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

long long g_conv1_act_launches = 0;

}  // namespace

int conv_tm_halo() { return HALO; }
int conv_tm_stat_sub() { return kStatSub; }
int conv_tm_cp(int C) { return (C + 31) & ~31; }
int conv_tm_rows(int T) { return T + 2 * HALO; }

int snake_variant(int act) {
    static int libm = -1;
    if (libm < 0) {
        const char* e = getenv("AFTER_SNAKE_LIBM");
        libm = e ? atoi(e) : 0;
    }
    return act == ACT_SNAKE && libm ? ACT_SNAKE_LIBM : act;
}

// Is GroupNorm -> Snake -> Conv1d(k = 1) (+ residual) of this size one launch of conv1_act_kernel?  The narrow, long stages
// (64 / 96 / 128 channels at T >= 12288: the decoder's last stage, the encoder's first two), where act_pad + conv are two
// bandwidth- and latency-bound launches: 14.7 us against 11 + 16 at one clip, 52.7 against 31 + 33 at eight (64 channels at
// T = 32768).  Measured and left on the two launches: 192 channels (37.6 us against 11 + 16 / 190 against 33 + 77) and 384
// (41.9 against 19 + 22 / 185 against 42 + 123) -- there the conv is MFMA-bound, and the fp32 MFMAs of this kernel lose to the
// bf16-pipe conv by more than the activated tensor's round trip costs.
bool conv1_act_eligible(int B, int T, int C, int G, bool stats) {
    static int on = -1;  // AFTER_AE_FUSE_K1=0: A/B switch (act_pad + conv launches)
    if (on < 0) {
        const char* e = getenv("AFTER_AE_FUSE_K1");
        on = e ? atoi(e) : 1;
    }
    if (!on) return false;
    if (C != 64 && C != 96 && C != 128) return false;
    const int BM = 64;
    if (T % BM) return false;
    if (stats && (G < 1 || G > 16 || C % G || ((C / G) & 3))) return false;
    return (long long)B * T / BM >= 192;
}

long long conv1_act_launches() { return g_conv1_act_launches; }

int launch_conv1_act(const Conv1ActRun& r, hipStream_t s) {
    AFTER_REQUIRE(conv1_act_eligible(r.B, r.T, r.C, r.G, r.stats_in || r.stats_out), AFTER_E_INVALID, "conv1_act: not a fused launch");
    Conv1ActArgs a;
    memset(&a, 0, sizeof(a));
    a.x = r.x, a.stats_in = r.stats_in, a.gamma = r.gamma, a.beta = r.beta, a.act_a = r.act_a, a.act_b = r.act_b;
    a.w = r.w, a.bias = r.bias, a.res = r.res, a.y = r.y, a.stats_out = r.stats_out;
    a.ldx = r.ldx > 0 ? r.ldx : r.C, a.ldw = r.ldw, a.res_ld = r.res_ld > 0 ? r.res_ld : r.C, a.y_ld = r.y_ld > 0 ? r.y_ld : r.C;
    a.T = r.T, a.C = r.C, a.G = r.G, a.sub_stride = r.sub_stride, a.act = snake_variant(r.act), a.eps = 1e-5f;
    AFTER_REQUIRE((a.ldx & 3) == 0 && (a.ldw & 3) == 0 && (a.res_ld & 3) == 0 && (a.y_ld & 3) == 0 && ((uintptr_t)r.x & 15) == 0 &&
                      ((uintptr_t)r.w & 15) == 0 && ((uintptr_t)r.y & 15) == 0 && (!r.res || ((uintptr_t)r.res & 15) == 0) &&
                      (!r.bias || ((uintptr_t)r.bias & 15) == 0),
                  AFTER_E_INVALID, "conv1_act: operand alignment");
    const long long rows = (long long)r.B * r.T;
    ++g_conv1_act_launches;
    switch (r.C) {
        case 64: return launch_conv1_act_cfg<4, 1, 4>(a, rows, s);
        case 96: return launch_conv1_act_cfg<2, 3, 4>(a, rows, s);
        case 128: return launch_conv1_act_cfg<4, 2, 4>(a, rows, s);
        default: break;
    }
    return AFTER_E_INVALID;
}

int launch_act_pad_tm(const ActPadTm& p, hipStream_t s) {
    ActTmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = p.x;
    a.y = p.y;
    a.y3 = p.y3;
    a.hscale = p.y3 ? p.hscale : 0.f;
    a.rows16 = conv_x6_rows(p.T);
    AFTER_REQUIRE(!p.y3 || !p.state, AFTER_E_INVALID, "act_pad_tm: plane output is for whole-clip passes");
    a.stats = p.stats;
    a.gamma = p.gamma;
    a.beta = p.beta;
    a.act_a = p.act_a;
    a.act_b = p.act_b;
    a.state = p.state;
    a.state_out = p.state ? p.state_out : nullptr;
    a.scale_b = p.scale_b;
    a.shift_b = p.shift_b;
    a.act = snake_variant(p.act);
    a.C = p.C;
    a.Cp = conv_tm_cp(p.C);
    a.T = p.T;
    a.Tp = conv_tm_rows(p.T);
    a.G = p.G;
    a.x_cm = p.x_cm;
    a.ldx = p.ldx > 0 ? p.ldx : p.C;
    a.pad_reflect = p.pad_reflect;
    a.sub_stride = p.sub_stride;
    a.x2 = p.x2;
    a.ldx2 = p.ldx2;
    a.stat_T = p.stat_T > 0 ? p.stat_T : p.T;
    AFTER_REQUIRE(!p.x2 || (!p.x_cm && (p.ldx2 & 3) == 0), AFTER_E_INVALID, "act_pad_tm: x2 needs time-major inputs");
    a.eps = 1e-5f;
    AFTER_REQUIRE(a.Cp <= 1024, AFTER_E_INVALID, "act_pad_tm: at most 1024 channels (got %d)", p.C);
    AFTER_REQUIRE(!p.x_cm ? ((a.ldx & 3) == 0 || p.C < 4) : true, AFTER_E_INVALID, "act_pad_tm: ldx %% 4 != 0");
    AFTER_REQUIRE(!p.stats || ((p.C % p.G) == 0 && p.G <= 16), AFTER_E_INVALID, "act_pad_tm: G | C, G <= 16");
    const int Q = a.Cp / 4, R = 256 / Q;
    // >= ~3 blocks per CU where the tensor allows, >= 4 passes per block
    int rpb = R * 4;
    while ((long long)cdiv(a.Tp, rpb) * p.B > 2048) rpb *= 2;
    a.rows_per_block = rpb;
    AFTER_REQUIRE(R >= 1, AFTER_E_INVALID, "act_pad_tm: too many channels");
    static int xcd_rows = -1;  // AFTER_ACT_XCD=0: A/B switch
    if (xcd_rows < 0) {
        const char* e = getenv("AFTER_ACT_XCD");
        xcd_rows = e ? atoi(e) : 1;
    }
    if (p.y3) {
        AFTER_REQUIRE(!p.x_cm && !p.x2 && !p.scale_b && !p.pad_reflect && (a.ldx & 3) == 0 && ((uintptr_t)p.x & 15) == 0,
                      AFTER_E_INVALID, "act_pad_tm: plane output takes plain time-major inputs");
        const int nkb = a.Cp / 32, nrb = a.rows16 / 16;
        static int loop_rb = -1;  // AFTER_ACT_LOOP=0: A/B switch; n: row blocks a wave of the looping kernel walks
        if (loop_rb < 0) {
            const char* e = getenv("AFTER_ACT_LOOP");
            loop_rb = e ? atoi(e) : 8;
        }
        if (loop_rb >= 4 && (long long)cdiv(nrb, loop_rb) * nkb * p.B >= 1024) {  // large tensors: waves that loop (>= one wave per SIMD)
            a.xcd_rows = 0;
            hipLaunchKernelGGL(act_pad_x6_loop_kernel, dim3(cdiv(cdiv(nrb, loop_rb) * nkb, 4), p.B), dim3(256), 0, s, a, nkb, loop_rb);
            AFTER_HIP_CHECK(hipGetLastError());
            return AFTER_OK;
        }
        int RB = 4;
        while (RB > 1 && (long long)cdiv(nrb, RB) * nkb * p.B < 4 * 768) RB >>= 1;  // >= 3 blocks per CU where the tensor allows
        int nwg = cdiv(cdiv(nrb, RB) * nkb, 4);
        a.xcd_rows = xcd_rows && nwg >= 64 && p.B == 1;
        if (a.xcd_rows) nwg = (nwg + 7) & ~7;
        hipLaunchKernelGGL(act_pad_x6_kernel, dim3(nwg, p.B), dim3(256), 0, s, a, nkb, RB);
        AFTER_HIP_CHECK(hipGetLastError());
        return AFTER_OK;
    }
    int nb = cdiv(a.Tp, rpb);
    a.xcd_rows = xcd_rows && nb >= 64 && p.B == 1;  // measured: +1.5 % decode at one clip, -0.6 % at eight
    if (a.xcd_rows) nb = (nb + 7) & ~7;
    hipLaunchKernelGGL(act_pad_tm_kernel, dim3(nb, p.B), dim3(256), 0, s, a);
    AFTER_HIP_CHECK(hipGetLastError());
    if (p.state && !p.state_out) {
        const int total4 = p.B * HALO * a.Cp / 4;
        hipLaunchKernelGGL(state_update_tm_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, s, p.y, p.state, a.Cp,
                           p.T, a.Tp, total4);
        AFTER_HIP_CHECK(hipGetLastError());
    }
    return AFTER_OK;
}

int launch_stats_accum_tm(const float* x, double* stats, int B, int C, int T, int G, hipStream_t s, int ld, int sub_stride) {
    AFTER_REQUIRE(G >= 1 && G <= 16 && C % G == 0, AFTER_E_INVALID, "stats_accum_tm: G <= 16, G | C");
    int rpb = C <= 128 ? 64 : 16;
    while ((long long)cdiv(T, rpb) * B > 1024) rpb *= 2;
    hipLaunchKernelGGL(stats_accum_tm_kernel, dim3(cdiv(T, rpb), B), dim3(256), 2 * (size_t)(C <= 128 ? 256 : C) * sizeof(float), s, x, ld > 0 ? ld : C, stats, C,
                       T, G, rpb, sub_stride);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

int launch_copy_cols_tm(const float* src, int ld_src, float* dst, int ld_dst, int C, size_t rows, hipStream_t s) {
    const int vec = ((C | ld_src | ld_dst) & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    const size_t n = vec ? rows * (C >> 2) : rows * C;
    hipLaunchKernelGGL(copy_cols_tm_kernel, dim3((unsigned)cdivll((long long)n, 256)), dim3(256), 0, s, src, ld_src,
                       dst, ld_dst, C, rows, vec);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

int launch_cm_to_tm(const float* x, float* y, int B, int C, int T, int ld, hipStream_t s) {
    hipLaunchKernelGGL(cm_to_tm_kernel, dim3(cdiv(T, 32), cdiv(C, 32), B), dim3(256), 0, s, x, y, C, T, ld);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

int launch_upsample_rows_tm(const float* x, float* y, int B, int C, int T, int r, hipStream_t s) {
    AFTER_REQUIRE((C & 3) == 0, AFTER_E_INVALID, "upsample_rows_tm: C %% 4 != 0");
    const size_t total4 = (size_t)B * T * r * (C >> 2);
    hipLaunchKernelGGL(upsample_rows_tm_kernel, dim3((unsigned)cdivll((long long)total4, 256)), dim3(256), 0, s, x, y,
                       C, T, r, total4);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

void conv_tm_plan(const ConvDmaPlanIn& in, ConvTmPlan* p) {
    p->Cp = conv_tm_cp(in.Cin);
    p->K = in.taps * p->Cp;
    p->dil = in.taps > 1 ? in.toff[0][1] - in.toff[0][0] : 1;
    p->ok = true;
    for (int ph = 0; ph < in.phases; ++ph) {
        for (int t = 1; t < in.taps; ++t)
            if (in.toff[ph][t] - in.toff[ph][t - 1] != p->dil) p->ok = false;  // uniform tap spacing only
        if (HALO + in.toff[ph][0] < 0 || in.toff[ph][in.taps - 1] > HALO) p->ok = false;
    }
    if (p->dil < 1) p->ok = false;
    p->w_floats = (size_t)in.phases * in.Cout * p->K;
}

int conv_tm_repack(const float* packed, float* out, const ConvDmaPlanIn& in, const ConvTmPlan& p,
                   hipStream_t s) {
    AFTER_REQUIRE(p.ok, AFTER_E_INVALID, "conv_tm: unsupported tap pattern");
    hipLaunchKernelGGL(repack_tm_kernel, dim3((unsigned)cdivll(p.w_floats, 256)), dim3(256), 0, s, packed, out,
                       in.phases, in.Cout, in.taps, in.Cin, pad16(in.Cin), p.Cp);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

int launch_conv_tm(const ConvTmRun& r, const ConvDmaPlanIn& in, const ConvTmPlan& p, hipStream_t s) {
    AFTER_REQUIRE(p.ok, AFTER_E_INVALID, "conv_tm: unsupported tap pattern");
    ConvTmArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = r.xp;
    a.w = r.w;
    a.bias = r.bias;
    a.res = r.res;
    a.y = r.y;
    a.stats = r.stats;
    a.Cp = p.Cp;
    a.Cout = in.Cout;
    a.Tp = r.Tp;
    a.Tout = r.Tout;
    a.K = p.K;
    a.phases = in.phases;
    // row pitch / batch stride of the input (defaults: the dense haloed buffer of act_pad_tm)
    a.x_ld = r.x_ld > 0 ? r.x_ld : p.Cp;
    a.x_bs = r.x_bs > 0 ? r.x_bs : (long long)r.Tp * a.x_ld;
    a.lda = in.istride * a.x_ld;
    a.ostride = in.ostride;
    a.y_ld = r.y_ld > 0 ? r.y_ld : in.Cout;
    a.y_coff = r.y_coff;
    a.y_bs = r.y_bs > 0 ? r.y_bs : (long long)r.Tout * in.Cout;
    a.res_ld = r.res_ld > 0 ? r.res_ld : in.Cout;
    a.res_coff = r.res_coff;
    a.res_cm = r.res_cm;
    a.res_bs = r.res_bs > 0 ? r.res_bs : (long long)r.Tout * in.Cout;
    a.bias_bstride = r.bias_bstride;
    a.y2 = r.y2;
    if (r.y2) {
        a.y2_scale = r.y2_scale;
        a.y2_shift = r.y2_shift;
        a.y2_pa = r.y2_pa;
        a.y2_pb = r.y2_pb;
        a.y2_act = snake_variant(r.y2_act);
        a.y2_clo = r.y2_clo;
        a.y2_chi = r.y2_chi > 0 ? r.y2_chi : in.Cout;
        a.y2_ld = r.y2_ld > 0 ? r.y2_ld : conv_tm_cp(a.y2_chi - a.y2_clo);
        a.y2_coff = r.y2_coff;
        a.y2_bs = r.y2_bs > 0 ? r.y2_bs : (long long)conv_tm_rows(r.Tout) * a.y2_ld;
        a.y2_add = r.y2_add;
        a.y2_add_ld = r.y2_add_ld;
        a.y2_add_coff = r.y2_add_coff;
        a.y2_add_bs = r.y2_add_bs > 0 ? r.y2_add_bs : (long long)r.Tout * r.y2_add_ld;
        a.y2_reflect = r.y2_reflect;
        a.y2_zero_halo = r.y2_reflect ? 0 : 1;
        AFTER_REQUIRE(r.y2_reflect <= HALO && r.y2_reflect < r.Tout, AFTER_E_INVALID, "conv_tm: reflect pad too wide");
        AFTER_REQUIRE(in.ostride == 1 || !r.y2_reflect, AFTER_E_INVALID, "conv_tm: reflect halo with phases");
    }
    a.Nn = r.Nn;
    a.G = r.G;
    a.sub_stride = r.sub_stride;
    a.y_cm = r.y_cm;
    a.post_scale = r.post_scale;
    a.post_shift = r.post_shift;
    a.post_bstride = r.post_bstride;
    a.out_act = r.out_act;
    const int npt = p.Cp / 32;
    a.magic = (65536 + npt - 1) / npt;  // exact for slab < 65536 / npt ... (taps * npt <= 8 * 32)
    a.extra = p.dil * a.x_ld - p.Cp;  // slab S of tap t starts at S * 32 + t * (dil * x_ld - Cp)
    for (int ph = 0; ph < in.phases; ++ph) {
        a.abase[ph] = (HALO + in.toff[ph][0]) * a.x_ld;
        a.ooff[ph] = in.ooff[ph];
    }
    AFTER_REQUIRE(!r.stats || (in.Cout % r.G == 0 && r.G <= 16), AFTER_E_INVALID,
                  "conv_tm: fused statistics need G | Cout, G <= 16 (Cout=%d G=%d)", in.Cout, r.G);
    AFTER_REQUIRE((size_t)r.Tp * a.x_ld < (1u << 28) && (size_t)in.Cout * p.K < (1u << 28), AFTER_E_INVALID,
                  "conv_tm: operand too large for 32-bit DMA offsets");
    AFTER_REQUIRE((a.x_ld & 3) == 0 && ((uintptr_t)r.xp & 15) == 0, AFTER_E_INVALID, "conv_tm: input row pitch / base alignment");
    const bool dil = in.taps > 1 && a.extra != 0;
    if (g_tm_force < 0) {
        const char* e = getenv("AFTER_CONV_TM_TILE");
        g_tm_force = e ? atoi(e) : 0;
    }
    const int N = in.Cout, K = p.K;
    const long long ny = (long long)r.B * in.phases;
    if (g_tm_force > 0) {
        static const int ksof[] = {0, 1, 1, 1, 2, 4, 2, 0, 0, 2, 0, 0, 0, 0, 0, 4, 2};
        AFTER_REQUIRE(g_tm_force <= 16 && ksof[g_tm_force] > 0 && K % (32 * ksof[g_tm_force]) == 0, AFTER_E_INVALID,
                      "conv_tm: tile %d needs K %% %d == 0 (K=%d)", g_tm_force, 32 * ksof[g_tm_force > 16 ? 1 : g_tm_force], K);
        return launch_tm_id(g_tm_force, a, r.B, dil, s);
    }
    // Tile choice by a small cost model fitted to the per-layer sweeps (scripts/bench_conv.py,
    // profiles/r2_bench_conv_*.jsonl): a launch costs the longest per-CU chain of MFMAs -- workgroups
    // per CU (ceil: 528 workgroups on 256 CUs run as 3 rounds, which is why 48 x 32 tiles lost 35 % to
    // 32 x 96 ones on the 768-channel stage) x waves sharing a SIMD x MFMAs per wave -- plus a per-round
    // fixed cost, plus the L2 -> LDS operand traffic of the tile shape (bytes per flop).
    struct Cand {
        int id, mb, nb, ks, rs;
    };
    static const Cand cands[] = {{1, 4, 3, 1, 2}, {2, 4, 2, 1, 2}, {3, 4, 1, 1, 2}, {6, 2, 3, 2, 1}, {9, 2, 2, 2, 1},
                                 {4, 3, 1, 2, 1}, {5, 3, 1, 4, 1}, {16, 1, 1, 2, 1}, {15, 1, 1, 4, 1}};
    int best = 0;
    double best_cost = 0;
    for (const Cand& c : cands) {
        if (K % (32 * c.ks)) continue;
        const int bm = 16 * c.mb, bn = 32 * c.nb;
        if (bn > 32 && N <= bn - 32) continue;  // a tile wider than the layer wastes whole column blocks
        const double wgs = (double)cdiv(r.Nn, bm) * cdiv(N, bn) * (double)ny;
        const double rounds = wgs <= 256 ? 1.0 : (wgs < 2048 ? (double)cdivll((long long)wgs, 256) : wgs / 256.0);
        const int waves = 2 * c.ks * c.rs;
        const double per_wave = (double)(c.mb / c.rs) * c.nb * (K / c.ks) / 4.0;  // MFMAs
        // waves resident per SIMD: with 3 or more a wave's DMA / barrier / fragment-read stalls hide
        // behind its neighbours' MFMAs (64 x 32 tiles beat 64 x 96 ones by 5-8 % at one clip)
        double wps = wgs * waves / 1024.0;
        wps = wps < 1 ? 1 : (wps > 4 ? 4 : wps);
        // one accumulator per wave: the 40-cycle dependent latency of the MFMA, not its 32-cycle issue
        const double cyc = (c.mb / c.rs) * c.nb == 1 ? 40.0 : 32.0;
        const double mfma = rounds * (waves / 4.0) * per_wave * cyc * (1.0 + 0.15 / wps);  // cycles
        const double fixed = rounds * 2500.0;
        const double bytes_per_flop = (bm + bn) * 4.0 / (2.0 * bm * bn);
        const double traffic = 2.0 * r.Nn * (double)N * K * ny * bytes_per_flop / (256.0 * 40.0);  // ~40 B/clk/CU
        double cost = (mfma > traffic ? mfma : traffic) + fixed + 0.25 * (mfma < traffic ? mfma : traffic);
        if (c.id == 1) cost *= 1.08;  // measured: 64 x 96 trails 64 x 64 by 1-5 % wherever both balance
        // k = 1 convs with a residual / statistics / second-output epilogue: the K loop is 6-24 slabs, the
        // epilogue moves as many bytes as the main loop, and the workgroups of a CU run in phase -- the
        // narrow tiles (7 waves per SIMD instead of 4-5) overlap them best.  Measured inside the decoder
        // (per-launch traces with forced tiles, B = 1 and 8): 64 x 32 beats 64 x 96 by 10-30 %, 64 x 64
        // by 0-15 %.
        if (in.taps == 1 && in.phases == 1 && (r.res || r.stats || r.y2)) cost *= c.nb == 3 ? 1.6 : (c.nb == 2 ? 1.3 : 1.0);
        if (!best || cost < best_cost) {
            best = c.id;
            best_cost = cost;
        }
    }
    // (same measurement: once 64 x 32 tiles fill the chip twice over they win every such launch)
    if (in.taps == 1 && in.phases == 1 && (r.res || r.stats || r.y2) &&
        (double)cdiv(r.Nn, 64) * cdiv(N, 32) * (double)ny >= 512)
        best = 3;
    AFTER_REQUIRE(best, AFTER_E_INVALID, "conv_tm: no tile configuration for K=%d", K);
    return launch_tm_id(best, a, r.B, dil, s);
}

}  // namespace after

// ---------------------------------------------------------------------------------------------
// Diagnostic / unit-test entry points (not on the reference's surface): one Conv1d layer on the
// time-major path -- act(x) into the haloed buffer, then the conv GEMM -- callable on its own for
// parity tests against a plain fp32 conv and for the per-layer tile sweeps of scripts/bench_conv.py.
struct after_convtm {
    after::ConvDmaPlanIn in;
    after::ConvTmPlan plan;
    after::Arena ar;
    float *w = nullptr, *bias = nullptr, *xp = nullptr, *xtm = nullptr, *ytm = nullptr, *res = nullptr;
    unsigned short *xp3 = nullptr, *w3 = nullptr;  // bf16-plane operands of the same layer (conv_x6.hip), where eligible
    double* stats = nullptr;
    int B, Cin, Cout, T, Tout, act;
};

extern "C" void after_convtm_destroy(after_convtm* h) {
    if (!h) return;
    h->ar.release();
    delete h;
}

extern "C" void after_convtm_set_tile(int id) { after::g_tm_force = id; }

// w: [Cout, Cin, k] (torch Conv1d layout), bias [Cout] or null.  Output length
// Tout = (T + pad_l + pad_r - (k-1) dil - 1) / stride + 1 with pad_r implied by Tout_hint.
extern "C" int after_convtm_create(const float* w, const float* bias, int B, int Cin, int Cout, int T, int Tout,
                                   int k, int dil, int stride, int left_pad, int act, after_convtm** out) {
    using namespace after;
    AFTER_REQUIRE(w && out && B > 0 && k >= 1 && k <= kMaxTaps, AFTER_E_INVALID, "convtm: bad argument");
    *out = nullptr;
    after_convtm* h = new (std::nothrow) after_convtm();
    AFTER_REQUIRE(h, AFTER_E_NOMEM, "out of host memory");
    memset(&h->in, 0, sizeof(h->in));
    h->in.Cin = Cin;
    h->in.Cout = Cout;
    h->in.taps = k;
    h->in.phases = 1;
    h->in.istride = stride;
    h->in.ostride = 1;
    for (int t = 0; t < k; ++t) h->in.toff[0][t] = t * dil - left_pad;
    conv_tm_plan(h->in, &h->plan);
    h->B = B;
    h->Cin = Cin;
    h->Cout = Cout;
    h->T = T;
    h->Tout = Tout;
    h->act = act;
    const size_t xpn = (size_t)B * conv_tm_rows(T) * h->plan.Cp, yn = (size_t)B * Tout * Cout;
    const size_t packed = (size_t)Cout * k * pad16(Cin);
    const bool x6 = conv_x6_eligible(h->in, h->plan);
    const size_t xp3n = x6 ? conv_x6_plane_elems(B, T, Cin) : 0, w3n = x6 ? conv_x6_weight_elems(h->in, h->plan) : 0;
    int rc = h->ar.init((h->plan.w_floats + packed + Cout + xpn + (size_t)B * T * Cin + 2 * yn) * sizeof(float) +
                        (xp3n + w3n) * sizeof(unsigned short) +
                        (size_t)conv_tm_stat_sub() * B * 8 * kStatWords * sizeof(double) + (1 << 16));
    auto fail = [&](int code) {
        after_convtm_destroy(h);
        return code;
    };
    if (rc != AFTER_OK) return fail(rc);
    float* pk = h->ar.take<float>(packed);
    h->w = h->ar.take<float>(h->plan.w_floats);
    h->bias = h->ar.take<float>(Cout);
    h->xp = h->ar.take<float>(xpn);
    h->xtm = h->ar.take<float>((size_t)B * T * Cin);
    h->ytm = h->ar.take<float>(yn);
    h->res = h->ar.take<float>(yn);
    h->stats = h->ar.take<double>((size_t)conv_tm_stat_sub() * B * 8 * kStatWords);
    if (x6) {
        h->xp3 = h->ar.take<unsigned short>(xp3n);
        h->w3 = h->ar.take<unsigned short>(w3n);
    }
    if (!h->stats || !h->plan.ok || (x6 && !h->w3)) {
        set_error("convtm: allocation failed or unsupported tap pattern");
        return fail(AFTER_E_INVALID);
    }
    if ((rc = pack_conv_weight(w, nullptr, pk, Cout, Cin, k, pad16(Cin), 0)) != AFTER_OK) return fail(rc);
    if ((rc = conv_tm_repack(pk, h->w, h->in, h->plan, 0)) != AFTER_OK) return fail(rc);
    if (x6 && (rc = conv_x6_split(h->w, h->w3, h->in, h->plan, 0)) != AFTER_OK) return fail(rc);
    if (bias) (void)hipMemcpy(h->bias, bias, Cout * sizeof(float), hipMemcpyDeviceToDevice);
    else (void)hipMemset(h->bias, 0, Cout * sizeof(float));
    (void)hipMemset(h->res, 0, yn * sizeof(float));
    (void)hipMemset(h->xtm, 0, (size_t)B * T * Cin * sizeof(float));
    if (hipDeviceSynchronize() != hipSuccess) return fail(AFTER_E_HIP);
    *out = h;
    return AFTER_OK;
}

// mode bit 0: act_pad (x: [B][Cin][T] when given, else the handle's time-major scratch);
// bit 1: conv (y: [B][Cout][Tout] when given, else time-major into the handle's scratch);
// bit 2: accumulate GroupNorm statistics in the conv epilogue; bit 3: add a (zero) residual;
// bit 4: the bf16-pipe path (conv_x6.hip: act_pad writes planes, the conv runs as split-bf16 MFMAs) -- refused
// where the layer is not eligible.
extern "C" int after_convtm_run(after_convtm* h, const float* x, float* y, int mode, void* stream) {
    using namespace after;
    AFTER_REQUIRE(h, AFTER_E_INVALID, "null handle");
    hipStream_t s = (hipStream_t)stream;
    const bool x6 = (mode & 16) != 0;
    AFTER_REQUIRE(!x6 || h->w3, AFTER_E_INVALID, "convtm: this layer has no bf16-pipe form (stride 1, <= 3 taps, Cout %% 4 == 0)");
    if (mode & 1) {
        ActPadTm p;
        memset(&p, 0, sizeof(p));
        p.x = x ? x : h->xtm;
        p.y = h->xp;
        p.y3 = x6 ? h->xp3 : nullptr;
        if (x6 && x) {  // the plane writer takes time-major inputs
            AFTER_TRY(launch_cm_to_tm(x, h->xtm, h->B, h->Cin, h->T, h->Cin, s));
            p.x = h->xtm;
        }
        p.act = h->act;
        p.B = h->B;
        p.C = h->Cin;
        p.T = h->T;
        p.G = h->Cin < 8 ? h->Cin : 8;
        p.x_cm = (x && !x6) ? 1 : 0;
        AFTER_TRY(launch_act_pad_tm(p, s));
    }
    if (mode & 2) {
        ConvTmRun r;
        memset(&r, 0, sizeof(r));
        r.xp = h->xp;
        r.w = h->w;
        r.bias = h->bias;
        r.res = (mode & 8) ? h->res : nullptr;
        r.y = y ? y : h->ytm;
        r.y_cm = y ? 1 : 0;
        r.B = h->B;
        r.Tp = conv_tm_rows(h->T);
        r.Tout = h->Tout;
        r.Nn = h->Tout;
        r.G = h->Cout < 8 ? h->Cout : 8;
        r.sub_stride = h->B * 8 * kStatWords;
        if ((mode & 4) && h->Cout % r.G == 0) {
            AFTER_HIP_CHECK(hipMemsetAsync(h->stats, 0, (size_t)conv_tm_stat_sub() * h->B * 8 * kStatWords * sizeof(double), s));
            r.stats = h->stats;
        }
        if (x6) {
            r.xp3 = h->xp3;
            r.w3 = h->w3;
            r.y = h->ytm;
            r.y_cm = 0;
            AFTER_TRY(launch_conv_x6(r, h->in, h->plan, s));
            if (y) {
                const size_t total = (size_t)h->B * h->Tout * h->Cout;
                hipLaunchKernelGGL(tm_to_cm_kernel, dim3((unsigned)cdivll((long long)total, 256)), dim3(256), 0, s, h->ytm, y, h->Cout,
                                   h->Tout, total);
                AFTER_HIP_CHECK(hipGetLastError());
            }
            return AFTER_OK;
        }
        AFTER_TRY(launch_conv_tm(r, h->in, h->plan, s));
    }
    return AFTER_OK;
}
extern "C" long long after_conv1_act_launches(void) { return after::conv1_act_launches(); }
