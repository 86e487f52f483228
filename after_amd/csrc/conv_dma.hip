// Second-generation conv path of the codec: "activate once, convolve by DMA".
//
// conv.hip applies GroupNorm + SnakeBeta while it stages the input tile, which costs one
// sinf per element PER OUTPUT-CHANNEL TILE (6x for the 384-channel layers) and forces the
// tile through VGPRs.  Here the normalised + activated tensor is materialised ONCE by
// act_pad_kernel into a zero-haloed scratch tensor (so conv padding is "read the halo",
// exactly the reference's pad-after-activation semantics, cached_conv.Conv1d), and the
// convolution becomes a pure implicit GEMM whose K slabs -- KC input channels x XW samples
// and BM output channels x (taps*KC) packed weights -- are streamed into LDS with
// global_load_lds_dwordx4 (no VGPR round trip) in a 2-deep ring, one raw s_barrier per
// channel block, fragment reads in inline asm (see gemm.hip for why).  The GroupNorm
// statistics of the NEXT layer are accumulated by this kernel's epilogue (row sums ->
// LDS -> one fp64 atomic pair per (workgroup, group)) so the separate full-tensor
// statistics pass disappears.
#include <cstdio>
#include <cstdlib>

#include "conv.h"

namespace after {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int HALO = 32;  // zero samples on both sides of every activated row

__device__ __forceinline__ float act_apply(float v, int act, float pa, float pb) {
    switch (act) {
        case ACT_SNAKE: {
            const float s = sinf(v * pa);
            return v + pb * (s * s);
        }
        case ACT_SILU:
            return v / (1.0f + expf(-v));
        case ACT_RELU:
            return fmaxf(v, 0.f);
        case ACT_TANH:
            return tanhf(v);
        default:
            return v;
    }
}

// y[b, c, HALO + t] = act(x[b, c, t] * scale + shift), halo and tail slack zeroed.
// scale/shift come from the fp64 statistics accumulated by the producer's epilogue
// (sum, sum of squares per (b, group)): nn.GroupNorm, biased variance, eps 1e-5.
struct ActArgs {
    const float* x;
    float* y;
    const double* stats;  // [B][G][2] or nullptr
    const float* gamma;   // with stats: GroupNorm weight; without: per-channel scale (or nullptr)
    const float* beta;    // with stats: GroupNorm bias;   without: per-channel shift
    const float* act_a;
    const float* act_b;
    const float* state;   // streaming: [B][C][HALO] activated samples preceding this chunk, or nullptr
    int act, C, T, Tp, G;
    float eps;
};

__global__ __launch_bounds__(256) void act_pad_kernel(ActArgs a) {
    const int c = blockIdx.y, b = blockIdx.z;
    const float* xr = a.x + ((size_t)b * a.C + c) * a.T;
    float* yr = a.y + ((size_t)b * a.C + c) * a.Tp;
    float sc = 1.f, sh = 0.f;
    if (a.stats) {
        const int Cg = a.C / a.G, g = c / Cg;
        const double n = (double)Cg * a.T;
        const double s = a.stats[((size_t)b * a.G + g) * 2], q = a.stats[((size_t)b * a.G + g) * 2 + 1];
        const double mean = s / n;
        double var = q / n - mean * mean;
        var = var < 0 ? 0 : var;
        const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
        sc = rstd * a.gamma[c];
        sh = a.beta[c] - (float)mean * sc;
    } else if (a.gamma) {  // BatchNorm(eval) folded to a per-channel affine
        sc = a.gamma[c];
        sh = a.beta[c];
    }
    const float pa = a.act_a ? a.act_a[c] : 0.f, pb = a.act_b ? a.act_b[c] : 0.f;
    // positions in the padded row, 4 per thread
    const int p0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= a.Tp) return;
    float4 o;
    float* op = reinterpret_cast<float*>(&o);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = p0 + k - HALO;
        float v = 0.f;
        if (t >= 0 && t < a.T) v = act_apply(xr[t] * sc + sh, a.act, pa, pb);
        else if (t < 0 && a.state) v = a.state[((size_t)b * a.C + c) * HALO + (HALO + t)];
        op[k] = v;
    }
    *reinterpret_cast<float4*>(yr + p0) = o;
}

// streaming: state[b][c][j] = ypad[b][c][T + j], j < HALO  (the last HALO activated samples,
// old context included when the chunk is shorter than the halo)
__global__ void state_update_kernel(const float* __restrict__ yp, float* __restrict__ state, int C,
                                    int T, int Tp, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = idx % HALO;
    const int bc = idx / HALO;
    state[idx] = yp[(size_t)bc * Tp + T + j];
}

struct ConvDmaGeom {
    int tiles_m, tiles_n, KC, XW, LD, nstage, xpieces, wpieces;
    int tmin[kMaxPhases];
};

// xp: [B][Cin][Tp] activated + haloed input.  w: [phase][stage][Cout_pad][LD] (Cout_pad =
// tiles_m * BM rows, zero rows past Cout).
struct ConvDmaArgs {
    const float* xp;
    const float* w;
    const float* bias;
    const float* res;
    float* y;
    double* stats;  // [B][G][2] accumulators of y (or nullptr)
    int B, Cin, Cout, Tp, Tout, taps, phases, istride, ostride, Nn, G;
    int dbg;  // AFTER_CONV_DBG (timing experiments, results invalid): 1 no DMA after stage 0, 2 no fragment reads, 4 no epilogue
    int toff[kMaxPhases][kMaxTaps];
    int ooff[kMaxPhases];
};

template <int MT, int NT>
__global__ __launch_bounds__(256) void conv_dma_kernel(ConvDmaArgs a, ConvDmaGeom gm) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KC = gm.KC, XW = gm.XW, LD = gm.LD;
    const int XS = gm.xpieces * 256;      // floats reserved for the X tile (whole DMA pieces)
    const int STAGE = XS + gm.wpieces * 256;
    __shared__ float gsum[2][8];          // per-group partial sums of this tile (stats epilogue)

    const int nwg = gm.tiles_m * gm.tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tn = bid / gm.tiles_m, tm = bid - tn * gm.tiles_m;
    const int ph = blockIdx.y, b = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wid >> 1) * (16 * MT), wn0 = (wid & 1) * (16 * NT);
    const int taps = a.taps;
    const int tmin = gm.tmin[ph];
    // first input sample of the tile in padded coordinates, rounded down to 16 bytes
    const int tfirst = n0 * a.istride + tmin + HALO;
    const int t0 = tfirst & ~3, skew = tfirst - t0;

    // ---- DMA maps: piece p covers linear 16-byte chunks [64 p, 64 p + 64) of a tile
    const int xcpr = XW >> 2;  // chunks per X row
    const float* xb = a.xp + (size_t)b * a.Cin * a.Tp;
    const int wcpr = LD >> 2;
    const float* wb = a.w + (size_t)ph * gm.nstage * (gm.tiles_m * BM) * LD + (size_t)m0 * LD;
    const int xtot = KC * xcpr, wtot = BM * wcpr;
    // Per-lane source offsets of this wave's pieces, computed ONCE: the per-stage issue is then an
    // add / min per piece.  (Recomputing row = idx / xcpr per piece and stage cost ~300 VALU
    // instructions per stage and wave -- and a wave's VALU work is lost matrix-pipe time.)
    constexpr int XP = 4, WP = 12;  // pieces per wave: KC * XW <= 4096 floats -> <= 16 X pieces;
                                    // BM * LD / 256 <= 42 W pieces for taps * KC <= 128
    int xrow[XP], xcol[XP];
    unsigned woffs[WP];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        int idx = (wid + 4 * i) * 64 + lane;
        idx = idx < xtot ? idx : xtot - 1;  // tail lanes re-fetch the last chunk into slack
        xrow[i] = idx / xcpr;
        xcol[i] = t0 + (idx - xrow[i] * xcpr) * 4;
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        int idx = (wid + 4 * i) * 64 + lane;
        idx = idx < wtot ? idx : wtot - 1;
        woffs[i] = (unsigned)idx * 4u;
    }
    auto issue = [&](int st, int slot) {
        float* base = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int p = wid + 4 * i;
            if (p < gm.xpieces) {
                int c = st * KC + xrow[i];
                c = c < a.Cin ? c : a.Cin - 1;  // rows past Cin meet zero weights
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(xb + (size_t)c * a.Tp + xcol[i]),
                                                 (lds_ptr_t)(base + p * 256), 16, 0, 0);
            }
        }
        const float* ws = wb + (size_t)st * (gm.tiles_m * BM) * LD;
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const int p = wid + 4 * i;
            if (p < gm.wpieces)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ws + woffs[i]), (lds_ptr_t)(base + XS + p * 256),
                                                 16, 0, 0);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, kq = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int nstage = gm.nstage;
    issue(0, 0);
    for (int st = 0; st < nstage; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of stage st landed
        __builtin_amdgcn_s_barrier();                      // everyone's share landed; slot (st+1)&1 free
        asm volatile("" ::: "memory");
        if (st + 1 < nstage && !(a.dbg & 1)) issue(st + 1, (st + 1) & 1);
        const unsigned xs = lds0 + ((st & 1) * STAGE) * 4;
        const unsigned wsb = xs + XS * 4;
        // k-blocks of 16 input channels x 1 tap, flattened over (tap, c16): the fragments of
        // k-block kb+1 are read from LDS while the MFMAs of k-block kb run (two register sets,
        // loop unrolled by two); only the first read of a stage is exposed.
        const int nkb = taps * (KC >> 4);
        const unsigned wrow = wsb + ((wm0 + frow) * LD + kq * 4) * 4;
        const unsigned xcol = xs + (kq * 4 * XW + (wn0 + frow) * a.istride + skew - tmin) * 4;
        f32x4 wa[2][MT];
        float xv[2][NT][4];
        int tap_n = 0, c16_n = 0;  // (tap, c16) of the next k-block to load
#define CONV_LOAD(p)                                                                                 \
    {                                                                                                \
        const unsigned wo__ = wrow + (tap_n * KC + c16_n) * 4;                                       \
        const unsigned xo__ = xcol + (c16_n * XW + a.toff[ph][tap_n]) * 4;                           \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                               \
            asm volatile("ds_read_b128 %0, %1" : "=v"(wa[p][i]) : "v"(wo__ + i * 16 * LD * 4));       \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                               \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                            \
                asm volatile("ds_read_b32 %0, %1"                                                    \
                             : "=v"(xv[p][j][t])                                                     \
                             : "v"(xo__ + (t * XW + j * 16 * a.istride) * 4));                       \
        c16_n += 16;                                                                                 \
        if (c16_n >= KC) {                                                                           \
            c16_n = 0;                                                                               \
            ++tap_n;                                                                                 \
        }                                                                                            \
    }
#define CONV_FENCE(p)                                                                                \
    {                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(wa[p][i]));            \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                               \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(xv[p][j][t]));      \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    }
#define CONV_MMA(p)                                                                                  \
    _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                    \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                               \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                           \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[p][j][t], wa[p][i][t], acc[i][j], 0, 0, 0);
        const bool rd = !(a.dbg & 2);
        if (rd) CONV_LOAD(0)
        for (int kb = 0; kb < nkb; kb += 2) {
            CONV_FENCE(0)
            if (kb + 1 < nkb && rd) CONV_LOAD(1)
            CONV_MMA(0)
            if (kb + 1 < nkb) {
                CONV_FENCE(1)
                if (kb + 2 < nkb && rd) CONV_LOAD(0)
                CONV_MMA(1)
            }
        }
#undef CONV_LOAD
#undef CONV_FENCE
#undef CONV_MMA
    }

    // ---- epilogue
    if (a.dbg & 4) {
        if (acc[0][0][0] == 123.456f) a.y[0] = 0.f;  // keep the accumulators alive
        return;
    }
    // The X fragment is fed as srcA, so the accumulator holds the transposed tile: lane l owns
    // output channel (l & 15) of W block i and four CONSECUTIVE positions 4 (l >> 4) + r of X block
    // j -> one float4 store / residual load per block, and the per-channel sums of the next
    // GroupNorm need only two cross-row shuffles.
    const int cl = lane & 15, tq = 4 * (lane >> 4);
    float* yb = a.y + (size_t)b * a.Cout * a.Tout;
    const float* rb = a.res ? a.res + (size_t)b * a.Cout * a.Tout : nullptr;
    const int Cg = a.stats ? a.Cout / a.G : 1;
    if (a.stats && tid < 16) gsum[tid >> 3][tid & 7] = 0.f;
    if (a.stats) __syncthreads();
    const bool vec = a.ostride == 1 && a.ooff[ph] == 0 && (a.Tout & 3) == 0;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int co = m0 + wm0 + i * 16 + cl;
        float rs = 0.f, rq = 0.f;
        if (co < a.Cout) {
            const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn0 + j * 16 + tq;
                f32x4 v = acc[i][j];
                v += bv;
                if (vec && n + 3 < a.Nn) {
                    const size_t o = (size_t)co * a.Tout + n;
                    if (rb) v += *reinterpret_cast<const f32x4*>(rb + o);
                    *reinterpret_cast<f32x4*>(yb + o) = v;
                    rs += (v[0] + v[1]) + (v[2] + v[3]);
                    rq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int to = (n + r) * a.ostride + a.ooff[ph];
                        if (n + r < a.Nn && to < a.Tout) {
                            float u = v[r];
                            if (rb) u += rb[(size_t)co * a.Tout + to];
                            yb[(size_t)co * a.Tout + to] = u;
                            rs += u;
                            rq += u * u;
                        }
                    }
                }
            }
        }
        if (a.stats) {
            // sum over the four position groups (lanes l, l ^ 16, l ^ 32, l ^ 48), then one LDS
            // atomic per channel
            rs += __shfl_xor(rs, 16, 64);
            rq += __shfl_xor(rq, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
            rq += __shfl_xor(rq, 32, 64);
            if (lane < 16 && co < a.Cout) {
                const int gl = co / Cg - m0 / Cg;  // group index local to the tile (< 8)
                atomicAdd(&gsum[0][gl & 7], rs);
                atomicAdd(&gsum[1][gl & 7], rq);
            }
        }
    }
    if (a.stats) {
        __syncthreads();
        if (tid < 8) {
            const int g = m0 / Cg + tid;
            const int glast = (min(m0 + BM, a.Cout) - 1) / Cg;
            if (g <= glast) {
                double* sp = a.stats + ((size_t)b * a.G + g) * 2;
                atomicAdd(sp, (double)gsum[0][tid]);
                atomicAdd(sp + 1, (double)gsum[1][tid]);
            }
        }
    }
}

// packed weights for the DMA kernel: out[phase][stage][Cout_pad][LD], element
// (co, tap, ci) of stage st at [st][co][tap * KC + ci], zero padded
__global__ void repack_dma_kernel(const float* __restrict__ w /*[phase][Cout][taps][Cin_pad]*/,
                                  float* __restrict__ out, int phases, int Cout, int Cout_pad, int taps,
                                  int Cin, int Cin_pad, int KC, int LD, int nstage) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)phases * nstage * Cout_pad * LD;
    if (idx >= total) return;
    const int k = idx % LD;
    const int co = (idx / LD) % Cout_pad;
    const int st = (idx / ((size_t)LD * Cout_pad)) % nstage;
    const int ph = idx / ((size_t)LD * Cout_pad * nstage);
    float v = 0.f;
    if (k < taps * KC && co < Cout) {
        const int tap = k / KC, ci = st * KC + (k - tap * KC);
        if (ci < Cin) v = w[(((size_t)ph * Cout + co) * taps + tap) * Cin_pad + ci];
    }
    out[idx] = v;
}

__global__ __launch_bounds__(256) void stats_accum_kernel(const float* __restrict__ x,
                                                          double* __restrict__ stats, int C, int T,
                                                          int G, int splits) {
    __shared__ float sh[2][4];
    const int split = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const int Cg = C / G;
    const size_t n = (size_t)Cg * T;
    const float* base = x + ((size_t)b * C + (size_t)g * Cg) * T;
    const size_t per = (n + splits - 1) / splits;
    const size_t lo = (size_t)split * per, hi = lo + per < n ? lo + per : n;
    float s = 0.f, q = 0.f;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        const float v = base[i];
        s += v;
        q += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = s;
        sh[1][threadIdx.x >> 6] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* sp = stats + ((size_t)b * G + g) * 2;
        atomicAdd(sp, (double)sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
        atomicAdd(sp + 1, (double)sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    }
}

}  // namespace

int launch_stats_accum(const float* x, double* stats, int B, int C, int T, int G, hipStream_t s) {
    long long sp = (long long)(C / G) * T / 8192;
    const int splits = sp < 1 ? 1 : (sp > 64 ? 64 : (int)sp);
    hipLaunchKernelGGL(stats_accum_kernel, dim3(splits, G, B), dim3(256), 0, s, x, stats, C, T, G, splits);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

// ----------------------------------------------------------------------------- host side
int conv_dma_halo() { return HALO; }

// padded row length for T samples: halo both sides + slack so that the last tile's DMA
// (which always fetches a whole XW-wide window) stays inside the allocation
int conv_dma_row(int T) { return ((T + 2 * HALO + 256 + 3) & ~3); }

int launch_act_pad(const float* x, float* y, const double* stats, const float* gamma,
                   const float* beta, const float* act_a, const float* act_b, int act, int B, int C,
                   int T, int G, hipStream_t s, float* state) {
    ActArgs a{x, y, stats, gamma, beta, act_a, act_b, state, act, C, T, conv_dma_row(T), G, 1e-5f};
    dim3 grid(cdiv(a.Tp, 1024), C, B);
    hipLaunchKernelGGL(act_pad_kernel, grid, dim3(256), 0, s, a);
    AFTER_HIP_CHECK(hipGetLastError());
    if (state) {
        const int total = B * C * HALO;
        hipLaunchKernelGGL(state_update_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, y, state, C, T,
                           a.Tp, total);
        AFTER_HIP_CHECK(hipGetLastError());
    }
    return AFTER_OK;
}

void conv_dma_plan(const ConvDmaPlanIn& in, ConvDmaPlan* p) {
    // tile: >= 2 workgroups per CU where possible (same rule as the GEMM)
    auto wgs = [&](int bm, int bn) {
        return (long long)cdiv(in.Cout, bm) * cdiv(in.Nn_hint, bn) * in.phases * in.B_hint;
    };
    int mt = 1, nt = 1;
    if (wgs(64, 64) >= 512) mt = 2, nt = 2;
    else if (wgs(32, 64) >= 512) mt = 1, nt = 2;
    else if (in.Cout >= 64 && wgs(64, 32) >= 512) mt = 2, nt = 1;
    // (below two workgroups per CU the 32x32 tile spreads the work over more CUs: 768 -> 768 @
    // T = 1024 is 384 tiles of 64x32 = 1.5 per CU, but 768 tiles of 32x32 = 3 per CU)
    if (in.istride > 1 && nt > 1) nt = 1, mt = in.Cout >= 64 ? 2 : 1;  // keep strided tiles narrow
    p->mt = mt;
    p->nt = nt;
    const int BN = 32 * nt;
    int span = 0;
    for (int ph = 0; ph < in.phases; ++ph) {
        int lo = in.toff[ph][0], hi = in.toff[ph][0];
        for (int t = 1; t < in.taps; ++t) {
            lo = in.toff[ph][t] < lo ? in.toff[ph][t] : lo;
            hi = in.toff[ph][t] > hi ? in.toff[ph][t] : hi;
        }
        span = (hi - lo) > span ? (hi - lo) : span;
    }
    int xw = (BN - 1) * in.istride + span + 1 + 3;  // + alignment skew
    xw = (xw + 3) & ~3;
    if ((xw & 7) != 4) xw += 4;                      // XW == 4 (mod 8): conflict-free b32 fragments
    p->XW = xw;
    int kc = 16;
    const int cin16 = pad16(in.Cin);
    while (kc + 16 <= cin16 && in.taps * (kc + 16) <= 128 && (kc + 16) * xw <= 4096) kc += 16;
    p->KC = kc;
    int ld = in.taps * kc;
    ld += ((40 - (ld & 63)) + 64) & 63;              // LD == 40 (mod 64): conflict-free b128 fragments
    p->LD = ld;
    p->nstage = cdiv(in.Cin, kc);
    p->Cout_pad = cdiv(in.Cout, 32 * mt) * 32 * mt;
    p->w_floats = (size_t)in.phases * p->nstage * p->Cout_pad * ld;
}

int conv_dma_repack(const float* packed, float* out, const ConvDmaPlanIn& in, const ConvDmaPlan& p,
                    hipStream_t s) {
    const size_t total = p.w_floats;
    hipLaunchKernelGGL(repack_dma_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, s, packed, out,
                       in.phases, in.Cout, p.Cout_pad, in.taps, in.Cin, pad16(in.Cin), p.KC, p.LD,
                       p.nstage);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

template <int MT, int NT>
static int launch_dma_cfg(const ConvDmaRun& r, const ConvDmaPlanIn& in, const ConvDmaPlan& p, hipStream_t s) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    ConvDmaGeom g;
    g.tiles_m = p.Cout_pad / BM;
    g.tiles_n = cdiv(r.Nn, BN);
    g.KC = p.KC;
    g.XW = p.XW;
    g.LD = p.LD;
    g.nstage = p.nstage;
    g.xpieces = cdiv(p.KC * (p.XW / 4), 64);
    g.wpieces = cdiv(BM * (p.LD / 4), 64);
    AFTER_REQUIRE(g.xpieces <= 16 && g.wpieces <= 48, AFTER_E_INVALID,
                  "conv_dma: tile needs %d + %d DMA pieces (kernel holds 16 + 48)", g.xpieces, g.wpieces);
    for (int ph = 0; ph < kMaxPhases; ++ph) {
        int lo = 0;
        if (ph < in.phases) {
            lo = in.toff[ph][0];
            for (int t = 1; t < in.taps; ++t) lo = in.toff[ph][t] < lo ? in.toff[ph][t] : lo;
        }
        g.tmin[ph] = lo;
    }
    ConvDmaArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = r.xp;
    a.w = r.w;
    a.bias = r.bias;
    a.res = r.res;
    a.y = r.y;
    a.stats = r.stats;
    a.B = r.B;
    a.Cin = in.Cin;
    a.Cout = in.Cout;
    a.Tp = r.Tp;
    a.Tout = r.Tout;
    a.taps = in.taps;
    a.phases = in.phases;
    a.istride = in.istride;
    a.ostride = in.ostride;
    a.Nn = r.Nn;
    a.G = r.G;
    {
        static int dbg = -1;
        if (dbg < 0) {
            const char* e = getenv("AFTER_CONV_DBG");
            dbg = e ? atoi(e) : 0;
        }
        a.dbg = dbg;
    }
    for (int ph = 0; ph < in.phases; ++ph) {
        for (int t = 0; t < in.taps; ++t) a.toff[ph][t] = in.toff[ph][t];
        a.ooff[ph] = in.ooff[ph];
    }
    const size_t lds = (size_t)2 * (g.xpieces + g.wpieces) * 256 * sizeof(float);
    AFTER_REQUIRE(lds <= 150 * 1024, AFTER_E_INVALID, "conv_dma: LDS ring too large (%zu B)", lds);
    static size_t attr = 0;
    if (lds > attr) {
        AFTER_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dma_kernel<MT, NT>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    hipLaunchKernelGGL((conv_dma_kernel<MT, NT>), dim3(g.tiles_m * g.tiles_n, in.phases, r.B), dim3(256),
                       lds, s, a, g);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

int launch_conv_dma(const ConvDmaRun& r, const ConvDmaPlanIn& in, const ConvDmaPlan& p, hipStream_t s) {
    if (p.mt == 2 && p.nt == 2) return launch_dma_cfg<2, 2>(r, in, p, s);
    if (p.mt == 1 && p.nt == 2) return launch_dma_cfg<1, 2>(r, in, p, s);
    if (p.mt == 2 && p.nt == 1) return launch_dma_cfg<2, 1>(r, in, p, s);
    return launch_dma_cfg<1, 1>(r, in, p, s);
}

}  // namespace after
