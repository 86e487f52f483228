// Implicit-GEMM 1-D convolution on the CDNA4 matrix cores + its satellites
// (GroupNorm statistics, weight-norm folding / packing).
//
// Reference ops covered: ConvBlock1d = GroupNorm -> SnakeBeta ->
// dilated Conv1d (after/autoencoder/networks/SimpleNetsStream.py:150-194), 1x1
// convs, Downsample1d (strided, :32-48), Upsample1d (ConvTranspose1d as `f` 2-tap
// phase convolutions, :51-70), V2ConvBlock1D = BatchNorm(eval) -> SiLU -> causal
// Conv1d (after/diffusion/networks/encoder.py:25-71) and the reflect-padded TDNN
// convs of ECAPA (ecapa_encoder.py:12-139).
//
// Mapping: GEMM M = output channels, N = output time positions, K = taps x input
// channels.  Activations stay [B, C, T] with time contiguous (the reference's
// layout), so the B operand of v_mfma_f32_16x16x4_f32 -- lane l supplies
// X[k = l>>4][n = l&15] -- is 16 consecutive time samples of one channel: global
// reads are coalesced along time and the LDS tile [KC channels][XW samples] is read
// with conflict-free ds_read_b32 (XW == 4 mod 8 puts the four k rows of a lane
// group on disjoint bank octets).  A dilated tap is just a column offset into the
// same LDS tile, so each input sample is fetched once per workgroup for all taps.
// The normalisation + activation in front of every conv is applied while the tile
// is written to LDS (GroupNorm / BatchNorm folded to a per-(b, channel) affine),
// i.e. the activated tensor never exists in HBM.  Weights are packed at create time
// to [phase][Cout][tap][Cin] so that the A operand -- lane l supplies
// W[m = l&15][4 consecutive k of k-quad l>>4] -- is one ds_read_b128 from a row
// padded to == 40 (mod 64) floats (conflict free for every b128 service group).
#include <cstdio>
#include <cstdlib>

#include "conv.h"

namespace after {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int XMAX = 14;  // staged input samples per thread per stage (KC*XW <= 3584)
constexpr int WMAX = 8;   // staged weight float4 per thread per stage (BM*taps*KC <= 8192)

__device__ __forceinline__ float apply_act(float v, int act, float pa, float pb) {
    switch (act) {
        case ACT_SNAKE: {  // core.py:217-260: x + sin^2(alpha x) / (beta + 1e-9)
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

struct ConvGeom {
    int tiles_m, tiles_n, KC, XW, WLD;
    int tmin[kMaxPhases];
};

template <int MT, int NT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a, ConvGeom gm) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KC = gm.KC, XW = gm.XW, WLD = gm.WLD;
    float* Xs = smem;             // [KC][XW]
    float* Ws = smem + KC * XW;   // [BM][WLD]   (KC*XW is a multiple of 4 -> 16-byte aligned)

    const int nwg = gm.tiles_m * gm.tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tn = bid / gm.tiles_m, tm = bid - tn * gm.tiles_m;
    const int ph = blockIdx.y, b = blockIdx.z;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid >> 1) * (16 * MT), wn0 = (wid & 1) * (16 * NT);
    const int taps = a.taps, Tin = a.Tin, Cin = a.Cin, Cinp = a.Cin_pad;
    const int tmin = gm.tmin[ph];
    const int t0 = n0 * a.istride + tmin;  // input sample of LDS column 0

    const float* __restrict__ xb = a.x + (size_t)b * a.x_bstride + (size_t)a.x_coff * Tin;
    const float* __restrict__ wp = a.w + (size_t)ph * a.Cout * taps * Cinp;
    const float* __restrict__ x2b =
        a.x2 ? a.x2 + (size_t)b * a.x2_bstride + (size_t)a.x2_coff * Tin : nullptr;

    // ---- per-thread staging maps (constant over the channel-block loop)
    const int xtotal = KC * XW;
    int xg[XMAX], xci[XMAX];
    bool xok[XMAX];
#pragma unroll
    for (int i = 0; i < XMAX; ++i) {
        const int idx = tid + 256 * i;
        const int ci = idx / XW, col = idx - ci * XW;
        int t = t0 + col;
        bool ok = idx < xtotal;
        if (a.pad == PAD_REFLECT) {
            if (t < 0) t = -t;
            if (t >= Tin) t = 2 * (Tin - 1) - t;
            ok = ok && t >= 0 && t < Tin;
        } else {
            ok = ok && t >= 0 && t < Tin;
        }
        xok[i] = ok;
        xci[i] = ci;
        xg[i] = ci * Tin + (ok ? t : 0);
    }
    const int wrow4 = (taps * KC) >> 2;  // float4 per weight row per stage
    const int wtotal4 = BM * wrow4;
    int wg[WMAX], wl[WMAX], wc[WMAX];
    bool wok[WMAX];
#pragma unroll
    for (int i = 0; i < WMAX; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / wrow4, rem = idx - row * wrow4;
        const int kc4 = KC >> 2;
        const int tap = rem / kc4, c4 = (rem - tap * kc4) * 4;
        wok[i] = idx < wtotal4 && (m0 + row) < a.Cout;
        wg[i] = ((m0 + row) * taps + tap) * Cinp + c4;
        wl[i] = row * WLD + tap * KC + c4;
        wc[i] = c4;
        if (idx >= wtotal4) wl[i] = -1;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float xr[XMAX];
    float4 wr[WMAX];
    const int nstage = (Cin + KC - 1) / KC;
    const int frow = lane & 15, kq = lane >> 4;

    // prologue: fetch stage 0
#pragma unroll
    for (int i = 0; i < XMAX; ++i)
        xr[i] = (xok[i] && xci[i] < Cin) ? (xb[xg[i]] + (x2b ? x2b[xg[i]] : 0.f)) : 0.f;
#pragma unroll
    for (int i = 0; i < WMAX; ++i)
        wr[i] = wok[i] ? *reinterpret_cast<const float4*>(wp + wg[i]) : make_float4(0.f, 0.f, 0.f, 0.f);

    for (int st = 0; st < nstage; ++st) {
        const int cb = st * KC;
        // ---- registers -> LDS, applying  act(x * scale + shift)
#pragma unroll
        for (int i = 0; i < XMAX; ++i) {
            const int idx = tid + 256 * i;
            if (idx < xtotal) {
                float v = 0.f;
                const int c = cb + xci[i];
                if (xok[i] && c < Cin) {
                    v = xr[i];
                    if (a.scale)
                        v = v * a.scale[(size_t)b * a.scale_bstride + c] +
                            a.shift[(size_t)b * a.scale_bstride + c];
                    if (a.act != ACT_NONE)
                        v = apply_act(v, a.act, a.act_a ? a.act_a[c] : 0.f, a.act_b ? a.act_b[c] : 0.f);
                }
                Xs[idx] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < WMAX; ++i)
            if (wl[i] >= 0) *reinterpret_cast<float4*>(Ws + wl[i]) = wr[i];
        __syncthreads();
        // ---- prefetch next stage
        if (st + 1 < nstage) {
            const int cn = cb + KC;
#pragma unroll
            for (int i = 0; i < XMAX; ++i)
                xr[i] = (xok[i] && cn + xci[i] < Cin)
                            ? (xb[(size_t)cn * Tin + xg[i]] + (x2b ? x2b[(size_t)cn * Tin + xg[i]] : 0.f))
                            : 0.f;
#pragma unroll
            for (int i = 0; i < WMAX; ++i)
                wr[i] = (wok[i] && cn + wc[i] < Cinp)
                            ? *reinterpret_cast<const float4*>(wp + wg[i] + cn)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // ---- MFMA over taps x 16-channel blocks
        for (int tap = 0; tap < taps; ++tap) {
            const float* Wt = Ws + (wm0 + frow) * WLD + tap * KC + kq * 4;
            const float* Xt = Xs + (kq * 4) * XW + (wn0 + frow) * a.istride + (a.toff[ph][tap] - tmin);
            for (int c16 = 0; c16 < KC; c16 += 16) {
                float4 wa[MT];
                float xbv[NT][4];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    wa[i] = *reinterpret_cast<const float4*>(Wt + i * 16 * WLD + c16);
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        xbv[j][t] = Xt[(c16 + t) * XW + j * 16 * a.istride];
#define AFTER_CONV_STEP(comp, t)                                                                 \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j)   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i].comp, xbv[j][t], acc[i][j], 0, 0, 0);
                AFTER_CONV_STEP(x, 0)
                AFTER_CONV_STEP(y, 1)
                AFTER_CONV_STEP(z, 2)
                AFTER_CONV_STEP(w, 3)
#undef AFTER_CONV_STEP
            }
        }
        __syncthreads();
    }

    // ---- epilogue: row = 4*(lane>>4) + r (output channel), col = lane & 15 (position)
    const int ccol = lane & 15, crow0 = 4 * (lane >> 4);
    float* __restrict__ yb = a.y + (size_t)b * a.y_bstride + (size_t)a.y_coff * a.Tout;
    const float* __restrict__ rb =
        a.res ? a.res + (size_t)b * a.res_bstride + (size_t)a.res_coff * a.Tout : nullptr;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn0 + j * 16 + ccol;
        if (n >= a.Nn) continue;
        const int to = n * a.ostride + a.ooff[ph];
        if (to >= a.Tout) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m0 + wm0 + i * 16 + crow0 + r;
                if (co >= a.Cout) continue;
                float v = acc[i][j][r] + (a.bias ? a.bias[(size_t)b * a.bias_bstride + co] : 0.f);
                if (rb) v += rb[(size_t)co * a.Tout + to];
                if (a.out_act != ACT_NONE) v = apply_act(v, a.out_act, 0.f, 0.f);
                if (a.post_scale)
                    v = v * a.post_scale[(size_t)b * a.post_bstride + co] + a.post_shift[(size_t)b * a.post_bstride + co];
                yb[(size_t)co * a.Tout + to] = v;
            }
        }
    }
}

template <int MT, int NT>
int launch_conv_cfg(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    ConvGeom g;
    g.tiles_m = cdiv(a.Cout, BM);
    g.tiles_n = cdiv(a.Nn, BN);
    int span = 0;
    for (int p = 0; p < a.phases; ++p) {
        int lo = a.toff[p][0], hi = a.toff[p][0];
        for (int t = 1; t < a.taps; ++t) {
            lo = a.toff[p][t] < lo ? a.toff[p][t] : lo;
            hi = a.toff[p][t] > hi ? a.toff[p][t] : hi;
        }
        g.tmin[p] = lo;
        span = (hi - lo) > span ? (hi - lo) : span;
    }
    for (int p = a.phases; p < kMaxPhases; ++p) g.tmin[p] = 0;
    int xw = (BN - 1) * a.istride + span + 1;
    xw += (4 - (xw & 7) + 8) & 7;  // XW == 4 (mod 8)
    g.XW = xw;
    // channel block: as large as the per-thread staging budgets allow (multiple of 16)
    int kc = 16;
    const int cin16 = pad16(a.Cin);
    while (kc + 16 <= cin16 && (kc + 16) * xw <= 256 * XMAX && BM * a.taps * (kc + 16) <= 1024 * WMAX &&
           a.taps * (kc + 16) <= 192)
        kc += 16;
    AFTER_REQUIRE(kc * xw <= 256 * XMAX && BM * a.taps * kc <= 1024 * WMAX, AFTER_E_INVALID,
                  "conv: tile does not fit the staging budget (taps=%d xw=%d)", a.taps, xw);
    g.KC = kc;
    int wld = a.taps * kc;
    wld += ((40 - (wld & 63)) + 64) & 63;  // WLD == 40 (mod 64)
    g.WLD = wld;
    const size_t lds = ((size_t)kc * xw + (size_t)BM * wld) * sizeof(float);
    AFTER_REQUIRE(lds <= 160 * 1024, AFTER_E_INVALID, "conv: LDS tile too large (%zu B)", lds);
    static size_t attr = 0;
    if (lds > attr) {
        AFTER_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<MT, NT>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    dim3 grid(g.tiles_m * g.tiles_n, a.phases, a.B);
    {
        static int log = -1;
        if (log < 0) {
            const char* e = getenv("AFTER_CONV_LOG");
            log = e ? atoi(e) : 0;
        }
        if (log)
            fprintf(stderr, "CONVLOG cfg=%dx%d Cin=%d Cout=%d Tin=%d Nn=%d taps=%d phases=%d istride=%d KC=%d XW=%d lds=%zu wgs=%d gflop=%.3f\n",
                    MT, NT, a.Cin, a.Cout, a.Tin, a.Nn, a.taps, a.phases, a.istride, g.KC, g.XW, lds,
                    g.tiles_m * g.tiles_n * a.phases * a.B,
                    2.0 * a.Cin * a.Cout * a.taps * (double)a.Nn * a.phases * a.B * 1e-9);
    }
    hipLaunchKernelGGL((conv_mfma_kernel<MT, NT>), grid, dim3(256), lds, s, a, g);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

// ------------------------------------------------------------------ GroupNorm stats
__global__ __launch_bounds__(256) void gn_affine_kernel(GnArgs a) {
    __shared__ double sh[2][4];
    __shared__ bool last;
    const int split = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const int Cg = a.C / a.G;
    const size_t n = (size_t)Cg * a.T;  // contiguous: the group's channels are adjacent rows
    const float* base = a.x + ((size_t)b * a.C + (size_t)g * Cg) * a.T;
    const size_t per = (n + a.splits - 1) / a.splits;
    const size_t lo = (size_t)split * per, hi = lo + per < n ? lo + per : n;
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    // float4 main body when aligned
    size_t i = lo + threadIdx.x;
    for (; i + 256 < hi; i += 512) {
        const float v0 = base[i], v1 = base[i + 256];
        s0 += v0;
        q0 += v0 * v0;
        s1 += v1;
        q1 += v1 * v1;
    }
    for (; i < hi; i += 256) {
        const float v0 = base[i];
        s0 += v0;
        q0 += v0 * v0;
    }
    double s = (double)s0 + (double)s1, q = (double)q0 + (double)q1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wid] = s;
        sh[1][wid] = q;
    }
    __syncthreads();
    const size_t slot = ((size_t)b * a.G + g);
    if (threadIdx.x == 0) {
        const double ts = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        const double tq = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        double* p = a.partials + (slot * a.splits + split) * 2;
        // write-through (sc1) stores + release, then take a ticket (cdna guide G16)
        __hip_atomic_store(p, ts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(a.tickets + slot, 1u, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
        last = (t == (unsigned)a.splits - 1);
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!last) return;
    // the last-arriving block of this (b, g) finalises in a fixed order (deterministic)
    __shared__ float mr[2];
    if (threadIdx.x == 0) {
        double ts = 0, tq = 0;
        const double* p = a.partials + slot * a.splits * 2;
        for (int k = 0; k < a.splits; ++k) {
            ts += __hip_atomic_load(p + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tq += __hip_atomic_load(p + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const double mean = ts / (double)n;
        double var = tq / (double)n - mean * mean;
        var = var < 0 ? 0 : var;
        mr[0] = (float)mean;
        mr[1] = (float)(1.0 / sqrt(var + (double)a.eps));
        __hip_atomic_store(a.tickets + slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Cg; c += 256) {
        const int ch = g * Cg + c;
        const float sc = mr[1] * a.gamma[ch];
        a.scale[(size_t)b * a.C + ch] = sc;
        a.shift[(size_t)b * a.C + ch] = a.beta[ch] - mr[0] * sc;
    }
}

__global__ void bn_affine_kernel(const float* w, const float* b, const float* rm, const float* rv,
                                 float* scale, float* shift, int C, int B, float eps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * B) return;
    const int c = i % C;
    const float sc = w[c] / sqrtf(rv[c] + eps);
    scale[i] = sc;
    shift[i] = b[c] - rm[c] * sc;
}

// scale[row] = g[row] / ||v[row, :]||   (torch weight_norm, dim 0)
__global__ __launch_bounds__(256) void wn_scale_kernel(const float* v, const float* g, float* scale,
                                                       int rows, int inner) {
    __shared__ float sh[4];
    const int r = blockIdx.x;
    float q = 0.f;
    for (int i = threadIdx.x; i < inner; i += 256) {
        const float x = v[(size_t)r * inner + i];
        q += x * x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) scale[r] = g[r] / sqrtf(sh[0] + sh[1] + sh[2] + sh[3]);
}

__global__ void pack_conv_kernel(const float* v, const float* scale, float* out, int Cout, int Cin,
                                 int k, int Cin_pad) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)Cout * k * Cin_pad;
    if (idx >= total) return;
    const int ci = idx % Cin_pad;
    const int tap = (idx / Cin_pad) % k;
    const int co = idx / ((size_t)Cin_pad * k);
    float val = 0.f;
    if (ci < Cin) val = v[((size_t)co * Cin + ci) * k + tap] * (scale ? scale[co] : 1.f);
    out[idx] = val;
}

// ConvTranspose1d(k = 2f, stride f, padding p = f/2) as f phase convolutions with 2 taps:
//   y[co, n*f + r] = sum_ci w[ci, co, phi] x[ci, n + c] + w[ci, co, phi + f] x[ci, n + c - 1]
//   with q = r + p, phi = q % f, c = q / f.      out[r][co][tap][ci]: tap 0 <-> x[n + c - 1]
__global__ void pack_convT_kernel(const float* v, const float* scale, float* out, int Cin, int Cout,
                                  int f, int Cin_pad, int pad) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)f * Cout * 2 * Cin_pad;
    if (idx >= total) return;
    const int ci = idx % Cin_pad;
    const int tap = (idx / Cin_pad) % 2;
    const int co = (idx / ((size_t)Cin_pad * 2)) % Cout;
    const int r = idx / ((size_t)Cin_pad * 2 * Cout);
    const int q = r + pad, phi = q % f;
    const int kidx = tap == 0 ? phi + f : phi;
    float val = 0.f;
    if (ci < Cin) val = v[((size_t)ci * Cout + co) * (2 * f) + kidx] * (scale ? scale[ci] : 1.f);
    out[idx] = val;
}

__global__ void inv_beta_kernel(const float* beta, float* out, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < C) out[i] = 1.0f / (beta[i] + 0.000000001f);
}

}  // namespace

int launch_conv(const ConvArgs& a, hipStream_t s) {
    AFTER_REQUIRE(a.taps >= 1 && a.taps <= kMaxTaps && a.phases >= 1 && a.phases <= kMaxPhases,
                  AFTER_E_INVALID, "conv: taps=%d phases=%d unsupported", a.taps, a.phases);
    AFTER_REQUIRE(a.B > 0 && a.Cin > 0 && a.Cout > 0 && a.Nn > 0, AFTER_E_INVALID, "conv: empty");
    AFTER_REQUIRE(a.Cin_pad % 16 == 0 && a.Cin_pad >= a.Cin, AFTER_E_INVALID, "conv: bad Cin_pad");
    // tile choice: >= 2 workgroups per CU where the problem allows it
    auto wgs = [&](int bm, int bn) {
        return (long long)cdiv(a.Cout, bm) * cdiv(a.Nn, bn) * a.phases * a.B;
    };
    if (a.istride == 1 && wgs(64, 64) >= 512) return launch_conv_cfg<2, 2>(a, s);
    if (wgs(32, 64) >= 512 && a.istride == 1) return launch_conv_cfg<1, 2>(a, s);
    if (a.Cout >= 64 && a.istride == 1 && wgs(64, 32) >= 256) return launch_conv_cfg<2, 1>(a, s);
    return launch_conv_cfg<1, 1>(a, s);
}

int gn_splits(int C, int T, int G) {
    const long long n = (long long)(C / G) * T;
    long long s = n / 16384;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return (int)s;
}

int launch_gn_affine(const GnArgs& a, hipStream_t s) {
    AFTER_REQUIRE(a.C % a.G == 0, AFTER_E_INVALID, "groupnorm: C=%d not divisible by G=%d", a.C, a.G);
    hipLaunchKernelGGL(gn_affine_kernel, dim3(a.splits, a.G, a.B), dim3(256), 0, s, a);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

int launch_bn_affine(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                     float* shift, int C, int B, float eps, hipStream_t s) {
    hipLaunchKernelGGL(bn_affine_kernel, dim3(cdiv(C * B, 256)), dim3(256), 0, s, w, b, rm, rv, scale,
                       shift, C, B, eps);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

static int wn_scale(const float* v, const float* g, float** scale_out, int rows, int inner,
                    hipStream_t s) {
    *scale_out = nullptr;
    if (!g) return AFTER_OK;
    float* sc = nullptr;
    AFTER_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&sc), rows * sizeof(float)));
    hipLaunchKernelGGL(wn_scale_kernel, dim3(rows), dim3(256), 0, s, v, g, sc, rows, inner);
    *scale_out = sc;
    return AFTER_OK;
}

int pack_conv_weight(const float* v, const float* g, float* out, int Cout, int Cin, int k,
                     int Cin_pad, hipStream_t s) {
    float* sc = nullptr;
    AFTER_TRY(wn_scale(v, g, &sc, Cout, Cin * k, s));
    const size_t total = (size_t)Cout * k * Cin_pad;
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, s, v, sc,
                       out, Cout, Cin, k, Cin_pad);
    AFTER_HIP_CHECK(hipGetLastError());
    AFTER_HIP_CHECK(hipStreamSynchronize(s));
    if (sc) (void)hipFree(sc);
    return AFTER_OK;
}

int pack_convT_weight(const float* v, const float* g, float* out, int Cin, int Cout, int f,
                      int Cin_pad, hipStream_t s, int pad) {
    if (pad < 0) pad = f / 2;
    float* sc = nullptr;
    AFTER_TRY(wn_scale(v, g, &sc, Cin, Cout * 2 * f, s));
    const size_t total = (size_t)f * Cout * 2 * Cin_pad;
    hipLaunchKernelGGL(pack_convT_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, s, v, sc,
                       out, Cin, Cout, f, Cin_pad, pad);
    AFTER_HIP_CHECK(hipGetLastError());
    AFTER_HIP_CHECK(hipStreamSynchronize(s));
    if (sc) (void)hipFree(sc);
    return AFTER_OK;
}

int snake_inv_beta(const float* beta, float* out, int C, hipStream_t s) {
    hipLaunchKernelGGL(inv_beta_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, beta, out, C);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

}  // namespace after
