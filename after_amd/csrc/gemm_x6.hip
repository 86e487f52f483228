// fp32 GEMM through the bf16 matrix pipe -- EXPERIMENTAL: reachable only through the diagnostic entry
// after_gemm_x6 (tests/test_gemm_gpu.py, scripts/bench_gemm_x6.py); no product path calls it.
//
// C[M,N] = epilogue(A[M,K] * W[N,K]^T), fp32 in and out as in gemm.hip.  Every fp32 operand is the
// exact sum of three bf16 numbers (8 + 8 + 8 significand bits; bf16 has fp32's exponent range, so the
// split needs no scaling and cannot overflow): x = h + m + l.  Of the nine piece products the six
// largest -- hh, hm, mh, mm, hl, lh -- are accumulated in fp32 by v_mfma_f32_16x16x32_bf16; each
// bf16 x bf16 product is exact in fp32, so the only roundings are the accumulator's, as in the fp32
// MFMA chain of gemm.hip.  Measured against fp64 (scripts/ubench/bf16_split.hip, K = 512, activations
// with outliers): max error 1.0e-5 against 2.8e-5 for the fp32 MFMA chain.  The bf16 pipe issues a
// 16x16x32 MFMA in 16 cycles (2.37 PFLOP/s measured) against 32 cycles for the fp32 16x16x4 one
// (155 TFLOP/s): six of them per 32-deep step against eight = 2.67x the fp32 MFMA rate.
//
// W is split once (after_gemm_x6_split: [N][3][K] bf16 planes); A stays fp32 in HBM and in LDS and is
// split into its three planes as the fragments are read (11 VALU ops per two floats), so no producer
// changes.  Pipeline: double-buffered LDS stages filled by LDS-DMA (A rows of 128 B with the 8-chunk
// XOR swizzle of gemm.hip, W plane rows of 64 B with a 4-chunk swizzle), one barrier per 32-deep slab.
//
// Status (round 2): correct (errors at or below the fp32 kernel's on every shape tried).  Best tiles:
// 128 x 96 with eight waves, 58 us at 6144 x 1536 x 512 against 74.4 us for gemm.hip; 48 x 32 with two
// k-parts, 15.4 us against 13.5 us at 768 rows.  A third ring stage is slower (fewer workgroups per CU).
// With the split and five of the six MFMAs disabled the 64 x 96 tile still takes 51 of its 64 us, and the
// per-phase cycle counts (after_gemm_x6_set_debug, scripts/gemm_x6_timeline.py: per slab 530 cycles to
// issue its seven 1-KB LDS-DMA loads, 480 for the 15 fragment reads, 1170 for split + MFMAs of which 576
// are MFMA issue) say why: three co-resident workgroups pull 80 KB per slab round through a CU's 64 B/clk
// load path -- 1250 cycles against 1730 of MFMA issue per SIMD, and the two do not overlap here.  The
// fp32 kernel moves 60 KB per 4600 MFMA cycles.  Making it pay needs >= 128 x 192 tiles (operand bytes per
// MFMA cycle are 4x the fp32 kernel's) in gemm.hip's ring with the side work dealt out behind the MFMAs.
#include <cstdint>
#include <cstdlib>

#include "common.h"
#include "gemm_pipe.h"

namespace after {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_erf_x6(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// round-to-nearest-even bf16 of x, as the high half of a dword (low half zero) -> exact float
__device__ __forceinline__ unsigned bf16_hi(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u & 0xFFFF0000u;
}

// two floats -> the packed (x0 | x1 << 16) bf16 planes h, m, l: x = h + m + l exactly (each remainder is
// exact in fp32; v_cvt_pk_bf16_f32 rounds to nearest even)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2 v = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {x0 - __uint_as_float(h << 16), x1 - __uint_as_float(h & 0xFFFF0000u)};
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    const f32x2 t = {r[0] - __uint_as_float(m << 16), r[1] - __uint_as_float(m & 0xFFFF0000u)};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
}

// a pointer the compiler cannot prove wave-uniform (it is: derived from the wave id) -> SGPR pair
__device__ __forceinline__ const unsigned char* uniform_ptr(const unsigned char* p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const unsigned char*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
}

__global__ void split3_kernel(const float* __restrict__ W, int ldw, unsigned short* __restrict__ W3, int N, int K) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)N * K) return;
    const int n = (int)(idx / K), k = (int)(idx - (size_t)n * K);
    const float x = W[(size_t)n * ldw + k];
    const unsigned h = bf16_hi(x);
    const float r = x - __uint_as_float(h);
    const unsigned m = bf16_hi(r);
    const unsigned l = bf16_hi(r - __uint_as_float(m));
    unsigned short* o = W3 + ((size_t)n * 3) * K + k;
    o[0] = (unsigned short)(h >> 16);
    o[(size_t)K] = (unsigned short)(m >> 16);
    o[2 * (size_t)K] = (unsigned short)(l >> 16);
}

template <int MB, int NB, int KS, int RS, int NS>
__global__ __launch_bounds__(128 * KS * RS) void gemm_x6_kernel(GemmArgs g, const unsigned short* __restrict__ W3,
                                                                int tiles_m, int tiles_n, int xcd_pm) {
    constexpr int BM = 16 * MB, BN = 32 * NB, MT = MB / RS, NT = NB;
    constexpr int NW = 2 * KS * RS;
    constexpr int A_BYTES = BM * 128, W_BYTES = 3 * BN * 64;
    constexpr int PART = A_BYTES + W_BYTES;  // one k-part of a stage
    constexpr int STAGE = KS * PART;
    constexpr int PA = BM / 8, PW = 3 * BN / 16;  // DMA pieces (1 KB each) per k-part
    constexpr int P = KS * (PA + PW);
    constexpr int LPS = (P + NW - 1) / NW;
    static_assert(MB % RS == 0 && BN % 16 == 0 && BM % 8 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    // ---- workgroup -> tile (XCD-aware map of gemm.hip)
    const int nwg = tiles_m * tiles_n;
    int tm, tn;
    if (xcd_pm > 0) {
        const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
        const int pn = 8 / xcd_pm;
        const int cm = tiles_m / xcd_pm, cn = tiles_n / pn;
        const int xi = xcd % xcd_pm, xj = xcd / xcd_pm;
        (void)cn;
        tm = xi * cm + li % cm;
        tn = xj * cn + li / cm;
    } else {
        int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        tn = bid / tiles_m;
        tm = bid - tn * tiles_m;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wid % KS, rp = (wid / KS) % RS, part = wid / (KS * RS);
    const int M = g.M, N = g.N, K = g.K, Kh = K / KS;
    const int nk = Kh / 32;

    // ---- DMA pieces of this wave: wave-uniform base + per-lane byte offset
    unsigned voff[LPS];
    const unsigned char* sbase[LPS];
    unsigned ldst[LPS], sstep[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const int p = wid + NW * i;
        voff[i] = 0;
        sbase[i] = nullptr;
        ldst[i] = 0;
        sstep[i] = 0;
        if (p >= P) continue;
        const int kp = p / (PA + PW), q = p - kp * (PA + PW);
        if (q < PA) {  // 8 rows of A, 128 B each
            const int row = q * 8 + (lane >> 3), pos = lane & 7;
            const int gm = min(m0 + row, M - 1);
            sbase[i] = reinterpret_cast<const unsigned char*>(g.A + (size_t)kp * Kh);
            voff[i] = ((unsigned)gm * (unsigned)g.lda + (unsigned)((pos ^ (row & 7)) * 4)) * 4u;
            ldst[i] = (unsigned)(kp * PART + q * 1024);
            sstep[i] = 128;
        } else {  // 16 rows of one W plane, 64 B each
            const int qq = q - PA, plane = qq / (BN / 16), r0 = (qq - plane * (BN / 16)) * 16;
            const int row = r0 + (lane >> 2), pos = lane & 3;
            const int gn = min(n0 + row, N - 1);
            sbase[i] = reinterpret_cast<const unsigned char*>(W3 + (size_t)kp * Kh);
            voff[i] = (((unsigned)gn * 3u + (unsigned)plane) * (unsigned)K + (unsigned)((pos ^ ((row >> 2) & 3)) * 8)) * 2u;
            ldst[i] = (unsigned)(kp * PART + A_BYTES + (plane * BN + r0) * 64);
            sstep[i] = 64;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem_raw;
    // the per-piece bases are wave-uniform (functions of the wave id): pin them in SGPRs once, so that a DMA
    // costs two scalar adds + s_mov m0 + the load
    unsigned sb_lo[LPS], sb_hi[LPS], sl_dst[LPS], sl_step[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const unsigned long long v = (unsigned long long)(uintptr_t)sbase[i];
        sb_lo[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
        sb_hi[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        sl_dst[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + ldst[i]));
        sl_step[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)sstep[i]);
    }
#define AFTER_X6_ISSUE(slab_, st_)                                                                         \
    _Pragma("unroll") for (int i__ = 0; i__ < LPS; ++i__) {                                                \
        if (wid + NW * i__ < P) {                                                                          \
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                   \
                         :                                                                                 \
                         : "s"(sl_dst[i__] + (unsigned)((st_) * STAGE)), "v"(voff[i__]),                    \
                           "s"((((unsigned long long)sb_hi[i__] << 32) | sb_lo[i__]) +                       \
                               (unsigned long long)((unsigned)(slab_) * sl_step[i__]))                      \
                         : "memory"); /* m0: see gemm.hip */                                               \
        }                                                                                                  \
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, kq = lane >> 4;
    unsigned w_off[NT], a_off[MT][2];  // per-lane LDS byte offsets of the fragments within a k-part of a stage
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int row = part * 16 * NB + j * 16 + frow;
        w_off[j] = (unsigned)(A_BYTES + row * 64 + ((kq ^ ((row >> 2) & 3)) * 16));
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = rp * (BM / RS) + i * 16 + frow;
        a_off[i][0] = (unsigned)(row * 128 + (((2 * kq) ^ (row & 7)) * 16));
        a_off[i][1] = (unsigned)(row * 128 + (((2 * kq + 1) ^ (row & 7)) * 16));
    }
    // pieces this wave moves per slab (ragged when P is not a multiple of the wave count)
    int npw = 0;
#pragma unroll
    for (int i = 0; i < LPS; ++i) npw += (wid + NW * i < P) ? 1 : 0;
    const bool full = npw == LPS;
    // NS-stage ring: slabs s + 1 .. s + NS - 2 stay in flight while slab s is consumed
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) AFTER_X6_ISSUE(s, s)
    unsigned long long t_wait = 0, t_bar = 0, t_issue = 0, t_lds = 0, t_split = 0, t_mma = 0, t0_ = 0, t_begin = 0;
    if (g.dbg) t_begin = __builtin_readcyclecounter();
    for (int s = 0; s < nk; ++s) {
        const int st = s % NS;
        if (g.dbg) t0_ = __builtin_readcyclecounter();
        if (s + NS - 2 <= nk - 1) {  // steady state: NS - 2 later slabs may still be in flight
            if (full) wait_vmcnt_imm<(NS - 2) * LPS>();
            else wait_vmcnt_imm<(NS - 2) * (LPS > 1 ? LPS - 1 : 0)>();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (g.dbg) {
            const unsigned long long t = __builtin_readcyclecounter();
            t_wait += t - t0_;
            t0_ = t;
        }
        __syncthreads();  // slab s has landed for every wave; the stage read in iteration s - 1 is free
        if (g.dbg) {
            const unsigned long long t = __builtin_readcyclecounter();
            t_bar += t - t0_;
            t0_ = t;
        }
        if (s + NS - 1 < nk) AFTER_X6_ISSUE(s + NS - 1, (s + NS - 1) % NS)
        if (g.dbg) {
            const unsigned long long t = __builtin_readcyclecounter();
            t_issue += t - t0_;
            t0_ = t;
        }
        const unsigned char* sa = smem_raw + st * STAGE + kh * PART;
        u32x4 wf[3][NT];
        f32x4 xr[MT][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[p][j] = *reinterpret_cast<const u32x4*>(sa + w_off[j] + p * BN * 64);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            xr[i][0] = *reinterpret_cast<const f32x4*>(sa + a_off[i][0]);
            xr[i][1] = *reinterpret_cast<const f32x4*>(sa + a_off[i][1]);
        }
        if (g.dbg) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long t = __builtin_readcyclecounter();
            t_lds += t - t0_;
            t0_ = t;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            unsigned h_[4], m_[4], l_[4];
            split_pair(xr[i][0][0], xr[i][0][1], h_[0], m_[0], l_[0]);
            split_pair(xr[i][0][2], xr[i][0][3], h_[1], m_[1], l_[1]);
            split_pair(xr[i][1][0], xr[i][1][1], h_[2], m_[2], l_[2]);
            split_pair(xr[i][1][2], xr[i][1][3], h_[3], m_[3], l_[3]);
            const u32x4 ah = {h_[0], h_[1], h_[2], h_[3]}, am = {m_[0], m_[1], m_[2], m_[3]},
                        al = {l_[0], l_[1], l_[2], l_[3]};
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Am = __builtin_bit_cast(bf16x8, am),
                         Al = __builtin_bit_cast(bf16x8, al);
            // W fragment as srcA: the accumulator holds C^T (four consecutive columns of one row per lane).
            // Smallest products first; the NT accumulators of a row block alternate, so that no MFMA
            // waits for the one issued just before it.
#define AFTER_X6_PROD(WP_, AP_)                                                                   \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[WP_][j]), AP_, acc[i][j], 0, 0, 0);
            AFTER_X6_PROD(2, Ah)
            AFTER_X6_PROD(0, Al)
            AFTER_X6_PROD(1, Am)
            AFTER_X6_PROD(1, Ah)
            AFTER_X6_PROD(0, Am)
            AFTER_X6_PROD(0, Ah)
#undef AFTER_X6_PROD
        }
        if (g.dbg) {
            asm volatile("s_nop 0" ::"v"(acc[MT - 1][NT - 1][0]));
            const unsigned long long t = __builtin_readcyclecounter();
            t_mma += t - t0_;
        }
    }
    if (g.dbg && tid == 0) {
        unsigned long long* d = g.dbg + (size_t)blockIdx.x * 8;
        d[0] = t_wait;
        d[1] = t_bar;
        d[2] = t_issue;
        d[3] = t_lds;
        d[4] = t_mma;  // split + MFMAs
        d[5] = __builtin_readcyclecounter() - t_begin;
        d[6] = t_split;
    }
#undef AFTER_X6_ISSUE

    // ---- split-K reduction through LDS in k-part order (bit-deterministic), as in gemm.hip
    float* red = reinterpret_cast<float*>(smem_raw);
    if constexpr (KS > 1) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                *reinterpret_cast<f32x4*>(red + ((wid * MT * NT + i * NT + j) * 64 + lane) * 4) = acc[i][j];
        __syncthreads();
    }
    const int crow = lane & 15, ccol0 = 4 * (lane >> 4);
    const bool vec_ok = ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0) &&
                        (g.epilogue != EPI_RESIDUAL ||
                         (((g.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.R) & 15) == 0)));
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int gn = n0 + part * 16 * NB + j * 16 + ccol0;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (gn + r < N) bv[r] = g.bias[gn + r];
        }
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
            if (g.epilogue == EPI_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = gelu_erf_x6(o[r]);
            } else if (g.epilogue == EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
            } else if (g.epilogue == EPI_SIGMOID) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = 1.0f / (1.0f + expf(-o[r]));
            }
            float* cp = g.C + (size_t)gm * g.ldc + gn;
            if (vec_ok && gn + 3 < N) {
                if (g.epilogue == EPI_RESIDUAL) o += *reinterpret_cast<const f32x4*>(g.R + (size_t)gm * g.ldr + gn);
                *reinterpret_cast<f32x4*>(cp) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (gn + r < N) {
                        float v = o[r];
                        if (g.epilogue == EPI_RESIDUAL) v += g.R[(size_t)gm * g.ldr + gn + r];
                        cp[r] = v;
                    }
            }
        }
    }
}

template <int MB, int NB, int KS, int RS, int NS = 2>
int launch_x6(const GemmArgs& g, const unsigned short* W3, hipStream_t stream) {
    constexpr int BM = 16 * MB, BN = 32 * NB;
    const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
    const size_t stage = (size_t)KS * (BM * 128 + 3 * BN * 64);
    const size_t red = KS > 1 ? (size_t)2 * KS * MB * NB * 256 * sizeof(float) : 0;
    const size_t lds = NS * stage > red ? NS * stage : red;
    static_assert((size_t)NS * KS * (BM * 128 + 3 * BN * 64) <= 160 * 1024, "stages exceed the LDS");
    static_assert(128 * KS * RS <= 1024, "too many waves");
    static bool attr_set = false;
    if (!attr_set) {
        AFTER_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6_kernel<MB, NB, KS, RS, NS>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    int pm = 0;
    double best = 0;
    for (int c = 1; c <= 8; c *= 2) {
        if (tiles_m % c || tiles_n % (8 / c)) continue;
        const double cost = (double)g.M / c + 1.5 * (double)g.N / (8 / c);
        if (pm == 0 || cost < best) {
            pm = c;
            best = cost;
        }
    }
    hipLaunchKernelGGL((gemm_x6_kernel<MB, NB, KS, RS, NS>), dim3(tiles_m * tiles_n), dim3(128 * KS * RS), lds, stream, g,
                       W3, tiles_m, tiles_n, pm);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

}  // namespace

int gemm_x6_split(const float* W, int ldw, unsigned short* W3, int N, int K, hipStream_t s) {
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)cdivll((long long)N * K, 256)), dim3(256), 0, s, W, ldw, W3, N, K);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

// tile: 0 = by shape; else 100 * MB + 10 * NB + KS with RS = 2 when KS == 1
int launch_gemm_x6(const GemmArgs& g, const unsigned short* W3, int tile, hipStream_t stream) {
    AFTER_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && W3, AFTER_E_INVALID, "gemm_x6: bad problem");
    AFTER_REQUIRE((g.K % 32) == 0 && (g.lda % 4) == 0 && ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)W3 % 16) == 0,
                  AFTER_E_INVALID, "gemm_x6: K %% 32 == 0, lda %% 4 == 0 and 16-byte aligned operands required");
    AFTER_REQUIRE((size_t)g.M * g.lda < (1u << 30) && (size_t)g.N * 3 * g.K < (1u << 31), AFTER_E_INVALID,
                  "gemm_x6: operand too large for 32-bit DMA offsets");
    AFTER_REQUIRE(g.epilogue != EPI_RESIDUAL || g.R != nullptr, AFTER_E_INVALID, "gemm_x6: residual epilogue without R");
    if (tile == 0) {
        const bool long_k = g.K >= 2 * g.N;
        if (g.M >= 1536) tile = long_k ? 631 : 831;
        else tile = (g.K % 64 == 0) ? (long_k ? 312 : 332) : 431;
    }
    if (tile % 10 > 1 && tile < 1000) AFTER_REQUIRE(g.K % (32 * (tile % 10)) == 0, AFTER_E_INVALID, "gemm_x6: K not divisible by the k-parts");
    switch (tile) {
        case 332: return launch_x6<3, 3, 2, 1>(g, W3, stream);
        case 312: return launch_x6<3, 1, 2, 1>(g, W3, stream);
        case 314: return launch_x6<3, 1, 4, 1>(g, W3, stream);
        case 322: return launch_x6<3, 2, 2, 1>(g, W3, stream);
        case 431: return launch_x6<4, 3, 1, 2>(g, W3, stream);
        case 421: return launch_x6<4, 2, 1, 2>(g, W3, stream);
        case 631: return launch_x6<6, 3, 1, 2>(g, W3, stream);
        case 1431: return launch_x6<4, 3, 1, 2, 3>(g, W3, stream);  // 1000 + tile: three stages
        case 1631: return launch_x6<6, 3, 1, 2, 3>(g, W3, stream);
        case 1332: return launch_x6<3, 3, 2, 1, 3>(g, W3, stream);
        case 1312: return launch_x6<3, 1, 2, 1, 3>(g, W3, stream);
        case 831: return launch_x6<8, 3, 1, 4, 2>(g, W3, stream);   // 128 x 96, 8 waves
        case 1831: return launch_x6<8, 3, 1, 4, 3>(g, W3, stream);
        case 861: return launch_x6<8, 6, 1, 4, 2>(g, W3, stream);   // 128 x 192, 8 waves
        default:
            set_error("gemm_x6: no tile %d", tile);
            return AFTER_E_INVALID;
    }
}

}  // namespace after

// diagnostics / tests: W3 = the three bf16 planes of W ([N][3][K] unsigned short), C = epi(A W^T + bias)
extern "C" int after_gemm_x6_split(const float* W, int ldw, unsigned short* W3, int N, int K, void* stream) {
    AFTER_REQUIRE(W && W3 && N > 0 && K > 0, AFTER_E_INVALID, "gemm_x6_split: bad argument");
    return after::gemm_x6_split(W, ldw, W3, N, K, (hipStream_t)stream);
}

static unsigned long long* g_x6_dbg = nullptr;
extern "C" void after_gemm_x6_set_debug(unsigned long long* dbg) { g_x6_dbg = dbg; }

extern "C" int after_gemm_x6(const float* A, int lda, const unsigned short* W3, const float* bias, const float* R,
                             int ldr, float* C, int ldc, int M, int N, int K, int epilogue, int tile, void* stream) {
    after::GemmArgs g{A, lda, nullptr, 0, bias, R, ldr, C, ldc, M, N, K, epilogue};
    g.dbg = g_x6_dbg;
    return after::launch_gemm_x6(g, W3, tile, (hipStream_t)stream);
}
