// fp32 GEMM through the bf16 matrix pipe: C[M,N] = epilogue(A[M,K] * W[N,K]^T + bias), fp32 result.
//
// The big Linears of the denoiser (reference transformerv2.py:251 qkv, :275-283 MLP) -- the same operation as
// gemm.hip, 2.5x faster in the matrix pipe.  Every fp32 operand is the exact sum of three bf16 numbers
// (8 + 8 + 8 significand bits; bf16 has fp32's exponent range, so the split needs no scaling and cannot
// overflow): x = h + m + l.  Of the nine piece products the six largest -- hh, hm, mh, mm, hl, lh -- are
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16; each bf16 x bf16 product is exact in fp32, so the only
// roundings are the accumulator's, as in the fp32 MFMA chain of gemm.hip (the three dropped products are
// <= 2^-24 of the term, i.e. below the rounding of the product itself).  Measured against fp64 (K = 512,
// activations with outliers): max error 1.0e-5 against 2.8e-5 for the fp32 MFMA chain; tests/test_gemm_gpu.py
// holds every tile to "error vs fp64 <= the fp32 kernel's".  The bf16 pipe issues a 16x16x32 MFMA in 16 cycles
// against 32 for the fp32 16x16x4 one: six per 32-deep step against eight = 2.67x the fp32 MFMA rate; the
// roofline of this kernel is therefore the bf16 peak / 6 (bench.py).
//
// BOTH operands arrive pre-split as bf16 planes in "x6 blocks" (common.h: 1-KB blocks [rows / 16][K / 32][plane],
// each block = one LDS-DMA piece = its own LDS image, so a wave instruction moves 1 KB of contiguous memory): W
// once at create (after_gemm_x6_split), activations by their PRODUCERS -- ln_mod_ln / the attention LayerNorm tail
// (denoiser.hip) and this kernel's own GELU epilogue (OUT3) write the three planes instead of fp32 -- so the
// fragment path of the GEMM has no VALU work at all (round 2's experiment split A while reading its
// fragments: 11 VALU ops per two floats, redone by every column tile; that was as expensive as the MFMAs).
//
// Pipeline = gemm.hip's balanced ring: a ring of NS stages per workgroup, one stage = one 32-deep slab of every
// k-part = [k-part][A planes h,m,l | W planes h,m,l] rows of 64 bytes, filled by LDS-DMA in 1-KB pieces
// (one x6 block: wave-uniform SGPR base + lane x 16 bytes; the blocks' chunk permutation c ^ f(r / 4),
// f = {0,2,3,1}, makes every 16-lane service group of the ds_read_b128 fragment reads touch 16 distinct
// 16-byte slots), one raw s_barrier per slab with counted vmcnt, the
// next slab's DMA pieces and fragment reads dealt out behind individual MFMAs, >= 2 waves per SIMD so that
// one wave's side work hides behind another's MFMAs.  Operand bytes per MFMA cycle are 4x the fp32 kernel's,
// so tiles are sized for the CU's ~64 B/clk L2->LDS path: bytes per slab 192 (BM + BN) against
// 24 BM BN / 256 MFMA cycles per SIMD (tile table and dispatch: launch_gemm_x6).
// k-parts are summed through LDS in a fixed order (bit-deterministic, no atomics).
#include <cstdint>
#include <cstdlib>

#include "common.h"
#include "gemm_pipe.h"
#include "gemm_x6_pipe.h"
#include "gemm_h3_pipe.h"

namespace after {
namespace {


// round-to-nearest-even bf16 of x, as the high half of a dword (low half zero) -> exact float
__device__ __forceinline__ unsigned bf16_hi(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u & 0xFFFF0000u;
}

// fp32 [N][K] (row stride ldw) -> x6 blocks; rows N .. pad16(N) - 1 are zero
__global__ void split3_kernel(const float* __restrict__ W, int ldw, unsigned short* __restrict__ W3, int N, int K) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t rows = x6_rows_padded(N);
    if (idx >= rows * K) return;
    const int n = (int)(idx / K), k = (int)(idx - (size_t)n * K);
    const float x = n < N ? W[(size_t)n * ldw + k] : 0.f;
    const unsigned h = bf16_hi(x);
    const float r = x - __uint_as_float(h);
    const unsigned m = bf16_hi(r);
    const unsigned l = bf16_hi(r - __uint_as_float(m));
    unsigned short* o = W3 + x6_offset(n, 0, k, K);
    o[0] = (unsigned short)(h >> 16);
    o[512] = (unsigned short)(m >> 16);
    o[1024] = (unsigned short)(l >> 16);
}

struct X6Args {
    const unsigned short* A3;  // x6 blocks of A [M][K]
    const unsigned short* W3;  // x6 blocks of W [N][K]
    const float* bias;
    const float* R;
    int ldr;
    float* C;                  // fp32 output [M][ldc]            (OUT3 = 0)
    unsigned short* C3;        // x6 blocks of the output [M][N]  (OUT3 = 1)
    int ldc;
    int M, N, K;
    int epilogue;
    unsigned long long* dbg;
};

// Epilogue shared by the gemm_x6 kernels: k-part reduction through LDS in k-part order (bit-deterministic, as in
// gemm.hip), bias, activation, residual, fp32 or plane (x6 block) stores.  acc: this wave's MT x NT blocks, rows
// row0 + 16 i, columns col0 + 16 j; w0 = first wave of the group of KS waves that share those blocks.
// bias_pre: the wave's bias values already in registers (persistent kernel), else loaded here.
template <int MT, int NT, int KS, int OUT3>
__device__ __forceinline__ void x6_epilogue(const X6Args& g, f32x4 (&acc)[MT][NT], unsigned char* smem_raw, int row0,
                                            int col0, int wid, int lane, int kh, int w0,
                                            const f32x4* bias_pre = nullptr) {
    const int M = g.M, N = g.N;
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
    // accumulator layout (transposed MFMA): lane l holds C[row = l & 15][col = 4 (l >> 4) + r]
    const int crow = lane & 15, ccol0 = 4 * (lane >> 4);
    const bool vec_ok = OUT3 || (((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0) &&
                                    (g.epilogue != EPI_RESIDUAL ||
                                     (((g.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.R) & 15) == 0))));
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int gn = col0 + j * 16 + ccol0;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias_pre) {
            bv = bias_pre[j];
        } else if (g.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (gn + r < N) bv[r] = g.bias[gn + r];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if ((i * NT + j) % KS != kh) continue;
            f32x4 o = acc[i][j];
            if constexpr (KS > 1) {
                o = *reinterpret_cast<const f32x4*>(red + ((w0 * MT * NT + i * NT + j) * 64 + lane) * 4);
#pragma unroll
                for (int q = 1; q < KS; ++q)
                    o += *reinterpret_cast<const f32x4*>(red + (((w0 + q) * MT * NT + i * NT + j) * 64 + lane) * 4);
            }
            const int gm = row0 + i * 16 + crow;
            if (gm >= M || gn >= N) continue;
            o += bv;
            if (g.epilogue == EPI_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = gelu_erf(o[r]);
            } else if (g.epilogue == EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
            } else if (g.epilogue == EPI_SIGMOID) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = 1.0f / (1.0f + expf(-o[r]));
            }
            if constexpr (OUT3) {
                // the next GEMM's A operand: the row's three bf16 planes into the x6 blocks of [M][N]
                if (gn + 3 < N) {
                    x6_store4(g.C3, gm, gn, N, o[0], o[1], o[2], o[3]);
                } else {
                    uint2 ph, pm, pl;
                    x6_split4(o[0], o[1], o[2], o[3], ph, pm, pl);
                    const unsigned hh[2] = {ph.x, ph.y}, mm[2] = {pm.x, pm.y}, ll[2] = {pl.x, pl.y};
                    unsigned short* cp3 = g.C3 + x6_offset(gm, 0, gn, N);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (gn + r < N) {
                            const int sh = (r & 1) * 16;
                            cp3[r] = (unsigned short)(hh[r >> 1] >> sh);
                            cp3[512 + r] = (unsigned short)(mm[r >> 1] >> sh);
                            cp3[1024 + r] = (unsigned short)(ll[r >> 1] >> sh);
                        }
                }
            } else {
                float* cpf = g.C + (size_t)gm * g.ldc + gn;
                if (vec_ok && gn + 3 < N) {
                    if (g.epilogue == EPI_RESIDUAL) o += *reinterpret_cast<const f32x4*>(g.R + (size_t)gm * g.ldr + gn);
                    *reinterpret_cast<f32x4*>(cpf) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (gn + r < N) {
                            float v = o[r];
                            if (g.epilogue == EPI_RESIDUAL) v += g.R[(size_t)gm * g.ldr + gn + r];
                            cpf[r] = v;
                        }
                }
            }
        }
    }
}

template <class C>
__global__ __launch_bounds__(64 * C::NW, C::WPS) void gemm_x6_kernel(X6Args g, int tiles_m, int tiles_n, int xcd_pm) {
    constexpr int BM = C::BM, BN = C::BN, MT = C::MT, NT = C::NT, KS = C::KS, RS = C::RS, NW = C::NW, NS = C::NS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    // ---- workgroup -> tile (XCD-aware pm x pn map of gemm.hip)
    const int nwg = tiles_m * tiles_n;
    int tm, tn;
    if (xcd_pm > 0) {
        const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
        const int pn = 8 / xcd_pm;
        const int cm = tiles_m / xcd_pm, cn = tiles_n / pn;
        const int xi = xcd % xcd_pm, xj = xcd / xcd_pm;
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
    const int kh = wid % KS, rp = (wid / KS) % RS, cp = wid / (KS * RS);  // k-part, row part, column part
    const int M = g.M, N = g.N, K = g.K, Kh = K / KS;
    const int nk = Kh / 32;

    X6State<C> c;
    c.wid = wid;
    c.lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem_raw;
    c.full = !C::RAGGED || (wid + NW * (C::LPS - 1) < C::P);
    // ---- DMA pieces of this wave: piece p = wid + NW i of the stage = k-part p / PPK, 16-row group p % PPK
    // = one x6 block (1 KB contiguous); consecutive slabs of a row group are 3 KB apart
    c.voff = (unsigned)lane * 16u;
    {
        const int kb = K >> 5, kbh = Kh >> 5;  // 32-deep blocks per row group / per k-part
#pragma unroll
        for (int i = 0; i < C::LPS; ++i) {
            int p = wid + NW * i;
            if (p >= C::P) p = C::P - 1;  // never issued (c.full == false)
            const int kp = p / C::PPK, q = p - kp * C::PPK;
            const unsigned short* base;
            if (q < C::GA) {
                const int plane = q / C::MB, grp = q - plane * C::MB;
                const int rb = min((m0 >> 4) + grp, (M - 1) >> 4);  // past the last row: a clamped block, results unused
                base = g.A3 + (((size_t)rb * kb + (size_t)kp * kbh) * 3 + plane) * 512;
            } else {
                const int qq = q - C::GA;
                const int plane = qq / C::NBK, grp = qq - plane * C::NBK;
                const int rb = min((n0 >> 4) + grp, (N - 1) >> 4);
                base = g.W3 + (((size_t)rb * kb + (size_t)kp * kbh) * 3 + plane) * 512;
            }
            // wave-uniform (function of the wave id): pin it in SGPRs
            const unsigned long long v = (unsigned long long)(uintptr_t)base;
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
            c.sb[i] = ((unsigned long long)hi << 32) | lo;
        }
    }
    // ---- fragment addresses: lane l -> row l & 15 of a 16-row block, 16-byte chunk l >> 4 of its 64-byte row
    {
        const int frow = lane & 15, kq = lane >> 4;
        const unsigned sw = (unsigned)((kq ^ swz4((frow >> 2) & 3)) * 16);
        c.a_rd = c.lds0 + (unsigned)(kh * C::PART + (rp * (BM / RS) + frow) * 64) + sw;
        c.w_rd = c.lds0 + (unsigned)(kh * C::PART + C::GA * 1024 + (cp * (BN / C::CP) + frow) * 64) + sw;
    }
#pragma unroll
    for (int q = 0; q <= C::ACC2; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) c.acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    unsigned long long t_start = 0, t_loop = 0, t_end = 0, r_start = 0;
    if (g.dbg) {
        t_start = __builtin_readcyclecounter();
        r_start = wall_clock64();
    }
    // ---- prologue: fill the ring, wait for slab 0, read its fragments
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < nk) x6_issue_slab<C>(c, s, s);
    x6_wait<C>(c, (nk < NS ? nk : NS) - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (g.dbg) t_loop = __builtin_readcyclecounter();
    x6_sides<C, 0, C::LPS, C::NWORK, true>(c, false, true, 0, 0, c.a_rd, c.w_rd);
    int kt = 0;
    for (; kt + 1 + NS < nk; kt += 2) {  // steady state: slab kt + 1 + NS exists
        x6_step<C, 0, true>(c, kt, nk);
        x6_step<C, 1, true>(c, kt + 1, nk);
    }
    for (; kt < nk; kt += 2) {
        x6_step<C, 0, false>(c, kt, nk);
        if (kt + 1 < nk) x6_step<C, 1, false>(c, kt + 1, nk);
    }
    if constexpr (C::ACC2) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) c.acc[0][i][j] += c.acc[1][i][j];
    }
    if (g.dbg) {
        asm volatile("s_nop 0" ::"v"(c.acc[0][MT - 1][NT - 1][0]));
        t_end = __builtin_readcyclecounter();
    }

    x6_epilogue<MT, NT, KS, C::OUT3>(g, c.acc[0], smem_raw, m0 + rp * (BM / RS), n0 + cp * (BN / C::CP), wid, lane, kh,
                                     (cp * RS + rp) * KS);
    if (g.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = g.dbg + (size_t)blockIdx.x * 8;
        d[0] = t_start;
        d[1] = t_loop;
        d[2] = t_end;
        d[3] = __builtin_readcyclecounter();
        d[4] = r_start;
        d[5] = wall_clock64();
        d[6] = __smid();
        d[7] = 0;
    }
}

// =====================================================================================================
// Persistent many-row variant (C::PERSIST): at 6144 token rows a CU owns two or three 192 x 96 tiles, and as separate
// workgroups each of them pays its own cold start (1.5 us until the first slab has landed) and its own store
// drain (3 us until the last store is acknowledged and the LDS / registers are released) with nothing to overlap
// them: one workgroup per CU is resident.  Here a workgroup walks its tiles itself, all in one column tile (the W
// panel stays in the XCD's L2, the bias in registers): after the last slab of tile t it issues the first NS slabs
// of tile t + 1, then runs tile t's epilogue -- whose stores are fire-and-forget -- and resumes MFMAs as soon as
// slab 0 has landed.  vmcnt counts loads and stores alike and retires in order: the wait for slab 0 allows the
// (NS - 1) younger slabs AND the epilogue's stores (issued after the DMAs, a compile-time count on full tiles) to
// stay in flight.  Requires KS = 1, full tiles (M % BM == N % BN == 0), an epilogue without residual, and the
// pm x pn XCD map (launch_x6p falls back to the plain kernel otherwise).
// epilogue of a FULL tile (no edges, aligned outputs, no residual): exactly MT x NT (x 3 for planes) store
// instructions and no other vector-memory operation -- the persistent kernel counts them in its vmcnt waits
template <int MT, int NT, int OUT3>
__device__ __forceinline__ void x6_epilogue_full(const X6Args& g, f32x4 (&acc)[MT][NT], int row0, int col0, int lane,
                                                 const f32x4* bias_pre) {
    const int crow = lane & 15, ccol0 = 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int gn = col0 + j * 16 + ccol0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            f32x4 o = acc[i][j] + bias_pre[j];
            if (g.epilogue == EPI_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = gelu_erf(o[r]);
            } else if (g.epilogue == EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
            } else if (g.epilogue == EPI_SIGMOID) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = 1.0f / (1.0f + expf(-o[r]));
            }
            const int gm = row0 + i * 16 + crow;
            if constexpr (OUT3) x6_store4(g.C3, gm, gn, g.N, o[0], o[1], o[2], o[3]);
            else *reinterpret_cast<f32x4*>(g.C + (size_t)gm * g.ldc + gn) = o;
        }
    }
}

template <class C>
__device__ __forceinline__ void x6_tile_sources(X6State<C>& c, const X6Args& g, int m0, int n0, int wid) {
    const int M = g.M, N = g.N, K = g.K, Kh = K / C::KS;
    const int kb = K >> 5, kbh = Kh >> 5;
#pragma unroll
    for (int i = 0; i < C::LPS; ++i) {
        int p = wid + C::NW * i;
        if (p >= C::P) p = C::P - 1;
        const int kp = p / C::PPK, q = p - kp * C::PPK;
        const unsigned short* base;
        if (q < C::GA) {
            const int plane = q / C::MB, grp = q - plane * C::MB;
            const int rb = min((m0 >> 4) + grp, (M - 1) >> 4);
            base = g.A3 + (((size_t)rb * kb + (size_t)kp * kbh) * 3 + plane) * 512;
        } else {
            const int qq = q - C::GA;
            const int plane = qq / C::NBK, grp = qq - plane * C::NBK;
            const int rb = min((n0 >> 4) + grp, (N - 1) >> 4);
            base = g.W3 + (((size_t)rb * kb + (size_t)kp * kbh) * 3 + plane) * 512;
        }
        const unsigned long long v = (unsigned long long)(uintptr_t)base;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        c.sb[i] = ((unsigned long long)hi << 32) | lo;
    }
}

// as x6_wait, with EXTRA younger vector-memory instructions (the previous tile's stores) allowed on top
template <class C, int EXTRA>
__device__ __forceinline__ void x6_wait_extra(const X6State<C>& c) {
    constexpr int L = C::LPS, L1 = C::LPS > 1 ? C::LPS - 1 : 0;
    static_assert((C::NS - 1) * L + EXTRA < 64, "vmcnt is a 6-bit counter");
    if (!C::RAGGED || c.full) wait_vmcnt_imm<(C::NS - 1) * L + EXTRA>();
    else wait_vmcnt_imm<(C::NS - 1) * L1 + EXTRA>();
}

template <class C>
__global__ __launch_bounds__(64 * C::NW, C::WPS) void gemm_x6p_kernel(X6Args g, int tiles_m, int tiles_n, int xcd_pm,
                                                                      int nwx) {
    static_assert(C::KS == 1 && C::ACC2 == 0 && C::NS == 2, "persistent tiles: no k-parts, two-stage ring");
    constexpr int BM = C::BM, BN = C::BN, MT = C::MT, NT = C::NT, RS = C::RS, NS = C::NS;
    constexpr int STORES = MT * NT * (C::OUT3 ? 3 : 1);  // vector-memory instructions of one full-tile epilogue
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int pn = 8 / xcd_pm;
    const int cm = tiles_m / xcd_pm, cn = tiles_n / pn;
    const int xi = xcd % xcd_pm, xj = xcd / xcd_pm;
    const int nlt = cm * cn;  // tiles of this XCD; workgroup li takes li, li + nwx, ... (column-tile fastest)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rp = wid % RS, cp = wid / RS;
    const int nk = g.K / 32;

    X6State<C> c;
    c.wid = wid;
    c.lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem_raw;
    c.full = !C::RAGGED || (wid + C::NW * (C::LPS - 1) < C::P);
    c.voff = (unsigned)lane * 16u;
    {
        const int frow = lane & 15, kq = lane >> 4;
        const unsigned sw = (unsigned)((kq ^ swz4((frow >> 2) & 3)) * 16);
        c.a_rd = c.lds0 + (unsigned)((rp * (BM / RS) + frow) * 64) + sw;
        c.w_rd = c.lds0 + (unsigned)(C::GA * 1024 + (cp * (BN / C::CP) + frow) * 64) + sw;
    }
    int lt = li;
    int m0 = (xi * cm + lt / cn) * BM, n0 = (xj * cn + lt % cn) * BN;
    // the workgroup stays in one column tile (nwx % cn == 0): its bias slice lives in registers
    f32x4 bias_pre[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        bias_pre[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g.bias) bias_pre[j] = *reinterpret_cast<const f32x4*>(g.bias + n0 + cp * (BN / C::CP) + j * 16 + 4 * (lane >> 4));
    }
    x6_tile_sources<C>(c, g, m0, n0, wid);
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < nk) x6_issue_slab<C>(c, s, s);
    bool first = true;
    while (true) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) c.acc[0][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (first) x6_wait<C>(c, NS - 1);
        else x6_wait_extra<C, STORES>(c);  // slab 0 of this tile: the younger slab and the last tile's stores may fly
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        x6_sides<C, 0, C::LPS, C::NWORK, true>(c, false, true, 0, 0, c.a_rd, c.w_rd);
        int kt = 0;
        if (!first) {
            // slab 1 was issued before the previous tile's stores (still draining, younger: allowed); slab 2 is
            // issued behind them, so from step 1 on the in-order counter needs no allowance (nk >= 4: launch_x6p)
            x6_step<C, 0, true, STORES>(c, 0, nk);
            x6_step<C, 1, true>(c, 1, nk);
            kt = 2;
        }
        first = false;
        for (; kt + 1 + NS < nk; kt += 2) {
            x6_step<C, 0, true>(c, kt, nk);
            x6_step<C, 1, true>(c, kt + 1, nk);
        }
        for (; kt < nk; kt += 2) {
            x6_step<C, 0, false>(c, kt, nk);
            if (kt + 1 < nk) x6_step<C, 1, false>(c, kt + 1, nk);
        }
        const int lt_next = lt + nwx;
        const bool has_next = lt_next < nlt;
        const int m0n = (xi * cm + lt_next / cn) * BM, n0n = (xj * cn + lt_next % cn) * BN;
        if (has_next) {
            __builtin_amdgcn_s_barrier();  // every wave is past its last read of the ring
            asm volatile("" ::: "memory");
            x6_tile_sources<C>(c, g, m0n, n0n, wid);
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (s < nk) x6_issue_slab<C>(c, s, s);
        }
        x6_epilogue_full<MT, NT, C::OUT3>(g, c.acc[0], m0 + rp * (BM / RS), n0 + cp * (BN / C::CP), lane, bias_pre);
        if (!has_next) break;
        lt = lt_next;
        m0 = m0n;
        n0 = n0n;
    }
}

template <class C>
int launch_x6(const X6Args& g, hipStream_t stream);

// persistent launch when the shape allows it, else the plain kernel of the same tile
template <class C, class CPlain>
int launch_x6p(const X6Args& g, hipStream_t stream) {
    const int tiles_m = cdiv(g.M, C::BM), tiles_n = cdiv(g.N, C::BN);
    int pm = 0;
    double best = 0;
    for (int cdv = 1; cdv <= 8; cdv *= 2) {
        if (tiles_m % cdv || tiles_n % (8 / cdv)) continue;
        const double cost = (double)g.M / cdv + (double)g.N / (8 / cdv);
        if (pm == 0 || cost < best) {
            pm = cdv;
            best = cost;
        }
    }
    const bool vec_ok = C::OUT3 || ((g.ldc & 3) == 0 && ((uintptr_t)g.C & 15) == 0);
    const int nk = g.K / 32;
    bool ok = pm > 0 && (g.M % C::BM) == 0 && (g.N % C::BN) == 0 && g.epilogue != EPI_RESIDUAL && vec_ok && !g.dbg &&
              nk >= 6 && (!g.bias || ((uintptr_t)g.bias & 15) == 0);
    static int persist = -1;  // AFTER_GEMM_X6_PERSIST=0: A/B switch (plain kernel of the same tile)
    if (persist < 0) {
        const char* e = getenv("AFTER_GEMM_X6_PERSIST");
        persist = e ? atoi(e) : 1;
    }
    ok = ok && persist;
    int nwx = 0;
    if (ok) {
        const int cm = tiles_m / pm, cn = tiles_n / (8 / pm), nlt = cm * cn;
        // 32 CUs per XCD x RES resident workgroups, a multiple of the column-tile count so that a workgroup keeps
        // its column tile, and no more workgroups than tiles
        nwx = 32 * C::RES;
        if (nwx > nlt) nwx = nlt;
        nwx -= nwx % cn;
        ok = nwx >= cn && nlt > nwx;  // (a single round gains nothing: leave it to the plain kernel)
    }
    if (!ok) return launch_x6<CPlain>(g, stream);
    const size_t lds = (size_t)C::NS * C::STAGE;
    static LdsAttr attr_set;
    AFTER_TRY(ensure_lds_attr(attr_set, reinterpret_cast<const void*>(gemm_x6p_kernel<C>), lds));
    hipLaunchKernelGGL((gemm_x6p_kernel<C>), dim3(8 * nwx), dim3(64 * C::NW), lds, stream, g, tiles_m, tiles_n, pm, nwx);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

// =====================================================================================================
// Few-row variant ("W in registers"): one workgroup per CU, each wave owns ONE 16-column block of W for its
// k-part (NT = 1), so no W element is shared between waves and W has no business in LDS: the wave loads its W
// fragments straight into a ring of D = 4 register slots (x6 blocks are stored in fragment order: a wave
// instruction still reads 1 KB of contiguous memory), three slabs ahead.  Only A -- shared by the NBK column waves --
// goes through the LDS ring, whose stages shrink to KS x 9 KB, so four of them fit and A runs four slabs ahead.
// At 768 rows the LDS-staged tile above could hold ONE slab in flight (2 x 55 KB stages): 54 KB per ~2100-cycle
// memory round trip = 25-40 B/clk per CU against the ~64 B/clk the L2 -> CU path delivers; here ~150 KB are in
// flight per CU (Little's law needs ~130).  Issue order per wave: A0 W0 A1 W1 A2 W2 A3 | step t: W(t+3) A(t+4);
// at the top of step t everything up to A(t+1) must have landed (W(t), older, with it): at most
// {W(t+1), A(t+2), W(t+2), A(t+3)} may still be in flight -- counted vmcnt, one raw s_barrier per slab.
// Requires nk = K / (32 KS) to be a multiple of 4.
template <int MB_, int NBK_, int KS_, int OUT3_>
struct X6WCfg {
    static constexpr int MB = MB_, NBK = NBK_, KS = KS_, OUT3 = OUT3_;
    static constexpr int BM = 16 * MB, BN = 16 * NBK, MT = MB, NW = KS * NBK, D = 4;
    static constexpr int GA = 3 * MB;            // A pieces (1 KB) per k-part per slab
    static constexpr int PART = GA * 1024, STAGE = KS * PART;
    static constexpr int PA = KS * GA;           // A pieces per stage
    static constexpr int LPA = (PA + NW - 1) / NW;
    static constexpr bool RAGGED = (PA % NW) != 0;
    static constexpr int NMMA = 6 * MT, NWORK = 3 + LPA + 3 * MT;
    static constexpr int WPS = (NW + 3) / 4;
    static_assert(D * STAGE <= 160 * 1024, "A ring exceeds the LDS");
    static_assert(NW <= 16 && LPA >= 1, "wave count");
};

template <class C>
struct X6WState {
    f32x4 acc[C::MT][1];
    u32x4 fa[2][3][C::MT];        // A fragments of two consecutive slabs
    u32x4 w[C::D][3];             // W fragments: register ring, slot = slab % 4, [plane]
    unsigned long long sa[C::LPA];  // source of each A piece (slab 0 of its k-part)
    unsigned long long wb;        // this wave's W block row, slab 0 of its k-part, plane h
    unsigned voff_a, voff_w;      // lane x 16 / the lane's fragment position inside an x6 block
    unsigned a_rd, lds0;
    int wid;
    bool full;
};

template <class C>
__device__ __forceinline__ void x6w_dma_a(const X6WState<C>& c, int i, int slab, int stage) {
    lds_dma16(c.lds0 + (unsigned)(stage * C::STAGE + (c.wid + C::NW * i) * 1024), c.voff_a, c.sa[i] + (unsigned long long)((unsigned)slab * 3072u));
}
template <class C>
__device__ __forceinline__ void x6w_issue_a(const X6WState<C>& c, int slab) {
#pragma unroll
    for (int i = 0; i < C::LPA; ++i)
        if (!C::RAGGED || i + 1 < C::LPA || c.full) x6w_dma_a<C>(c, i, slab, slab & 3);
}
template <class C, int SLOT, int PL>
__device__ __forceinline__ void x6w_load_w(X6WState<C>& c, int slab) {
    // register destination: hipcc does not count this load -- the step's vmcnt + x6w_fence_w order its consumers
    asm volatile("global_load_dwordx4 %0, %1, %2"
                 : "=v"(c.w[SLOT][PL])
                 : "v"(c.voff_w), "s"(c.wb + (unsigned long long)((unsigned)slab * 3072u + PL * 1024u))
                 : "memory");
}
template <class C, int SLOT>
__device__ __forceinline__ void x6w_issue_w(X6WState<C>& c, int slab) {
    x6w_load_w<C, SLOT, 0>(c, slab);
    x6w_load_w<C, SLOT, 1>(c, slab);
    x6w_load_w<C, SLOT, 2>(c, slab);
}
template <class C, int SLOT>
__device__ __forceinline__ void x6w_fence_w(X6WState<C>& c) {
    asm volatile("" : "+v"(c.w[SLOT][0]), "+v"(c.w[SLOT][1]), "+v"(c.w[SLOT][2]));
}
template <class C, int SET, int I>
__device__ __forceinline__ void x6w_fence_a(X6WState<C>& c) {
    if constexpr (I < 3 * C::MT) {
        asm volatile("" : "+v"(c.fa[SET][I / C::MT][I % C::MT]));
        x6w_fence_a<C, SET, I + 1>(c);
    }
}
// at most NG "W + A" issue groups, NW3 lone W loads (3 instructions) and NA lone A groups of this wave in flight
template <class C, int NG, int NW3, int NA>
__device__ __forceinline__ void x6w_wait(const X6WState<C>& c) {
    constexpr int L = C::LPA, L1 = C::LPA - 1;
    if (!C::RAGGED || c.full) wait_vmcnt_imm<NG * (3 + L) + 3 * NW3 + NA * L>();
    else wait_vmcnt_imm<NG * (3 + L1) + 3 * NW3 + NA * L1>();
}

// side-work item I of step kt: W loads of slab kt + 3 (slot (kt + 3) & 3), A pieces of slab kt + 4 (stage kt & 3),
// A fragment reads of slab kt + 1 into the other set
template <class C, int CUR, int SLOT, bool ISSUE_W, bool ISSUE_A, bool MORE, int I>
__device__ __forceinline__ void x6w_side(X6WState<C>& c, int kt, unsigned a_next) {
    if constexpr (I < 3) {
        if constexpr (ISSUE_W) x6w_load_w<C, (SLOT + 3) & 3, I>(c, kt + 3);
    } else if constexpr (I < 3 + C::LPA) {
        constexpr int i = I - 3;
        if constexpr (ISSUE_A)
            if (!C::RAGGED || i + 1 < C::LPA || c.full) x6w_dma_a<C>(c, i, kt + 4, SLOT);
    } else {
        constexpr int R = I - 3 - C::LPA, pl = R / C::MT, i = R % C::MT;
        if constexpr (MORE)
            asm volatile("ds_read_b128 %0, %1 offset:%2"
                         : "=v"(c.fa[CUR ^ 1][pl][i])
                         : "v"(a_next), "i"((pl * C::BM + i * 16) * 64));
    }
}
template <class C, int CUR, int SLOT, bool ISSUE_W, bool ISSUE_A, bool MORE, int I, int IEND>
__device__ __forceinline__ void x6w_sides(X6WState<C>& c, int kt, unsigned a_next) {
    if constexpr (I < IEND) {
        x6w_side<C, CUR, SLOT, ISSUE_W, ISSUE_A, MORE, I>(c, kt, a_next);
        x6w_sides<C, CUR, SLOT, ISSUE_W, ISSUE_A, MORE, I + 1, IEND>(c, kt, a_next);
    }
}
template <class C, int CUR, int SLOT, bool ISSUE_W, bool ISSUE_A, bool MORE, int S>
__device__ __forceinline__ void x6w_mma(X6WState<C>& c, int kt, unsigned a_next) {
    if constexpr (S < C::NMMA) {
        constexpr int p = S / C::MT, i = S % C::MT;
        c.acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, c.w[SLOT][kWP[p]]),
                                                              __builtin_bit_cast(bf16x8, c.fa[CUR][kAP[p]][i]),
                                                              c.acc[i][0], 0, 0, 0);
        constexpr int w0 = (S * C::NWORK) / C::NMMA, w1 = ((S + 1) * C::NWORK) / C::NMMA;
        x6w_sides<C, CUR, SLOT, ISSUE_W, ISSUE_A, MORE, w0, w1>(c, kt, a_next);
        __builtin_amdgcn_sched_barrier(0);
        x6w_mma<C, CUR, SLOT, ISSUE_W, ISSUE_A, MORE, S + 1>(c, kt, a_next);
    }
}
// MODE 0: steady (kt + 4 < nk); 1..4: step nk - 4 .. nk - 1
template <class C, int CUR, int SLOT, int MODE>
__device__ __forceinline__ void x6w_step(X6WState<C>& c, int kt) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    x6w_fence_a<C, CUR, 0>(c);
    if constexpr (MODE <= 1) x6w_wait<C, 2, 0, 0>(c);       // W(t+1) A(t+2) W(t+2) A(t+3)
    else if constexpr (MODE == 2) x6w_wait<C, 1, 1, 0>(c);  // W(nk-2) A(nk-1) W(nk-1)
    else if constexpr (MODE == 3) x6w_wait<C, 0, 1, 0>(c);  // W(nk-1)
    else x6w_wait<C, 0, 0, 0>(c);
    x6w_fence_w<C, SLOT>(c);
    __builtin_amdgcn_sched_barrier(0);
    constexpr bool MORE = MODE != 4;
    if constexpr (MORE) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    const unsigned a_next = c.a_rd + (unsigned)(((SLOT + 1) & 3) * C::STAGE);
    x6w_mma<C, CUR, SLOT, (MODE <= 1), (MODE == 0), MORE, 0>(c, kt, a_next);
}

template <class C>
__global__ __launch_bounds__(64 * C::NW, C::WPS) void gemm_x6w_kernel(X6Args g, int tiles_m, int tiles_n, int xcd_pm) {
    constexpr int BM = C::BM, BN = C::BN, MT = C::MT, KS = C::KS, NW = C::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int nwg = tiles_m * tiles_n;
    int tm, tn;
    if (xcd_pm > 0) {
        const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
        const int pn = 8 / xcd_pm;
        const int cm = tiles_m / xcd_pm, cn = tiles_n / pn;
        const int xi = xcd % xcd_pm, xj = xcd / xcd_pm;
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
    const int kh = wid % KS, cp = wid / KS;  // k-part, column block
    const int M = g.M, N = g.N, K = g.K, Kh = K / KS;
    const int nk = Kh / 32, kb = K >> 5, kbh = Kh >> 5;

    X6WState<C> c;
    c.wid = wid;
    c.lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem_raw;
    c.full = !C::RAGGED || (wid + NW * (C::LPA - 1) < C::PA);
    c.voff_a = (unsigned)lane * 16u;
    {
        const int frow = lane & 15, kq = lane >> 4;
        const unsigned sw = (unsigned)((kq ^ swz4((frow >> 2) & 3)) * 16);
        c.voff_w = (unsigned)(frow * 64) + sw;
        c.a_rd = c.lds0 + (unsigned)(kh * C::PART + frow * 64) + sw;
    }
#pragma unroll
    for (int i = 0; i < C::LPA; ++i) {
        int p = wid + NW * i;
        if (p >= C::PA) p = C::PA - 1;  // never issued
        const int kp = p / C::GA, q = p - kp * C::GA;
        const int plane = q / C::MB, grp = q - plane * C::MB;
        const int rb = min((m0 >> 4) + grp, (M - 1) >> 4);
        const unsigned long long v =
            (unsigned long long)(uintptr_t)(g.A3 + (((size_t)rb * kb + (size_t)kp * kbh) * 3 + plane) * 512);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        c.sa[i] = ((unsigned long long)hi << 32) | lo;
    }
    {
        const int rb = min((n0 >> 4) + cp, (N - 1) >> 4);
        const unsigned long long v = (unsigned long long)(uintptr_t)(g.W3 + ((size_t)rb * kb + (size_t)kh * kbh) * 3 * 512);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        c.wb = ((unsigned long long)hi << 32) | lo;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) c.acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};

    unsigned long long t_start = 0, t_loop = 0, t_end = 0, r_start = 0;
    if (g.dbg) {
        t_start = __builtin_readcyclecounter();
        r_start = wall_clock64();
    }
    // ---- prologue: A0 W0 A1 W1 (A2 W2 A3 follow behind the first barrier: a wave's A0 piece queues behind whatever
    // the CU's other eleven waves issued before it, and every wave needs every A0 piece), wait for A0, read its
    // fragments.  The issue ORDER is the one the step waits count on: A0 W0 A1 W1 A2 W2 A3 | W3 A4 | ...
    x6w_issue_a<C>(c, 0);
    x6w_issue_w<C, 0>(c, 0);
    x6w_issue_a<C>(c, 1);
    x6w_issue_w<C, 1>(c, 1);
    x6w_wait<C, 1, 1, 0>(c);  // everything but A0: W0, the group A1 + W1
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    x6w_issue_a<C>(c, 2);
    x6w_issue_w<C, 2>(c, 2);
    x6w_issue_a<C>(c, 3);
    if (g.dbg) t_loop = __builtin_readcyclecounter();
    x6w_sides<C, 1, 0, false, false, true, 3 + C::LPA, C::NWORK>(c, 0, c.a_rd);  // slab 0's fragments into set 0
    int kt = 0;
    for (; kt + 4 < nk; kt += 4) {
        x6w_step<C, 0, 0, 0>(c, kt);
        x6w_step<C, 1, 1, 0>(c, kt + 1);
        x6w_step<C, 0, 2, 0>(c, kt + 2);
        x6w_step<C, 1, 3, 0>(c, kt + 3);
    }
    x6w_step<C, 0, 0, 1>(c, kt);
    x6w_step<C, 1, 1, 2>(c, kt + 1);
    x6w_step<C, 0, 2, 3>(c, kt + 2);
    x6w_step<C, 1, 3, 4>(c, kt + 3);
    if (g.dbg) {
        asm volatile("s_nop 0" ::"v"(c.acc[MT - 1][0][0]));
        t_end = __builtin_readcyclecounter();
    }
    x6_epilogue<MT, 1, KS, C::OUT3>(g, c.acc, smem_raw, m0, n0 + cp * 16, wid, lane, kh, cp * KS);
    if (g.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = g.dbg + (size_t)blockIdx.x * 8;
        d[0] = t_start;
        d[1] = t_loop;
        d[2] = t_end;
        d[3] = __builtin_readcyclecounter();
        d[4] = r_start;
        d[5] = wall_clock64();
        d[6] = __smid();
        d[7] = 0;
    }
}

template <class C>
int launch_x6w(const X6Args& g, hipStream_t stream) {
    const int tiles_m = cdiv(g.M, C::BM), tiles_n = cdiv(g.N, C::BN);
    const size_t ring = (size_t)C::D * C::STAGE;
    const size_t red = C::KS > 1 ? (size_t)C::NW * C::MT * 1024 : 0;
    const size_t lds = ring > red ? ring : red;
    static LdsAttr attr_set;
    AFTER_TRY(ensure_lds_attr(attr_set, reinterpret_cast<const void*>(gemm_x6w_kernel<C>), lds));
    int pm = 0;
    double best = 0;
    for (int cdv = 1; cdv <= 8; cdv *= 2) {
        if (tiles_m % cdv || tiles_n % (8 / cdv)) continue;
        const double cost = (double)g.M / cdv + (double)g.N / (8 / cdv);
        if (pm == 0 || cost < best) {
            pm = cdv;
            best = cost;
        }
    }
    hipLaunchKernelGGL((gemm_x6w_kernel<C>), dim3(tiles_m * tiles_n), dim3(64 * C::NW), lds, stream, g, tiles_m, tiles_n,
                       pm);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

template <class C>
int launch_x6(const X6Args& g, hipStream_t stream) {
    const int tiles_m = cdiv(g.M, C::BM), tiles_n = cdiv(g.N, C::BN);
    const size_t ring = (size_t)C::NS * C::STAGE;
    const size_t red = C::KS > 1 ? (size_t)C::NW * C::MT * C::NT * 1024 : 0;
    const size_t lds = ring > red ? ring : red;
    static LdsAttr attr_set;
    AFTER_TRY(ensure_lds_attr(attr_set, reinterpret_cast<const void*>(gemm_x6_kernel<C>), lds));
    // XCD grid pm x (8 / pm): minimise the per-XCD operand footprint A / pm + W / pn
    int pm = 0;
    double best = 0;
    for (int cdv = 1; cdv <= 8; cdv *= 2) {
        if (tiles_m % cdv || tiles_n % (8 / cdv)) continue;
        const double cost = (double)g.M / cdv + (double)g.N / (8 / cdv);
        if (pm == 0 || cost < best) {
            pm = cdv;
            best = cost;
        }
    }
    hipLaunchKernelGGL((gemm_x6_kernel<C>), dim3(tiles_m * tiles_n), dim3(64 * C::NW), lds, stream, g, tiles_m, tiles_n,
                       pm);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

// Tile table.  id = the `tile` argument of launch_gemm_x6 / after_gemm_x6.
//   id  tile     waves (k x row x col parts)  WG/CU   bytes/MFMA-cycle ratio   meant for
//    1  48 x 96   12 (2 x 1 x 6)               1       1.0                     M = 768: 256 workgroups, N % 96 == 0
//    2  48 x 32    8 (4 x 1 x 2)               1       0.6                     M = 768, narrow N, long K (MLP down)
//    3  96 x 96    4 (1 x 2 x 2)               2       1.5                     many rows: two desynchronised workgroups per CU
//    4  96 x 128   8 (1 x 2 x 4)               1       1.7                     many rows, N % 128 == 0 (MLP down: 256 WGs at M = 6144); ACC2
//    5  192 x 96   8 (1 x 4 x 2)               1       2.0                     many rows, two rounds at M = 6144
//    6  96 x 96    8 (2 x 2 x 2)               1       1.5                     M = 1536..3072
//    7  128 x 96   8 (1 x 4 x 2)               1       1.7
//    8  48 x 48    8 (2 x 1 x 3)... (small N)
template <int OUT3>
int dispatch_x6(const X6Args& g, int tile, hipStream_t stream) {
    switch (tile) {
        case 1: return launch_x6<X6Cfg<3, 6, 2, 1, 6, 2, OUT3, 1>>(g, stream);
        case 2: return launch_x6<X6Cfg<3, 2, 4, 1, 2, 2, OUT3, 1>>(g, stream);
        case 3: return launch_x6<X6Cfg<6, 6, 1, 2, 2, 2, OUT3, 2>>(g, stream);
        case 4: return launch_x6<X6Cfg<6, 8, 1, 2, 4, 2, OUT3, 1, 1>>(g, stream);
        case 5: return launch_x6p<X6Cfg<12, 6, 1, 4, 2, 2, OUT3, 1, 0, 1>, X6Cfg<12, 6, 1, 4, 2, 2, OUT3, 1>>(g, stream);
        case 15: return launch_x6<X6Cfg<12, 6, 1, 4, 2, 2, OUT3, 1>>(g, stream);  // tile 5 without the persistent walk (A/B)
        case 6: return launch_x6<X6Cfg<6, 6, 2, 2, 2, 2, OUT3, 1>>(g, stream);
        case 7: return launch_x6<X6Cfg<8, 6, 1, 4, 2, 2, OUT3, 1>>(g, stream);
        case 8: return launch_x6<X6Cfg<3, 3, 2, 1, 3, 2, OUT3, 2>>(g, stream);
        case 9: return launch_x6<X6Cfg<6, 6, 1, 2, 2, 3, OUT3, 1>>(g, stream);
        case 11: return launch_x6w<X6WCfg<3, 6, 2, OUT3>>(g, stream);  // 48 x 96, W in registers
        case 12: return launch_x6w<X6WCfg<3, 2, 4, OUT3>>(g, stream);  // 48 x 32, W in registers
        default:
            set_error("gemm_x6: no tile %d", tile);
            return AFTER_E_INVALID;
    }
}

struct X6TileInfo {
    int id, bm, bn, ks, res;
};
constexpr X6TileInfo kX6Tiles[] = {{1, 48, 96, 2, 1},  {2, 48, 32, 4, 1},  {3, 96, 96, 1, 2},  {4, 96, 128, 1, 1}, {5, 192, 96, 1, 1},
                                   {6, 96, 96, 2, 1},  {7, 128, 96, 1, 1}, {8, 48, 48, 2, 2},  {9, 96, 96, 1, 1},
                                   {11, 48, 96, 8, 1}, {12, 48, 32, 16, 1}, {15, 192, 96, 1, 1}};  // W-in-register tiles: ks = 4 x k-parts (nk % 4 == 0)

}  // namespace

namespace {
// W [N][K] (row stride ldw) x scale -> h3 blocks (two fp16 pieces per element; rows padded to 16 with zeros)
__global__ __launch_bounds__(256) void h3_split_kernel(const float* __restrict__ W, int ldw, unsigned short* __restrict__ out, int N, int K, float scale) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // four consecutive k of one row
    const int k4 = K / 4;
    const size_t rows = x6_rows_padded(N);
    if (idx >= rows * k4) return;
    const int r = (int)(idx / k4), k = 4 * (int)(idx % k4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < N) v = *reinterpret_cast<const float4*>(W + (size_t)r * ldw + k);
    h3_store4(out, r, k, K, v.x * scale, v.y * scale, v.z * scale, v.w * scale);
}
}  // namespace

int gemm_h3_split(const float* W, int ldw, unsigned short* W2, int N, int K, float scale, hipStream_t s) {
    hipLaunchKernelGGL(h3_split_kernel, dim3((unsigned)cdivll((long long)x6_rows_padded(N) * K / 4, 256)), dim3(256), 0, s, W, ldw, W2, N, K, scale);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

int gemm_x6_split(const float* W, int ldw, unsigned short* W3, int N, int K, hipStream_t s) {
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)cdivll((long long)x6_rows_padded(N) * K, 256)), dim3(256), 0, s, W,
                       ldw, W3, N, K);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

static int g_x6_force_tile = -1;  // AFTER_GEMM_X6_TILE (diagnostics): pin one tile for every launch

// Tile by shape: a cost model over the table -- workgroups per CU x slabs x (max(MFMA cycles, L2 -> LDS cycles at
// 64 B/clk) + the part of the smaller one that does not overlap) + a fixed prologue / epilogue per round of resident
// workgroups -- fitted to scripts/bench_gemm_x6.py sweeps (profiles/r3_gemm_x6_sweep.jsonl).  The W-in-register
// tiles (11, 12) overlap almost completely (deep prefetch) but exist for one workgroup per CU only.
int gemm_x6_pick_tile(int M, int N, int K) {
    if (g_x6_force_tile < 0) {
        const char* e = getenv("AFTER_GEMM_X6_TILE");
        g_x6_force_tile = e ? atoi(e) : 0;
    }
    if (g_x6_force_tile > 0) return g_x6_force_tile;
    int best = 0;
    double best_cost = 0;
    for (const X6TileInfo& t : kX6Tiles) {
        if (t.id == 9 || t.id == 15) continue;  // diagnostics only
        if (K % (32 * t.ks) != 0) continue;
        const bool wreg = t.id == 11 || t.id == 12;
        const long long wgs = (long long)cdiv(M, t.bm) * cdiv(N, t.bn);
        const long long per_cu = (wgs + 255) / 256;
        const long long rounds = (wgs + 256LL * t.res - 1) / (256LL * t.res);
        const double slabs = (double)K / 32.0;  // per workgroup, all k-parts together
        const double mfma = (double)t.bm * t.bn * 24.0 / 256.0;  // cycles per SIMD per slab
        const double load = 192.0 * (t.bm + t.bn) / 64.0;
        const double overlap = wreg ? 0.1 : 0.35;
        const double per_slab = (mfma > load ? mfma : load) + overlap * (mfma < load ? mfma : load);
        const double cost = per_cu * slabs * per_slab + rounds * (wreg ? 6500.0 : 6000.0);
        if (best == 0 || cost < best_cost) {
            best = t.id;
            best_cost = cost;
        }
    }
    return best;
}

// C (fp32) or C3 (bf16 planes, OUT3) = epi(A3 W3^T + bias); tile 0 = by shape
int launch_gemm_x6(const X6GemmArgs& a, int tile, hipStream_t stream) {
    AFTER_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.A3 && a.W3 && (a.C || a.C3), AFTER_E_INVALID, "gemm_x6: bad problem");
    AFTER_REQUIRE((a.K % 32) == 0 && ((uintptr_t)a.A3 % 16) == 0 && ((uintptr_t)a.W3 % 16) == 0, AFTER_E_INVALID,
                  "gemm_x6: K %% 32 == 0 and 16-byte aligned operands required");
    AFTER_REQUIRE(!a.C3 || ((a.N % 32) == 0 && ((uintptr_t)a.C3 % 16) == 0), AFTER_E_INVALID,
                  "gemm_x6: plane output needs N %% 32 == 0");
    AFTER_REQUIRE(a.epilogue != EPI_RESIDUAL || (a.R != nullptr && !a.C3), AFTER_E_INVALID,
                  "gemm_x6: residual epilogue needs R and an fp32 output");
    if (tile == 0) tile = gemm_x6_pick_tile(a.M, a.N, a.K);
    AFTER_REQUIRE(tile > 0, AFTER_E_INVALID, "gemm_x6: no tile for M=%d N=%d K=%d", a.M, a.N, a.K);
    for (const X6TileInfo& t : kX6Tiles)
        if (t.id == tile)
            AFTER_REQUIRE(a.K % (32 * t.ks) == 0, AFTER_E_INVALID, "gemm_x6: K not divisible by the tile's k-parts");
    X6Args g{a.A3, a.W3, a.bias, a.R, a.ldr, a.C, a.C3, a.ldc, a.M, a.N, a.K, a.epilogue, a.dbg};
    return a.C3 ? dispatch_x6<1>(g, tile, stream) : dispatch_x6<0>(g, tile, stream);
}

}  // namespace after

// diagnostics / tests: the three bf16 planes of an fp32 matrix as x6 blocks (pad16(N) x 3 x K unsigned short)
extern "C" int after_gemm_x6_split(const float* W, int ldw, unsigned short* W3, int N, int K, void* stream) {
    AFTER_REQUIRE(W && W3 && N > 0 && K > 0, AFTER_E_INVALID, "gemm_x6_split: bad argument");
    return after::gemm_x6_split(W, ldw, W3, N, K, (hipStream_t)stream);
}

static unsigned long long* g_x6_dbg = nullptr;
extern "C" void after_gemm_x6_set_debug(unsigned long long* dbg) { g_x6_dbg = dbg; }

extern "C" int after_gemm_x6_pick_tile(int M, int N, int K) { return after::gemm_x6_pick_tile(M, N, K); }
extern "C" long long after_gemm_x6_offset(int row, int plane, int col, int K) {
    return (long long)after::x6_offset(row, plane, col, K);
}

extern "C" int after_gemm_x6(const unsigned short* A3, const unsigned short* W3, const float* bias, const float* R,
                             int ldr, float* C, unsigned short* C3, int ldc, int M, int N, int K, int epilogue, int tile,
                             void* stream) {
    AFTER_REQUIRE((C != nullptr) != (C3 != nullptr), AFTER_E_INVALID, "gemm_x6: exactly one of C (fp32) / C3 (planes)");
    after::X6GemmArgs a{A3, W3, bias, R, ldr, C, C3, ldc, M, N, K, epilogue, g_x6_dbg};
    return after::launch_gemm_x6(a, tile, (hipStream_t)stream);
}
