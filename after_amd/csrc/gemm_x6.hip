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

namespace after {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float gelu_erf_x6(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

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

// chunk swizzle of the 64-byte plane rows: q = (row / 4) % 4 -> {0, 2, 3, 1}
__device__ __forceinline__ int swz4(int q) { return (0x78 >> (2 * q)) & 3; }

template <int MB_, int NBK_, int KS_, int RS_, int CP_, int NS_, int OUT3_, int RES_, int ACC2_ = 0>
struct X6Cfg {
    static constexpr int MB = MB_, NBK = NBK_, KS = KS_, RS = RS_, CP = CP_, NS = NS_, OUT3 = OUT3_, RES = RES_;
    // ACC2: even / odd slabs accumulate into separate registers (two chains of half the length, summed once at
    // the end: the rounding-error growth of a k-part twice as short; for the long-K tiles without a K split)
    static constexpr int ACC2 = ACC2_;
    static constexpr int BM = 16 * MB, BN = 16 * NBK;
    static constexpr int MT = MB / RS, NT = NBK / CP;   // 16x16 blocks per wave
    static constexpr int NW = KS * RS * CP;             // waves
    static constexpr int GA = 3 * MB, GW = 3 * NBK;     // 1-KB pieces (16 rows of one plane) per k-part: A, W
    static constexpr int PPK = GA + GW;
    static constexpr int PART = PPK * 1024;             // bytes of one k-part of a stage
    static constexpr int STAGE = KS * PART;
    static constexpr int P = KS * PPK;                  // pieces per stage
    static constexpr int LPS = (P + NW - 1) / NW;       // pieces per wave per slab (the last may be missing)
    static constexpr bool RAGGED = (P % NW) != 0;
    static constexpr int NMMA = 6 * MT * NT;
    static constexpr int NREAD = 3 * (MT + NT);
    static constexpr int NWORK = LPS + NREAD;
    static constexpr int WPS = (NW * RES + 3) / 4;      // waves per SIMD the register budget must allow
    static_assert(MB % RS == 0 && NBK % CP == 0, "tile shape");
    static_assert(NS * STAGE <= 160 * 1024, "ring exceeds the LDS");
    static_assert(NW <= 16, "too many waves");
    static_assert(PART <= 65536, "fragment offsets are 16-bit immediates");
};

template <class C>
struct X6State {
    f32x4 acc[C::ACC2 + 1][C::MT][C::NT];
    u32x4 fa[2][3][C::MT], fw[2][3][C::NT];  // fragments of two consecutive slabs: [set][plane][block]
    unsigned voff;                           // per-lane byte offset inside a DMA piece (lane x 16)
    unsigned long long sb[C::LPS];           // wave-uniform source address of each piece's x6 block, slab 0 of its k-part
    unsigned a_rd, w_rd;                     // per-lane LDS byte address of this wave's A / W fragments, stage 0
    unsigned lds0;
    int wid;
    bool full;                               // this wave moves LPS pieces per slab (else LPS - 1)
};

// products in the order of increasing magnitude: (W plane, A plane), 0 = h, 1 = m, 2 = l
constexpr int kWP[6] = {2, 0, 1, 1, 0, 0};
constexpr int kAP[6] = {0, 2, 1, 0, 1, 0};

template <class C>
__device__ __forceinline__ void x6_dma(const X6State<C>& c, int i, int slab, int stage) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :
                 : "s"(c.lds0 + (unsigned)(stage * C::STAGE + (c.wid + C::NW * i) * 1024)), "v"(c.voff),
                   "s"(c.sb[i] + (unsigned long long)((unsigned)slab * 3072u))
                 : "memory");  // m0: reserved register, see gemm.hip (AFTER_BAL_DMA) and after_amd/build.py
}

template <class C>
__device__ __forceinline__ void x6_issue_slab(const X6State<C>& c, int slab, int stage) {
#pragma unroll
    for (int i = 0; i < C::LPS; ++i)
        if (!C::RAGGED || i + 1 < C::LPS || c.full) x6_dma<C>(c, i, slab, stage);
}

// side-work item W of a slab step: W < LPS -> DMA piece W of slab kt + NS into the ring slot just retired;
// else fragment read W - LPS of slab kt + 1 (A blocks then W blocks, plane-major) into set NXT
// STEADY: both unconditional (the steady-state MFMA stream has no branches); else the runtime flags decide
template <class C, int NXT, int W, bool STEADY>
__device__ __forceinline__ void x6_side(X6State<C>& c, bool refill, bool more, int slab_new, int stage_new,
                                        unsigned a_next, unsigned w_next) {
    if constexpr (W < C::LPS) {
        if ((STEADY || refill) && (!C::RAGGED || W + 1 < C::LPS || c.full)) x6_dma<C>(c, W, slab_new, stage_new);
    } else {
        constexpr int R = W - C::LPS;
        if (STEADY || more) {
            if constexpr (R < 3 * C::MT) {
                constexpr int pl = R / C::MT, i = R % C::MT;
                asm volatile("ds_read_b128 %0, %1 offset:%2"
                             : "=v"(c.fa[NXT][pl][i])
                             : "v"(a_next), "i"((pl * C::BM + i * 16) * 64));
            } else {
                constexpr int R2 = R - 3 * C::MT;
                constexpr int pl = R2 / C::NT, j = R2 % C::NT;
                asm volatile("ds_read_b128 %0, %1 offset:%2"
                             : "=v"(c.fw[NXT][pl][j])
                             : "v"(w_next), "i"((pl * C::BN + j * 16) * 64));
            }
        }
    }
}

template <class C, int NXT, int W, int WEND, bool STEADY>
__device__ __forceinline__ void x6_sides(X6State<C>& c, bool refill, bool more, int slab_new, int stage_new,
                                         unsigned a_next, unsigned w_next) {
    if constexpr (W < WEND) {
        x6_side<C, NXT, W, STEADY>(c, refill, more, slab_new, stage_new, a_next, w_next);
        x6_sides<C, NXT, W + 1, WEND, STEADY>(c, refill, more, slab_new, stage_new, a_next, w_next);
    }
}

// MFMA S of the slab in set CUR, followed by its share of the side work
template <class C, int CUR, int S, bool STEADY>
__device__ __forceinline__ void x6_mma(X6State<C>& c, bool refill, bool more, int slab_new, int stage_new,
                                       unsigned a_next, unsigned w_next) {
    if constexpr (S < C::NMMA) {
        constexpr int p = S / (C::MT * C::NT), i = (S / C::NT) % C::MT, j = S % C::NT;
        // W fragment as srcA: the accumulator holds C^T (four consecutive columns of one row per lane)
        constexpr int AS = C::ACC2 ? CUR : 0;
        c.acc[AS][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, c.fw[CUR][kWP[p]][j]),
                                                                  __builtin_bit_cast(bf16x8, c.fa[CUR][kAP[p]][i]),
                                                                  c.acc[AS][i][j], 0, 0, 0);
        constexpr int w0 = (S * C::NWORK) / C::NMMA, w1 = ((S + 1) * C::NWORK) / C::NMMA;
        x6_sides<C, CUR ^ 1, w0, w1, STEADY>(c, refill, more, slab_new, stage_new, a_next, w_next);
        __builtin_amdgcn_sched_barrier(0);
        x6_mma<C, CUR, S + 1, STEADY>(c, refill, more, slab_new, stage_new, a_next, w_next);
    }
}

template <class C, int SET, int I>
__device__ __forceinline__ void x6_fence_regs(X6State<C>& c) {
    if constexpr (I < 3 * C::MT) {
        asm volatile("" : "+v"(c.fa[SET][I / C::MT][I % C::MT]));
        x6_fence_regs<C, SET, I + 1>(c);
    } else if constexpr (I < 3 * (C::MT + C::NT)) {
        constexpr int R = I - 3 * C::MT;
        asm volatile("" : "+v"(c.fw[SET][R / C::NT][R % C::NT]));
        x6_fence_regs<C, SET, I + 1>(c);
    }
}

// wait until at most `slabs` (<= NS - 1) of this wave's later slabs are still in flight
template <class C>
__device__ __forceinline__ void x6_wait(const X6State<C>& c, int slabs) {
    constexpr int L = C::LPS, L1 = C::LPS > 1 ? C::LPS - 1 : 0;
    static_assert(3 * L < 64, "vmcnt is a 6-bit counter");
    if (!C::RAGGED || c.full) {
        if (slabs >= 3 && C::NS >= 4) wait_vmcnt_imm<3 * L>();
        else if (slabs == 2 && C::NS >= 3) wait_vmcnt_imm<2 * L>();
        else if (slabs == 1) wait_vmcnt_imm<L>();
        else wait_vmcnt_imm<0>();
    } else {
        if (slabs >= 3 && C::NS >= 4) wait_vmcnt_imm<3 * L1>();
        else if (slabs == 2 && C::NS >= 3) wait_vmcnt_imm<2 * L1>();
        else if (slabs == 1) wait_vmcnt_imm<L1>();
        else wait_vmcnt_imm<0>();
    }
}

// one slab: retire slab kt's fragment reads, publish slab kt + 1 (one barrier), then slab kt's MFMAs with the
// refill of the freed ring slot (slab kt + NS) and slab kt + 1's fragment reads dealt out behind them.
// STEADY (kt + NS < nk): no conditions in the MFMA stream; the last NS slabs take the runtime flags.
template <class C, int CUR, bool STEADY>
__device__ __forceinline__ void x6_step(X6State<C>& c, int kt, int nk) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    x6_fence_regs<C, CUR, 0>(c);
    __builtin_amdgcn_sched_barrier(0);
    const bool more = STEADY || kt + 1 < nk, refill = STEADY || kt + C::NS < nk;
    if (more) {
        // in flight after slab kt + 1: slabs kt + 2 .. min(kt + NS - 1, nk - 1)
        if constexpr (STEADY) x6_wait<C>(c, C::NS - 2);
        else x6_wait<C>(c, (kt + C::NS - 1 < nk - 1 ? kt + C::NS - 1 : nk - 1) - (kt + 1));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    const int sn = (kt + 1) % C::NS;
    const unsigned a_next = c.a_rd + (unsigned)(sn * C::STAGE), w_next = c.w_rd + (unsigned)(sn * C::STAGE);
    x6_mma<C, CUR, 0, STEADY>(c, refill, more, kt + C::NS, kt % C::NS, a_next, w_next);
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

    // ---- split-K reduction through LDS in k-part order (bit-deterministic), as in gemm.hip
    float* red = reinterpret_cast<float*>(smem_raw);
    if constexpr (KS > 1) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                *reinterpret_cast<f32x4*>(red + ((wid * MT * NT + i * NT + j) * 64 + lane) * 4) = c.acc[0][i][j];
        __syncthreads();
    }
    // accumulator layout (transposed MFMA): lane l holds C[row = l & 15][col = 4 (l >> 4) + r]
    const int crow = lane & 15, ccol0 = 4 * (lane >> 4);
    const bool vec_ok = C::OUT3 || (((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0) &&
                                    (g.epilogue != EPI_RESIDUAL ||
                                     (((g.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.R) & 15) == 0))));
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int gn = n0 + cp * (BN / C::CP) + j * 16 + ccol0;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (gn + r < N) bv[r] = g.bias[gn + r];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if ((i * NT + j) % KS != kh) continue;
            f32x4 o = c.acc[0][i][j];
            if constexpr (KS > 1) {
                const int w0 = (cp * RS + rp) * KS;  // first wave of this (row part, column part)
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
            if constexpr (C::OUT3) {
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
int launch_x6(const X6Args& g, hipStream_t stream) {
    const int tiles_m = cdiv(g.M, C::BM), tiles_n = cdiv(g.N, C::BN);
    const size_t ring = (size_t)C::NS * C::STAGE;
    const size_t red = C::KS > 1 ? (size_t)C::NW * C::MT * C::NT * 1024 : 0;
    const size_t lds = ring > red ? ring : red;
    static bool attr_set = false;
    if (!attr_set) {
        AFTER_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6_kernel<C>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
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
        case 5: return launch_x6<X6Cfg<12, 6, 1, 4, 2, 2, OUT3, 1>>(g, stream);
        case 6: return launch_x6<X6Cfg<6, 6, 2, 2, 2, 2, OUT3, 1>>(g, stream);
        case 7: return launch_x6<X6Cfg<8, 6, 1, 4, 2, 2, OUT3, 1>>(g, stream);
        case 8: return launch_x6<X6Cfg<3, 3, 2, 1, 3, 2, OUT3, 2>>(g, stream);
        case 9: return launch_x6<X6Cfg<6, 6, 1, 2, 2, 3, OUT3, 1>>(g, stream);
        default:
            set_error("gemm_x6: no tile %d", tile);
            return AFTER_E_INVALID;
    }
}

struct X6TileInfo {
    int id, bm, bn, ks, res;
};
constexpr X6TileInfo kX6Tiles[] = {{1, 48, 96, 2, 1},  {2, 48, 32, 4, 1},  {3, 96, 96, 1, 2},  {4, 96, 128, 1, 1}, {5, 192, 96, 1, 1},
                                   {6, 96, 96, 2, 1},  {7, 128, 96, 1, 1}, {8, 48, 48, 2, 2},  {9, 96, 96, 1, 1}};

}  // namespace

int gemm_x6_split(const float* W, int ldw, unsigned short* W3, int N, int K, hipStream_t s) {
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)cdivll((long long)x6_rows_padded(N) * K, 256)), dim3(256), 0, s, W,
                       ldw, W3, N, K);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

static int g_x6_force_tile = -1;  // AFTER_GEMM_X6_TILE (diagnostics): pin one tile for every launch

// Tile by shape: a cost model over the table -- rounds of resident workgroups x (slabs x max(MFMA cycles,
// L2->LDS cycles at 64 B/clk) + a fixed prologue / epilogue) -- fitted to scripts/bench_gemm_x6.py sweeps.
int gemm_x6_pick_tile(int M, int N, int K) {
    if (g_x6_force_tile < 0) {
        const char* e = getenv("AFTER_GEMM_X6_TILE");
        g_x6_force_tile = e ? atoi(e) : 0;
    }
    if (g_x6_force_tile > 0) return g_x6_force_tile;
    int best = 0;
    double best_cost = 0;
    for (const X6TileInfo& t : kX6Tiles) {
        if (t.id == 9) continue;  // diagnostics only
        if (K % (32 * t.ks) != 0) continue;
        const long long wgs = (long long)cdiv(M, t.bm) * cdiv(N, t.bn);
        const long long rounds = (wgs + 256LL * t.res - 1) / (256LL * t.res);
        const double slabs = (double)K / 32.0;  // per workgroup, all k-parts together
        const double mfma = t.res * (double)t.bm * t.bn * 24.0 / 256.0;  // cycles per SIMD per slab round
        const double load = t.res * 192.0 * (t.bm + t.bn) / 64.0;
        const double per_slab = (mfma > load ? mfma : load) + 0.35 * (mfma < load ? mfma : load);
        const double cost = rounds * (slabs * per_slab + 6000.0);
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
