// fp32 GEMMs on the f16 matrix pipe with TWO-piece operands ("h3": three exact MFMAs per fp32 product block).
//
// Every operand element x is carried as  x * 2^s = h + l + e,  h = RN_f16(x 2^s), l = RN_f16(x 2^s - h): two fp16 numbers, a
// 22-bit significand (|e| <= 2^-22 |x 2^s|; elements below 2^-3 in scaled units keep an ABSOLUTE error <= 2^-25, the fp16
// subnormal quantum).  s is a per-tensor power of two chosen from a guaranteed bound of the tensor (weights: their max; LayerNorm
// outputs: sqrt(E) max|w| + max|b|; the MLP hidden layer: the largest row L1 norm of the up-projection x that bound + max|bias|),
// so nothing overflows fp16's range and the scaling itself is exact.  A product block is
//     acc += Wh Al + Wh Ah + Wl Ah        (v_mfma_f32_16x16x32_f16, fp32 accumulate; every piece product is exact in fp32)
// i.e. the fp32 product of the operands up to the dropped Wl Al term (<= 2^-22 relative) and the pieces' representation error
// (<= 2^-22 relative each): per-term errors 4 - 8x BELOW what the fp32 accumulation of a K = 512 .. 1536 dot product commits in the
// reference's own arithmetic, and uncorrelated.  Measured against fp64 the result is as close as the fp32 MFMA chain's
// (tests/test_gemm_gpu.py::test_h3_*).  Half the MFMAs and two thirds of the operand bytes of the
// three-plane bf16 form of gemm_x6_pipe.h -- whose K loops ran at the matrix pipes' issue limit AND at the clock the chip sustains
// under them (profiles/r6_gstag.txt): fewer instructions per flop is the only way down.
//
// Storage ("h3 blocks"): the x6 block layout of common.h with two planes -- 1-KB blocks [R / 16][K / 32][plane h, l] of 16 rows x
// 32 k of fp16, row r at byte 64 r, its four 16-byte chunks XOR-permuted; a block is one LDS-DMA piece and its own LDS image.
//
// Two rings, both fed by loader waves (see gemm_x6_pipe.h for the reasoning: x6l_* / x6r_*):
//   h3l_*  fragments of two consecutive slabs in registers (small tiles: MLP-down, 96 x 128)
//   h3r_*  ROLLING fragments for the 192 x 192 tiles: W l and A h double-buffered by slab parity, W h and A l re-read for the next
//          slab as soon as their last product of the current slab has been issued -- with the products in the order
//          (Wh, Al) (Wh, Ah) (Wl, Ah) no read is left exposed at the end of a slab (the three-plane form left six)
#pragma once
#include <cstdint>

#include "gemm_x6_pipe.h"

namespace after {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__host__ __device__ inline size_t h3_elems(int rows, int K) { return x6_rows_padded(rows) * 2 * (size_t)K; }
// element (unsigned short) offset of (row r, plane p, column k) -- k % 8 consecutive elements stay contiguous
__host__ __device__ inline size_t h3_offset(int r, int p, int k, int K) {
    const int rr = r & 15, c = (k & 31) >> 3;
    const int slot = c ^ ((0x78 >> (2 * ((rr >> 2) & 3))) & 3);
    return (((size_t)(r >> 4) * (size_t)(K >> 5) + (size_t)(k >> 5)) * 2 + (size_t)p) * 512 + (size_t)(rr * 32 + slot * 8 + (k & 7));
}
// four consecutive (already scaled) floats -> their two fp16 pieces
__device__ __forceinline__ void h3_split4(float x0, float x1, float x2, float x3, uint2& h, uint2& l) {
    const f32x2 v0 = {x0, x1}, v1 = {x2, x3};
    const f16x2 h0 = __builtin_convertvector(v0, f16x2), h1 = __builtin_convertvector(v1, f16x2);
    const f32x2 r0 = v0 - __builtin_convertvector(h0, f32x2), r1 = v1 - __builtin_convertvector(h1, f32x2);
    h.x = __builtin_bit_cast(unsigned, h0), h.y = __builtin_bit_cast(unsigned, h1);
    l.x = __builtin_bit_cast(unsigned, __builtin_convertvector(r0, f16x2));
    l.y = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, f16x2));
}
__device__ __forceinline__ void h3_store4(unsigned short* base, int r, int k, int K, float x0, float x1, float x2, float x3) {
    uint2 h, l;
    h3_split4(x0, x1, x2, x3, h, l);
    unsigned short* p = base + h3_offset(r, 0, k, K);
    *reinterpret_cast<uint2*>(p) = h;
    *reinterpret_cast<uint2*>(p + 512) = l;
}
// the largest power of two s with bound x s <= 2^15 (half of fp16's range: room for the rounding of the bound itself), within
// [2^-12, 2^14]
__host__ inline float h3_scale_for(float bound) {
    float s = 16384.0f;
    while (s > 1.0f / 4096.0f && !(bound * s <= 32768.0f)) s *= 0.5f;
    return s;
}

// ===================================================================================================== loader ring, two fragment sets
template <int MB_, int NBK_, int RS_, int CP_, int NS_, int ACC2_ = 0, int SC1_ = 0>
struct H3LCfg {
    static constexpr int SPLIT = 1, NPL = 2, TIER = 0;
    static constexpr int MB = MB_, NBK = NBK_, RS = RS_, CP = CP_, NS = NS_, ACC2 = ACC2_, SC1 = SC1_, KS = 1, CONV = 0;
    static constexpr int BM = 16 * MB, BN = 16 * NBK, MT = MB / RS, NT = NBK / CP, NW = RS * CP;
    static constexpr int GA = 2 * MB, GW = 2 * NBK, PPK = GA + GW, PPKI = PPK, STAGE = PPK * 1024;
    static constexpr int NMMA = 3 * MT * NT, NREAD = 2 * (MT + NT);
    static_assert(MB % RS == 0 && NBK % CP == 0 && NW == 8, "tile shape (eight waves)");
    static_assert(NS * STAGE <= 160 * 1024, "ring exceeds the LDS");
    static_assert(STAGE <= 65536, "fragment offsets are 16-bit immediates");
};

template <class C>
struct H3LState {
    f32x4 acc[C::ACC2 + 1][C::MT][C::NT];
    u32x4 fa[2][2][C::MT], fw[2][2][C::NT];  // fragments of two consecutive slabs: [set][plane h, l][block]
    unsigned voff;
    unsigned long long a_src, w_src;  // loaders: slab 0 of the tile's first A / W row group, plane h (wave-uniform)
    unsigned rgs;                     // bytes between consecutive 16-row groups of an operand: (K / 32) x 2048
    unsigned a_rd, w_rd, lds0;
};

// one LDS-DMA piece: issue-order item I of a slab (the GW weight pieces first, then the GA activation pieces)
template <class C, class S, int I>
__device__ __forceinline__ void h3_dma(const S& c, int slab, int stage) {
    constexpr bool isW = I < C::GW;
    constexpr int q = isW ? I : I - C::GW;
    constexpr int plane = isW ? q / C::NBK : q / C::MB, grp = isW ? q % C::NBK : q % C::MB;
    constexpr int piece = isW ? C::GA + q : q;
    const unsigned long long src = (isW ? c.w_src : c.a_src) + (unsigned long long)((unsigned)grp * c.rgs + (unsigned)plane * 1024u + (unsigned)slab * 2048u);
    const unsigned dst = c.lds0 + (unsigned)(stage * C::STAGE + piece * 1024);
    lds_dma16<C::SC1 ? 16 : 0>(dst, c.voff, src);
}
template <class C, int NL, int LID>
struct H3Role {
    static constexpr int ND = LID < 0 ? 0 : (C::PPK - LID + NL - 1) / NL;  // DMA items of this wave per slab
};
template <class C, class S, int NL, int LID, int W = 0>
__device__ __forceinline__ void h3_issue_mine(const S& c, int slab, int stage) {
    if constexpr (LID >= 0 && LID + NL * W < C::PPK) {
        h3_dma<C, S, LID + NL * W>(c, slab, stage);
        h3_issue_mine<C, S, NL, LID, W + 1>(c, slab, stage);
    }
}

// side-work item W of a slab step: W < ND -> this loader's DMA item W of slab kt + NS; then the fragment reads of slab kt + 1
// (A blocks then W blocks, plane-major) into set NXT
template <class C, int NL, int LID, int NXT, int W, bool STEADY>
__device__ __forceinline__ void h3l_side(H3LState<C>& c, bool refill, bool more, int slab_new, int stage_new, unsigned a_next, unsigned w_next) {
    constexpr int ND = H3Role<C, NL, LID>::ND;
    if constexpr (W < ND) {
        if (STEADY || refill) h3_dma<C, H3LState<C>, LID + NL * W>(c, slab_new, stage_new);
    } else {
        constexpr int R = W - ND;
        if (STEADY || more) {
            if constexpr (R < 2 * C::MT) {
                constexpr int pl = R / C::MT, i = R % C::MT;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.fa[NXT][pl][i]) : "v"(a_next), "i"((pl * C::BM + i * 16) * 64));
            } else {
                constexpr int R2 = R - 2 * C::MT;
                constexpr int pl = R2 / C::NT, j = R2 % C::NT;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.fw[NXT][pl][j]) : "v"(w_next), "i"((pl * C::BN + j * 16) * 64));
            }
        }
    }
}
template <class C, int NL, int LID, int NXT, int W, int WEND, bool STEADY>
__device__ __forceinline__ void h3l_sides(H3LState<C>& c, bool refill, bool more, int slab_new, int stage_new, unsigned a_next, unsigned w_next) {
    if constexpr (W < WEND) {
        h3l_side<C, NL, LID, NXT, W, STEADY>(c, refill, more, slab_new, stage_new, a_next, w_next);
        h3l_sides<C, NL, LID, NXT, W + 1, WEND, STEADY>(c, refill, more, slab_new, stage_new, a_next, w_next);
    }
}
// products in the order of increasing magnitude: (W plane, A plane), 0 = h, 1 = l
constexpr int kH3WP[3] = {0, 1, 0};
constexpr int kH3AP[3] = {1, 0, 0};
template <class C, int NL, int LID, int CUR, int S, bool STEADY>
__device__ __forceinline__ void h3l_mma(H3LState<C>& c, bool refill, bool more, int slab_new, int stage_new, unsigned a_next, unsigned w_next) {
    if constexpr (S < C::NMMA) {
        constexpr int p = S / (C::MT * C::NT), i = (S / C::NT) % C::MT, j = S % C::NT;
        constexpr int AS = C::ACC2 ? CUR : 0;
        // W fragment as srcA: the accumulator holds C^T (four consecutive columns of one row per lane)
        c.acc[AS][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, c.fw[CUR][kH3WP[p]][j]),
                                                                 __builtin_bit_cast(f16x8, c.fa[CUR][kH3AP[p]][i]), c.acc[AS][i][j], 0, 0, 0);
        constexpr int NWK = H3Role<C, NL, LID>::ND + C::NREAD;
        constexpr int w0 = (S * NWK) / C::NMMA, w1 = ((S + 1) * NWK) / C::NMMA;
        h3l_sides<C, NL, LID, CUR ^ 1, w0, w1, STEADY>(c, refill, more, slab_new, stage_new, a_next, w_next);
        __builtin_amdgcn_sched_barrier(0);
        h3l_mma<C, NL, LID, CUR, S + 1, STEADY>(c, refill, more, slab_new, stage_new, a_next, w_next);
    }
}
template <class C, int SET>
__device__ __forceinline__ void h3l_fence(H3LState<C>& c) {
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int i = 0; i < C::MT; ++i) asm volatile("" : "+v"(c.fa[SET][pl][i]));
#pragma unroll
        for (int j = 0; j < C::NT; ++j) asm volatile("" : "+v"(c.fw[SET][pl][j]));
    }
}
// one slab: retire slab kt's fragment reads, publish slab kt + 1 (a loader waits for its pieces -- all but the EXTRA youngest
// vector-memory instructions of the wave, for callers with stores in flight -- then one barrier), then slab kt's MFMAs with the
// side work dealt out behind them
template <class C, int NL, int LID, int CUR, bool STEADY, int EXTRA = 0>
__device__ __forceinline__ void h3l_step(H3LState<C>& c, int kt, int nk) {
    constexpr int ND = H3Role<C, NL, LID>::ND;
    static_assert((C::NS - 2) * ND + EXTRA < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    h3l_fence<C, CUR>(c);
    __builtin_amdgcn_sched_barrier(0);
    const bool more = STEADY || kt + 1 < nk, refill = STEADY || kt + C::NS < nk;
    if (more) {
        if constexpr (LID >= 0) {
            // in flight behind slab kt + 1: slabs kt + 2 .. min(kt + NS - 1, nk - 1), ND pieces of this wave each
            if (C::NS >= 4 && (STEADY || kt + 3 < nk)) wait_vmcnt_imm<(C::NS >= 4 ? 2 * ND : 0) + EXTRA>();
            else if (C::NS >= 3 && (STEADY || kt + 2 < nk)) wait_vmcnt_imm<(C::NS >= 3 ? ND : 0) + EXTRA>();
            else wait_vmcnt_imm<EXTRA>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    const int sn = (kt + 1) % C::NS;
    const unsigned a_next = c.a_rd + (unsigned)(sn * C::STAGE), w_next = c.w_rd + (unsigned)(sn * C::STAGE);
    h3l_mma<C, NL, LID, CUR, 0, STEADY>(c, refill, more, kt + C::NS, kt % C::NS, a_next, w_next);
}
template <class C, int NL, int LID>
__device__ __forceinline__ void h3l_fill(const H3LState<C>& c) {
    static_assert(C::NS >= 2 && C::NS <= 4, "ring depth");
    h3_issue_mine<C, H3LState<C>, NL, LID>(c, 0, 0);
    h3_issue_mine<C, H3LState<C>, NL, LID>(c, 1, 1);
    if constexpr (C::NS >= 3) h3_issue_mine<C, H3LState<C>, NL, LID>(c, 2, 2);
    if constexpr (C::NS >= 4) h3_issue_mine<C, H3LState<C>, NL, LID>(c, 3, 3);
}
// the K loop of a tile whose ring fill has been issued (nk >= NS + 2, even).  EXTRA: see x6l_main
template <class C, int NL, int LID, int EXTRA = 0>
__device__ __forceinline__ void h3l_main(H3LState<C>& c, int nk) {
    constexpr int ND = H3Role<C, NL, LID>::ND;
    static_assert((C::NS - 1) * ND + EXTRA < 64, "vmcnt is a 6-bit counter");
    if constexpr (LID >= 0) wait_vmcnt_imm<(C::NS - 1) * ND + EXTRA>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    h3l_sides<C, NL, LID, 0, ND, ND + C::NREAD, true>(c, false, true, 0, 0, c.a_rd, c.w_rd);
    int kt = 0;
    if constexpr (EXTRA > 0) {
        h3l_step<C, NL, LID, 0, true, EXTRA>(c, 0, nk);
        h3l_step<C, NL, LID, 1, true, (C::NS >= 3 ? EXTRA : 0)>(c, 1, nk);
        kt = 2;
    }
    for (; kt + 1 + C::NS < nk; kt += 2) {
        h3l_step<C, NL, LID, 0, true>(c, kt, nk);
        h3l_step<C, NL, LID, 1, true>(c, kt + 1, nk);
    }
    for (; kt < nk; kt += 2) {
        h3l_step<C, NL, LID, 0, false>(c, kt, nk);
        if (kt + 1 < nk) h3l_step<C, NL, LID, 1, false>(c, kt + 1, nk);
    }
}

// ===================================================================================================== rolling fragments
template <int MB_, int NBK_, int RS_, int CP_, int NS_, int SC1_ = 0>
struct H3RCfg {
    static constexpr int SPLIT = 1, NPL = 2, TIER = 0;
    static constexpr int MB = MB_, NBK = NBK_, RS = RS_, CP = CP_, NS = NS_, SC1 = SC1_, KS = 1, CONV = 0, ACC2 = 0;
    static constexpr int BM = 16 * MB, BN = 16 * NBK, MT = MB / RS, NT = NBK / CP, NW = RS * CP;
    static constexpr int GA = 2 * MB, GW = 2 * NBK, PPK = GA + GW, PPKI = PPK, STAGE = PPK * 1024;
    static constexpr int PN = MT * NT, NMMA = 3 * PN, NREAD = 2 * (MT + NT);
    static_assert(MB % RS == 0 && NBK % CP == 0 && NW == 8, "tile shape (eight waves)");
    static_assert(NS * STAGE <= 160 * 1024, "ring exceeds the LDS");
    static_assert(STAGE <= 65536, "fragment offsets are 16-bit immediates");
    static_assert(NS == 2 || NS == 3, "ring depth");
};
template <class C>
struct H3RState {
    f32x4 acc[C::MT][C::NT];
    u32x4 wh[C::NT], wl[2][C::NT];  // W fragments: h rolling, l by slab parity
    u32x4 al[C::MT], ah[2][C::MT];  // A fragments: l rolling, h by slab parity
    unsigned voff;
    unsigned long long a_src, w_src;
    unsigned rgs;
    unsigned a_rd, w_rd, lds0;
    unsigned prof[4], tprev;  // (X6R_PROF)
};
// fragment read R of the next slab, in issue order: [0, MT) A h (other set) | [MT, MT + NT) W l (other set) | [.., + MT) A l |
// [.., + NT) W h
template <class C, int NXT, int R>
__device__ __forceinline__ void h3r_read(H3RState<C>& c, unsigned a_base, unsigned w_base) {
    constexpr int MT = C::MT, NT = C::NT;
    if constexpr (R < MT) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.ah[NXT][R]) : "v"(a_base), "i"((0 * C::BM + R * 16) * 64));
    } else if constexpr (R < MT + NT) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.wl[NXT][R - MT]) : "v"(w_base), "i"((1 * C::BN + (R - MT) * 16) * 64));
    } else if constexpr (R < 2 * MT + NT) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.al[R - MT - NT]) : "v"(a_base), "i"((1 * C::BM + (R - MT - NT) * 16) * 64));
    } else {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.wh[R - 2 * MT - NT]) : "v"(w_base), "i"((0 * C::BN + (R - 2 * MT - NT) * 16) * 64));
    }
}
template <class C, int NXT, int R, int REND>
__device__ __forceinline__ void h3r_reads(H3RState<C>& c, unsigned a_base, unsigned w_base) {
    if constexpr (R < REND) {
        h3r_read<C, NXT, R>(c, a_base, w_base);
        h3r_reads<C, NXT, R + 1, REND>(c, a_base, w_base);
    }
}
template <class C>
struct H3RSched {
    static constexpr int MT = C::MT, NT = C::NT, PN = C::PN;
    static constexpr int ready(int R) {  // first MFMA index behind which read R may be issued
        return R < MT + NT ? 0                 // the double-buffered sets: any time
               : R < 2 * MT + NT ? PN - 1      // A l: after the first product (Wh, Al)
                                 : 2 * PN - 1; // W h: after the second (Wh, Ah)
    }
    static constexpr int slot(int R) {  // reads go out in order, one per MFMA from `ready` on
        int s = -1;
        for (int r = 0; r <= R; ++r) {
            const int rd = ready(r);
            s = rd > s + 1 ? rd : s + 1;
            if (s > C::NMMA - 1) s = C::NMMA - 1;
        }
        return s;
    }
};
template <class C, int NL, int LID, int CUR, int S, int R, int D>
__device__ __forceinline__ void h3r_mma(H3RState<C>& c, bool refill, bool more, int slab_new, int stage_new, unsigned a_next, unsigned w_next) {
    if constexpr (S < C::NMMA) {
        constexpr int PN = C::PN, p = S / PN, i = (S % PN) / C::NT, j = S % C::NT;
        const u32x4& wf = p == 2 ? c.wl[CUR][j] : c.wh[j];
        const u32x4& af = p == 0 ? c.al[i] : c.ah[CUR][i];
        c.acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, af), c.acc[i][j], 0, 0, 0);
        constexpr int ND = H3Role<C, NL, LID>::ND;
        constexpr int NDD = ND > 0 ? ND : 1;
        constexpr int DSP = (C::NMMA * 2 / 3) / NDD > 0 ? (C::NMMA * 2 / 3) / NDD : 1;
        constexpr bool dma_here = D < ND && S == D * DSP;
        if constexpr (dma_here) {
            if (refill) h3_dma<C, H3RState<C>, LID + NL * D>(c, slab_new, stage_new);
        }
        constexpr int NR = C::NREAD;
        constexpr int r1 = [] {
            int r = R;
            while (r < NR && H3RSched<C>::slot(r) <= S) ++r;
            return r;
        }();
        if constexpr (r1 > R) {
            if (more) h3r_reads<C, CUR ^ 1, R, r1>(c, a_next, w_next);
        }
        __builtin_amdgcn_sched_barrier(0);
        h3r_mma<C, NL, LID, CUR, S + 1, r1, (dma_here ? D + 1 : D)>(c, refill, more, slab_new, stage_new, a_next, w_next);
    }
}
template <class C, int SET>
__device__ __forceinline__ void h3r_fence(H3RState<C>& c) {
#pragma unroll
    for (int i = 0; i < C::MT; ++i) asm volatile("" : "+v"(c.ah[SET][i]), "+v"(c.al[i]));
#pragma unroll
    for (int j = 0; j < C::NT; ++j) asm volatile("" : "+v"(c.wl[SET][j]), "+v"(c.wh[j]));
}
// one slab (kt: CUR = kt & 1): its fragments have landed (requested during the previous step); publish slab kt + 1 (a loader waits
// for its pieces of it; pieces of slab kt + 2 may stay in flight on a three-stage ring), one barrier; the MFMAs with the refill of
// this slab's ring slot (slab kt + NS) and the reads for slab kt + 1 dealt out behind them
template <class C, int NL, int LID, int CUR>
__device__ __forceinline__ void h3r_step(H3RState<C>& c, int kt, int nk) {
    constexpr int ND = H3Role<C, NL, LID>::ND;
    // (-DX6R_PROF=1, timing experiments: shader-cycle counters of a wave's steps -- [0] waiting for its fragment reads, [1] for its
    //  DMA pieces, [2] in the barrier, [3] in the MFMA stream: see x6r_step)
    unsigned t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (X6R_PROF) {
        t0 = (unsigned)__builtin_readcyclecounter();
        if (c.tprev) c.prof[3] += t0 - c.tprev;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    h3r_fence<C, CUR>(c);
    __builtin_amdgcn_sched_barrier(0);
    if (X6R_PROF) t1 = (unsigned)__builtin_readcyclecounter();
    const bool more = kt + 1 < nk, refill = kt + C::NS < nk;
    if (more) {
        if constexpr (LID >= 0) {
            if (C::NS >= 3 && kt + 2 < nk) wait_vmcnt_imm<(C::NS >= 3 ? ND : 0)>();
            else wait_vmcnt_imm<0>();
        }
        if (X6R_PROF) t2 = (unsigned)__builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else if (X6R_PROF) {
        t2 = t1;
    }
    if (X6R_PROF) {
        t3 = (unsigned)__builtin_readcyclecounter();
        c.prof[0] += t1 - t0, c.prof[1] += t2 - t1, c.prof[2] += t3 - t2;
        c.tprev = t3;
    }
    const int sn = (kt + 1) % C::NS;
    const unsigned a_next = c.a_rd + (unsigned)(sn * C::STAGE), w_next = c.w_rd + (unsigned)(sn * C::STAGE);
    h3r_mma<C, NL, LID, CUR, 0, 0, 0>(c, refill, more, kt + C::NS, kt % C::NS, a_next, w_next);
}
// one tile for this wave's role: ring fill, slab 0 published and read, the K loop (nk >= NS, even)
template <class C, int NL, int LID>
__device__ __forceinline__ void h3r_tile(H3RState<C>& c, int nk) {
    constexpr int ND = H3Role<C, NL, LID>::ND;
    static_assert((C::NS - 1) * ND < 64, "vmcnt is a 6-bit counter");
    h3_issue_mine<C, H3RState<C>, NL, LID>(c, 0, 0);
    h3_issue_mine<C, H3RState<C>, NL, LID>(c, 1, 1);
    if constexpr (C::NS == 3) h3_issue_mine<C, H3RState<C>, NL, LID>(c, 2, 2);
    if constexpr (LID >= 0) wait_vmcnt_imm<(C::NS - 1) * ND>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    h3r_reads<C, 0, 0, C::NREAD>(c, c.a_rd, c.w_rd);  // slab 0: every group, the parity sets into set 0
    for (int kt = 0; kt < nk; kt += 2) {
        h3r_step<C, NL, LID, 0>(c, kt, nk);
        h3r_step<C, NL, LID, 1>(c, kt + 1, nk);
    }
}

}  // namespace
}  // namespace after
