// The pipeline of the split-bf16 GEMM (gemm_x6.hip: fp32 products as six exact bf16 MFMAs on operands that travel as
// three bf16 planes in 1-KB "x6 blocks", common.h) -- tile configuration, per-wave state, LDS-DMA issue, the
// slab step with its side work dealt out behind individual MFMAs.  Shared by gemm_x6.hip (the Linears of the denoiser)
// and conv_x6.hip (the codec's stride-1 convolutions as the same GEMM over shifted row windows: C::CONV).
#pragma once
#include <cstdint>

#include "common.h"
#include "gemm_pipe.h"

namespace after {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// chunk swizzle of the 64-byte plane rows: q = (row / 4) % 4 -> {0, 2, 3, 1}
__device__ __forceinline__ int swz4(int q) { return (0x78 >> (2 * q)) & 3; }

template <int MB_, int NBK_, int KS_, int RS_, int CP_, int NS_, int OUT3_, int RES_, int ACC2_ = 0, int PERSIST_ = 0,
          int CONV_ = 0, int SC1_ = 0, int TIER_ = 0, int SPLIT_ = 0>
struct X6Cfg {
    // TIER 1 (the opt-in bf16 tolerance tier of the batch sampler, after_denoiser_set_gemm_path(h, 3)): only the h x h product of
    // every block is issued -- bf16 operands (the top planes of the exact splits), fp32 accumulate
    static constexpr int TIER = TIER_;
    // SPLIT 0: three bf16 planes per operand, six products per block.  SPLIT 1: TWO fp16 pieces per operand under a power-of-two
    // scale, three products (gemm_h3_pipe.h says why and how exact; here: the every-wave-its-own-pieces ring x6_* below, for
    // conv_x6.hip's GroupNorm-bounded convs -- the loader-wave rings x6l_* / x6r_* have their fp16 twins in gemm_h3_pipe.h)
    static constexpr int SPLIT = SPLIT_, NPL = SPLIT_ ? 2 : 3, NPROD = SPLIT_ ? 3 : 6;
    // SC1: the DMA pieces are sc1 loads -- they miss the CU's vector L1 and are served by the XCD's L2: for operands that
    // another workgroup of the same XCD wrote earlier in the SAME kernel (the clip-per-XCD sampler of denoiser.hip)
    static constexpr int SC1 = SC1_;
    // CONV: the A operand is a window of rows of a longer plane tensor, shifted per tap (conv_x6.hip): a stage holds
    // MB + 1 ALIGNED 16-row blocks per plane (whole 1-KB pieces, whatever the shift) and the fragment reads start
    // shift % 16 rows into them; slab S of the K axis = (tap S / (Cp / 32), channel block S % (Cp / 32))
    static constexpr int CONV = CONV_;
    // PERSIST: a workgroup walks several tiles of one column tile; the next tile's first slabs are in flight while
    // the finished tile's epilogue runs, and its stores drain behind the next tile's MFMAs (launch_x6)
    static constexpr int PERSIST = PERSIST_;
    static constexpr int MB = MB_, NBK = NBK_, KS = KS_, RS = RS_, CP = CP_, NS = NS_, OUT3 = OUT3_, RES = RES_;
    // ACC2: even / odd slabs accumulate into separate registers (two chains of half the length, summed once at
    // the end: the rounding-error growth of a k-part twice as short; for the long-K tiles without a K split)
    static constexpr int ACC2 = ACC2_;
    static constexpr int BM = 16 * MB, BN = 16 * NBK;
    static constexpr int MT = MB / RS, NT = NBK / CP;   // 16x16 blocks per wave
    static constexpr int NW = KS * RS * CP;             // waves
    static constexpr int AB = MB + CONV;                 // 16-row A blocks per plane in a stage
    static constexpr int GA = NPL * AB, GW = NPL * NBK; // 1-KB pieces (16 rows of one plane) per k-part: A, W
    static constexpr int PPK = GA + GW;
    static constexpr int PPKI = TIER ? AB + NBK : PPK;  // pieces a slab's loaders ISSUE (TIER 1: the h planes only; the stage keeps its layout)
    static constexpr int PART = PPK * 1024;             // bytes of one k-part of a stage
    static constexpr int STAGE = KS * PART;
    static constexpr int P = KS * PPK;                  // pieces per stage
    static constexpr int LPS = (P + NW - 1) / NW;       // pieces per wave per slab (the last may be missing)
    static constexpr bool RAGGED = (P % NW) != 0;
    static constexpr int NMMA = NPROD * MT * NT;
    static constexpr int NREAD = NPL * (MT + NT);
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
    u32x4 fa[2][C::NPL][C::MT], fw[2][C::NPL][C::NT];  // fragments of two consecutive slabs: [set][plane][block]
    unsigned voff;                           // per-lane byte offset inside a DMA piece (lane x 16)
    unsigned long long sb[C::LPS];           // wave-uniform source address of each piece's x6 block, slab 0 of its k-part
    unsigned a_rd, w_rd;                     // per-lane LDS byte address of this wave's A / W fragments, stage 0
    unsigned lds0;
    int wid;
    bool full;                               // this wave moves LPS pieces per slab (else LPS - 1)
    // CONV only: per-tap fragment addresses (row shift % 16 folded in), per-tap source offsets of the A pieces
    // (bytes, relative to tap 0's first block), which of this wave's pieces are A pieces, slab -> tap multiplier
    unsigned a_rd3[3];       // tap 0's fragment address, then the increments tap 0 -> 1, 1 -> 2
    int doff[3];             // byte offset of tap 0's A pieces relative to slab x 3072, then the increments tap 0 -> 1, 1 -> 2
    int amask[C::LPS];       // -1: piece i is an A piece, 0: a W piece
    int kslab[C::LPS];       // first slab of piece i's k-part on the whole K axis (taps are a function of the global slab)
    int kslab_w;             // the same for the k-part this wave computes
    int magic;
};

// CONV: tap of slab S (S * magic >> 16 == S / cpb for the slab counts in use)
template <class C>
__device__ __forceinline__ int x6_tap(const X6State<C>& c, int slab) {
    return (slab * c.magic) >> 16;
}

// products in the order of increasing magnitude: (W plane, A plane), 0 = h, 1 = m, 2 = l
constexpr int kWP[6] = {2, 0, 1, 1, 0, 0};
constexpr int kAP[6] = {0, 2, 1, 0, 1, 0};
// ... of the two-piece form (SPLIT 1): 0 = h, 1 = l
constexpr int kWP3[3] = {0, 1, 0};
constexpr int kAP3[3] = {1, 0, 0};

// source address of DMA piece i of slab `slab`: consecutive K blocks of a row group are 3 KB apart; CONV: the A pieces
// of tap t come from the row blocks doff[t] further on, their K block is the slab's channel block
template <class C>
__device__ __forceinline__ unsigned long long x6_src(const X6State<C>& c, int i, int slab) {
    if constexpr (C::CONV) {
        const int t = x6_tap<C>(c, slab + c.kslab[i]);
        // (sign masks instead of selects: the address must stay in scalar registers)
        const int d = (c.doff[0] + (c.doff[1] & ((0 - t) >> 31)) + (c.doff[2] & ((1 - t) >> 31))) & c.amask[i];
        return c.sb[i] + (unsigned long long)(long long)(slab * (C::NPL * 1024) + d);
    } else {
        return c.sb[i] + (unsigned long long)((unsigned)slab * (unsigned)(C::NPL * 1024));
    }
}

template <class C>
__device__ __forceinline__ void x6_dma(const X6State<C>& c, int i, int slab, int stage) {
    lds_dma16<C::SC1 ? 16 : 0>(c.lds0 + (unsigned)(stage * C::STAGE + (c.wid + C::NW * i) * 1024), c.voff, x6_src<C>(c, i, slab));
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
            if constexpr (R < C::NPL * C::MT) {
                constexpr int pl = R / C::MT, i = R % C::MT;
                asm volatile("ds_read_b128 %0, %1 offset:%2"
                             : "=v"(c.fa[NXT][pl][i])
                             : "v"(a_next), "i"((pl * C::AB * 16 + i * 16) * 64));
            } else {
                constexpr int R2 = R - C::NPL * C::MT;
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
        if constexpr (C::SPLIT) {
            typedef _Float16 hf16x8 __attribute__((ext_vector_type(8)));
            c.acc[AS][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hf16x8, c.fw[CUR][kWP3[p]][j]),
                                                                     __builtin_bit_cast(hf16x8, c.fa[CUR][kAP3[p]][i]),
                                                                     c.acc[AS][i][j], 0, 0, 0);
        } else if constexpr (!C::TIER || (kWP[p] == 0 && kAP[p] == 0))
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
    if constexpr (I < C::NPL * C::MT) {
        asm volatile("" : "+v"(c.fa[SET][I / C::MT][I % C::MT]));
        x6_fence_regs<C, SET, I + 1>(c);
    } else if constexpr (I < C::NPL * (C::MT + C::NT)) {
        constexpr int R = I - C::NPL * C::MT;
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
// EXTRA (persistent kernel, first steps of a later tile): younger stores of the previous tile that may stay in flight.
template <class C, int CUR, bool STEADY, int EXTRA = 0>
__device__ __forceinline__ void x6_step(X6State<C>& c, int kt, int nk) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    x6_fence_regs<C, CUR, 0>(c);
    __builtin_amdgcn_sched_barrier(0);
    const bool more = STEADY || kt + 1 < nk, refill = STEADY || kt + C::NS < nk;
    if (more) {
        // in flight after slab kt + 1: slabs kt + 2 .. min(kt + NS - 1, nk - 1)
        if constexpr (STEADY && EXTRA > 0) {
            constexpr int L = C::LPS, L1 = C::LPS > 1 ? C::LPS - 1 : 0;
            if (!C::RAGGED || c.full) wait_vmcnt_imm<(C::NS - 2) * L + EXTRA>();
            else wait_vmcnt_imm<(C::NS - 2) * L1 + EXTRA>();
        } else if constexpr (STEADY) x6_wait<C>(c, C::NS - 2);
        else x6_wait<C>(c, (kt + C::NS - 1 < nk - 1 ? kt + C::NS - 1 : nk - 1) - (kt + 1));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    const int sn = (kt + 1) % C::NS;
    unsigned a_base = c.a_rd;
    if constexpr (C::CONV) {
        // (sign masks, not selects between fields: a select of two loads becomes a load through a selected address and
        // pins the whole state in scratch memory)
        const int t = x6_tap<C>(c, kt + 1 + c.kslab_w);
        a_base = c.a_rd3[0] + (c.a_rd3[1] & (unsigned)((0 - t) >> 31)) + (c.a_rd3[2] & (unsigned)((1 - t) >> 31));
    }
    const unsigned a_next = a_base + (unsigned)(sn * C::STAGE), w_next = c.w_rd + (unsigned)(sn * C::STAGE);
    x6_mma<C, CUR, 0, STEADY>(c, refill, more, kt + C::NS, kt % C::NS, a_next, w_next);
}


// =====================================================================================================
// Loader-wave variant of the same ring (KS = 1 tiles).  In the pipeline above every wave issues its own share of a slab's
// DMA pieces, and a DMA instruction occupies its wave until the CU's one vector-memory path has taken it (~16 cycles per 1-KB
// piece once the path is busy): with eight waves dealing out seven pieces each at the same time, every wave sits in that
// queue for ~900 cycles per slab while the matrix pipes get two MFMAs per piece.  Here ONE wave (wave 0) issues ALL pieces of a
// slab -- one behind each of its MFMAs, W pieces (fabric latency) first -- and the other waves' instruction streams are
// fragment reads and MFMAs only.  The loader's SIMD partner runs its MFMAs while the loader queues; the source address of a
// piece is SALU arithmetic on two wave-uniform bases (no per-piece address registers).
template <class C>
struct X6LState {
    f32x4 acc[C::ACC2 + 1][C::MT][C::NT];
    u32x4 fa[2][3][C::MT], fw[2][3][C::NT];  // fragments of two consecutive slabs: [set][plane][block]
    unsigned voff;                           // per-lane byte offset inside a DMA piece (lane x 16)
    unsigned long long a_src, w_src;         // loader: slab 0 of the tile's first A / W row group, plane h (wave-uniform)
    unsigned rgs;                            // bytes between consecutive 16-row groups of an operand: (K / 32) x 3072
    unsigned a_rd, w_rd;                     // per-lane LDS byte address of this wave's A / W fragments, stage 0
    unsigned lds0;
};

// issue-order item W of a slab -> piece of the stage: the GW weight pieces first, then the GA activation pieces
template <class C, int W>
__device__ __forceinline__ void x6l_dma(const X6LState<C>& c, int slab, int stage) {
    static_assert(C::KS == 1 && C::CONV == 0 && C::SPLIT == 0, "loader-wave ring: plain three-plane tiles without k-parts (fp16 twin: gemm_h3_pipe.h)");
    constexpr int GWI = C::TIER ? C::NBK : C::GW;  // (TIER 1: the issue-order list holds the h-plane pieces only)
    constexpr bool isW = W < GWI;
    constexpr int q = isW ? W : W - GWI;
    constexpr int plane = isW ? q / C::NBK : q / C::MB, grp = isW ? q % C::NBK : q % C::MB;
    constexpr int piece = isW ? C::GA + q : q;  // position inside the stage (1-KB units)
    const unsigned long long src = (isW ? c.w_src : c.a_src) + (unsigned long long)((unsigned)grp * c.rgs + (unsigned)plane * 1024u + (unsigned)slab * 3072u);
    const unsigned dst = c.lds0 + (unsigned)(stage * C::STAGE + piece * 1024);
    lds_dma16<C::SC1 ? 16 : 0>(dst, c.voff, src);
}

// Roles: NL loader waves (waves 0 .. NL - 1: on different SIMDs) share a slab's pieces -- loader LID issues the issue-order
// items LID, LID + NL, ... -- and LID = -1 is a compute wave (fragment reads and MFMAs only).  A single loader has 54 DMA
// instructions to issue per slab beside its own MFMAs and becomes the slowest wave; with all eight waves as loaders every
// wave is in the memory path's queue at the same time.
template <class C, int NL, int LID>
struct X6LRole {
    static constexpr int ND = LID < 0 ? 0 : (C::PPKI - LID + NL - 1) / NL;  // DMA items of this wave per slab
    static constexpr int NWK = ND + C::NREAD;
};

// side-work item W of a slab step: W < ND -> this loader's DMA item W of slab kt + NS into the ring slot just retired; then
// (all waves) the fragment reads of slab kt + 1 (A blocks then W blocks, plane-major) into set NXT
template <class C, int NL, int LID, int NXT, int W, bool STEADY>
__device__ __forceinline__ void x6l_side(X6LState<C>& c, bool refill, bool more, int slab_new, int stage_new, unsigned a_next,
                                         unsigned w_next) {
    constexpr int ND = X6LRole<C, NL, LID>::ND;
    if constexpr (W < ND) {
        if (STEADY || refill) x6l_dma<C, LID + NL * W>(c, slab_new, stage_new);
    } else {
        constexpr int R = W - ND;
        if (STEADY || more) {
            if constexpr (R < 3 * C::MT) {
                constexpr int pl = R / C::MT, i = R % C::MT;
                if constexpr (!C::TIER || pl == 0)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.fa[NXT][pl][i]) : "v"(a_next), "i"((pl * C::AB * 16 + i * 16) * 64));
            } else {
                constexpr int R2 = R - 3 * C::MT;
                constexpr int pl = R2 / C::NT, j = R2 % C::NT;
                if constexpr (!C::TIER || pl == 0)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.fw[NXT][pl][j]) : "v"(w_next), "i"((pl * C::BN + j * 16) * 64));
            }
        }
    }
}

template <class C, int NL, int LID, int NXT, int W, int WEND, bool STEADY>
__device__ __forceinline__ void x6l_sides(X6LState<C>& c, bool refill, bool more, int slab_new, int stage_new, unsigned a_next,
                                          unsigned w_next) {
    if constexpr (W < WEND) {
        x6l_side<C, NL, LID, NXT, W, STEADY>(c, refill, more, slab_new, stage_new, a_next, w_next);
        x6l_sides<C, NL, LID, NXT, W + 1, WEND, STEADY>(c, refill, more, slab_new, stage_new, a_next, w_next);
    }
}

template <class C, int NL, int LID, int CUR, int S, bool STEADY>
__device__ __forceinline__ void x6l_mma(X6LState<C>& c, bool refill, bool more, int slab_new, int stage_new, unsigned a_next,
                                        unsigned w_next) {
    if constexpr (S < C::NMMA) {
        constexpr int p = S / (C::MT * C::NT), i = (S / C::NT) % C::MT, j = S % C::NT;
        constexpr int AS = C::ACC2 ? CUR : 0;
        if constexpr (!C::TIER || (kWP[p] == 0 && kAP[p] == 0))
            c.acc[AS][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, c.fw[CUR][kWP[p]][j]),
                                                                      __builtin_bit_cast(bf16x8, c.fa[CUR][kAP[p]][i]), c.acc[AS][i][j], 0, 0, 0);
        constexpr int NWK = X6LRole<C, NL, LID>::NWK;
        constexpr int w0 = (S * NWK) / C::NMMA, w1 = ((S + 1) * NWK) / C::NMMA;
        x6l_sides<C, NL, LID, CUR ^ 1, w0, w1, STEADY>(c, refill, more, slab_new, stage_new, a_next, w_next);
        __builtin_amdgcn_sched_barrier(0);
        x6l_mma<C, NL, LID, CUR, S + 1, STEADY>(c, refill, more, slab_new, stage_new, a_next, w_next);
    }
}

template <class C, int SET, int I>
__device__ __forceinline__ void x6l_fence_regs(X6LState<C>& c) {
    if constexpr (I < 3 * C::MT) {
        asm volatile("" : "+v"(c.fa[SET][I / C::MT][I % C::MT]));
        x6l_fence_regs<C, SET, I + 1>(c);
    } else if constexpr (I < 3 * (C::MT + C::NT)) {
        constexpr int R = I - 3 * C::MT;
        asm volatile("" : "+v"(c.fw[SET][R / C::NT][R % C::NT]));
        x6l_fence_regs<C, SET, I + 1>(c);
    }
}

// one slab: retire slab kt's fragment reads, publish slab kt + 1 (a loader waits for its pieces -- all but the EXTRA
// youngest vector-memory instructions of the wave, for callers with stores in flight -- then one barrier), then slab kt's
// MFMAs with the side work dealt out behind them.
template <class C, int NL, int LID, int CUR, bool STEADY, int EXTRA = 0>
__device__ __forceinline__ void x6l_step(X6LState<C>& c, int kt, int nk) {
    constexpr int ND = X6LRole<C, NL, LID>::ND;
    static_assert((C::NS - 2) * ND + EXTRA < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    x6l_fence_regs<C, CUR, 0>(c);
    __builtin_amdgcn_sched_barrier(0);
    const bool more = STEADY || kt + 1 < nk, refill = STEADY || kt + C::NS < nk;
    if (more) {
        if constexpr (LID >= 0) {
            // in flight after slab kt + 1: slabs kt + 2 .. min(kt + NS - 1, nk - 1), ND pieces of this wave each
            if (C::NS >= 3 && (STEADY || kt + 2 < nk)) wait_vmcnt_imm<(C::NS >= 3 ? ND : 0) + EXTRA>();
            else wait_vmcnt_imm<EXTRA>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    const int sn = (kt + 1) % C::NS;
    const unsigned a_next = c.a_rd + (unsigned)(sn * C::STAGE), w_next = c.w_rd + (unsigned)(sn * C::STAGE);
    x6l_mma<C, NL, LID, CUR, 0, STEADY>(c, refill, more, kt + C::NS, kt % C::NS, a_next, w_next);
}

// this loader's pieces of one whole slab, back to back (ring fill)
template <class C, int NL, int LID, int W = 0>
__device__ __forceinline__ void x6l_issue_mine(const X6LState<C>& c, int slab, int stage) {
    if constexpr (LID >= 0 && W < X6LRole<C, NL, LID>::ND) {
        x6l_dma<C, LID + NL * W>(c, slab, stage);
        x6l_issue_mine<C, NL, LID, W + 1>(c, slab, stage);
    }
}

// ring fill of a tile for this wave's role: its pieces of slabs 0 .. NS - 1 (nk >= NS)
template <class C, int NL, int LID>
__device__ __forceinline__ void x6l_fill(const X6LState<C>& c) {
    static_assert(C::NS == 2 || C::NS == 3, "ring depth");
    x6l_issue_mine<C, NL, LID>(c, 0, 0);
    x6l_issue_mine<C, NL, LID>(c, 1, 1);
    if constexpr (C::NS == 3) x6l_issue_mine<C, NL, LID>(c, 2, 2);
}

// the K loop of a tile whose ring fill has been issued: publish slab 0, read its fragments, the slab steps.  EXTRA: vector-
// memory instructions this wave issued AFTER the fill (the previous tile's epilogue stores) -- they may stay in flight while
// slabs 0 and 1 are waited for (vmcnt counts loads and stores alike and retires in order).
template <class C, int NL, int LID, int EXTRA = 0>
__device__ __forceinline__ void x6l_main(X6LState<C>& c, int nk) {
    constexpr int ND = X6LRole<C, NL, LID>::ND;
    static_assert((C::NS - 1) * ND + EXTRA < 64, "vmcnt is a 6-bit counter");
    if constexpr (LID >= 0) wait_vmcnt_imm<(C::NS - 1) * ND + EXTRA>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    x6l_sides<C, NL, LID, 0, ND, ND + C::NREAD, true>(c, false, true, 0, 0, c.a_rd, c.w_rd);
    int kt = 0;
    if constexpr (EXTRA > 0) {  // (nk >= NS + 2: both steps are steady)
        x6l_step<C, NL, LID, 0, true, EXTRA>(c, 0, nk);
        x6l_step<C, NL, LID, 1, true, (C::NS == 3 ? EXTRA : 0)>(c, 1, nk);
        kt = 2;
    }
    for (; kt + 1 + C::NS < nk; kt += 2) {  // steady state: slab kt + 1 + NS exists
        x6l_step<C, NL, LID, 0, true>(c, kt, nk);
        x6l_step<C, NL, LID, 1, true>(c, kt + 1, nk);
    }
    for (; kt < nk; kt += 2) {
        x6l_step<C, NL, LID, 0, false>(c, kt, nk);
        if (kt + 1 < nk) x6l_step<C, NL, LID, 1, false>(c, kt + 1, nk);
    }
}


// =====================================================================================================
// ROLLING-fragment variant for the largest tiles (192 x 192, 96 x 256: the fewest operand bytes per MFMA).  Their accumulators
// (72 / 48 registers per wave) leave no room for two sets of fragments, so a plane's fragments are re-read for the NEXT slab
// as soon as the plane's LAST product of the current slab has been issued (an MFMA has read its A / B operands long before
// the LDS data returns); only the activations' h plane -- used by the first and the last product -- is double-buffered.
// Products in the order (W plane, A plane) = (h,h) (h,m) (h,l) (m,m) (m,h) (l,h): W h and A l are free after the third
// product, A m after the fourth, W m after the fifth, W l after the sixth -- its NT reads are the only ones that nothing hides.
// Two-stage ring, loader waves 0 .. NL - 1 (x6l_* above), one barrier per slab: all reads of a step fetch the NEXT slab, whose
// slot the loaders refill one step later.
#ifndef X6R_PROF
#define X6R_PROF 0
#endif
template <int MB_, int NBK_, int RS_, int CP_, int SC1_ = 0, int TIER_ = 0>
struct X6RCfg {
    static constexpr int MB = MB_, NBK = NBK_, RS = RS_, CP = CP_, SC1 = SC1_, NS = 2, KS = 1, CONV = 0, ACC2 = 0, TIER = TIER_;
    static constexpr int SPLIT = 0, NPL = 3;
    static constexpr int BM = 16 * MB, BN = 16 * NBK, MT = MB / RS, NT = NBK / CP, NW = RS * CP, AB = MB;
    static constexpr int GA = 3 * MB, GW = 3 * NBK, PPK = GA + GW, STAGE = PPK * 1024;
    static constexpr int PPKI = TIER ? MB + NBK : PPK;  // pieces ISSUED per slab (TIER 1: the h planes only; the stage keeps its layout)
    static constexpr int PN = MT * NT, NMMA = 6 * PN, NREAD = 3 * (MT + NT);
    static_assert(MB % RS == 0 && NBK % CP == 0 && NW == 8, "tile shape (eight waves)");
    static_assert(NS * STAGE <= 160 * 1024, "ring exceeds the LDS");
    static_assert((2 * BM + BM) * 64 < 65536 && (2 * BN + BN) * 64 < 65536, "fragment offsets are 16-bit immediates");
};

template <class C>
struct X6RState {
    f32x4 acc[C::MT][C::NT];
    u32x4 w[3][C::NT];     // W fragments [plane h, m, l][column block]
    u32x4 al[C::MT], am[C::MT], ah[2][C::MT];  // A fragments; h: two sets, by slab parity
    unsigned voff;
    unsigned long long a_src, w_src;  // loaders: slab 0 of the tile's first A / W row group, plane h
    unsigned rgs;                     // bytes between consecutive 16-row groups of an operand: (K / 32) x 3072
    unsigned a_rd, w_rd, lds0;
    unsigned prof[4], tprev;          // (X6R_PROF)
};

template <class C, int I>
__device__ __forceinline__ void x6r_dma(const X6RState<C>& c, int slab, int stage) {  // issue-order item I: W pieces first
    constexpr int GWI = C::TIER ? C::NBK : C::GW;
    constexpr bool isW = I < GWI;
    constexpr int q = isW ? I : I - GWI;
    constexpr int plane = isW ? q / C::NBK : q / C::MB, grp = isW ? q % C::NBK : q % C::MB;
    constexpr int piece = isW ? C::GA + q : q;
    const unsigned long long src = (isW ? c.w_src : c.a_src) + (unsigned long long)((unsigned)grp * c.rgs + (unsigned)plane * 1024u + (unsigned)slab * 3072u);
    const unsigned dst = c.lds0 + (unsigned)(stage * C::STAGE + piece * 1024);
    lds_dma16<C::SC1 ? 16 : 0>(dst, c.voff, src);
}
template <class C, int NL, int LID, int W = 0>
__device__ __forceinline__ void x6r_issue_mine(const X6RState<C>& c, int slab, int stage) {
    if constexpr (LID >= 0 && LID + NL * W < C::PPKI) {
        x6r_dma<C, LID + NL * W>(c, slab, stage);
        x6r_issue_mine<C, NL, LID, W + 1>(c, slab, stage);
    }
}

// fragment read R of the groups, in re-read order: [0, MT) A h (the other set) | [MT, MT + NT) W h | then A l (MT) | A m (MT) |
// W m (NT) | W l (NT)
template <class C, int NXT, int R>
__device__ __forceinline__ void x6r_read(X6RState<C>& c, unsigned a_base, unsigned w_base) {
    constexpr int MT = C::MT, NT = C::NT;
    if constexpr (R < MT) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.ah[NXT][R]) : "v"(a_base), "i"((0 * C::BM + R * 16) * 64));
    } else if constexpr (R < MT + NT) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.w[0][R - MT]) : "v"(w_base), "i"((0 * C::BN + (R - MT) * 16) * 64));
    } else if constexpr (C::TIER) {
        // (the m and l planes are neither fetched nor read)
    } else if constexpr (R < 2 * MT + NT) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.al[R - MT - NT]) : "v"(a_base), "i"((2 * C::BM + (R - MT - NT) * 16) * 64));
    } else if constexpr (R < 3 * MT + NT) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.am[R - 2 * MT - NT]) : "v"(a_base), "i"((1 * C::BM + (R - 2 * MT - NT) * 16) * 64));
    } else if constexpr (R < 3 * MT + 2 * NT) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.w[1][R - 3 * MT - NT]) : "v"(w_base), "i"((1 * C::BN + (R - 3 * MT - NT) * 16) * 64));
    } else {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.w[2][R - 3 * MT - 2 * NT]) : "v"(w_base), "i"((2 * C::BN + (R - 3 * MT - 2 * NT) * 16) * 64));
    }
}
template <class C, int NXT, int R, int REND>
__device__ __forceinline__ void x6r_reads(X6RState<C>& c, unsigned a_base, unsigned w_base) {
    if constexpr (R < REND) {
        x6r_read<C, NXT, R>(c, a_base, w_base);
        x6r_reads<C, NXT, R + 1, REND>(c, a_base, w_base);
    }
}

// the side work behind MFMA S of a step: the earliest MFMA after which read R may go out (its group's last product is issued),
// one read per MFMA from there on; the loader's DMA items one per DSP MFMAs from the start
template <class C>
struct X6RSched {
    static constexpr int MT = C::MT, NT = C::NT, PN = C::PN;
    // first MFMA index behind which read R may be issued
    static constexpr int ready(int R) {
        return R < MT ? 0                          // A h, other set: any time
               : R < 2 * MT + NT ? 3 * PN - 1      // W h, A l: after the third product
               : R < 3 * MT + NT ? 4 * PN - 1      // A m: after the fourth
               : R < 3 * MT + 2 * NT ? 5 * PN - 1  // W m: after the fifth
                                     : 6 * PN - 1; // W l: after the sixth (the end)
    }
    // reads are issued in order R = 0, 1, ..: slot(R) = max(ready(R), slot(R - 1) + 1)
    static constexpr int slot(int R) {
        int s = -1;
        for (int r = 0; r <= R; ++r) {
            const int rd = ready(r);
            s = rd > s + 1 ? rd : s + 1;
            if (s > C::NMMA - 1) s = C::NMMA - 1;  // (the tail: everything left goes behind the last MFMA)
        }
        return s;
    }
};

template <class C, int NL, int LID, int CUR, int S, int R, int D>
__device__ __forceinline__ void x6r_mma(X6RState<C>& c, bool refill, bool more, int slab_new, int stage_new, unsigned a_next, unsigned w_next) {
    if constexpr (S < C::NMMA) {
        constexpr int PN = C::PN, p = S / PN, i = (S % PN) / C::NT, j = S % C::NT;
        constexpr int WP[6] = {0, 0, 0, 1, 1, 2}, AP[6] = {0, 1, 2, 1, 0, 0};  // (W plane, A plane) of product p; 0 = h, 1 = m, 2 = l
        const u32x4& af = AP[p] == 0 ? c.ah[CUR][i] : (AP[p] == 1 ? c.am[i] : c.al[i]);
        if constexpr (!C::TIER || p == 0)  // (TIER 1: the h x h product only -- X6Cfg::TIER)
            c.acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, c.w[WP[p]][j]), __builtin_bit_cast(bf16x8, af), c.acc[i][j], 0, 0, 0);
        // side work behind this MFMA: DMA items (loaders: one per DSP MFMAs), then the reads whose slot this is
        constexpr int ND = LID < 0 ? 0 : (C::PPKI - LID + NL - 1) / NL;
        constexpr int DSP = ND > 0 ? (C::NMMA / 2) / ND > 0 ? (C::NMMA / 2) / ND : 1 : 1;
        constexpr bool dma_here = D < ND && S == D * DSP;
        if constexpr (dma_here) {
            if (refill) x6r_dma<C, LID + NL * D>(c, slab_new, stage_new);
        }
        constexpr int NR = C::NREAD;
        constexpr int r1 = [] {
            int r = R;
            while (r < NR && X6RSched<C>::slot(r) <= S) ++r;
            return r;
        }();
        if constexpr (r1 > R) {
            if (more) x6r_reads<C, CUR ^ 1, R, r1>(c, a_next, w_next);
        }
        __builtin_amdgcn_sched_barrier(0);
        x6r_mma<C, NL, LID, CUR, S + 1, r1, (dma_here ? D + 1 : D)>(c, refill, more, slab_new, stage_new, a_next, w_next);
    }
}

template <class C, int SET>
__device__ __forceinline__ void x6r_fence(X6RState<C>& c) {
#pragma unroll
    for (int i = 0; i < C::MT; ++i) asm volatile("" : "+v"(c.ah[SET][i]), "+v"(c.al[i]), "+v"(c.am[i]));
#pragma unroll
    for (int j = 0; j < C::NT; ++j) asm volatile("" : "+v"(c.w[0][j]), "+v"(c.w[1][j]), "+v"(c.w[2][j]));
}

// one slab (kt: CUR = kt & 1): every fragment of this slab has landed (they were all requested during the previous step);
// publish slab kt + 1 (a loader waits for its pieces), one barrier; the MFMAs with the refill of this slab's ring slot (slab
// kt + 2) and the re-reads for slab kt + 1 dealt out behind them
// (-DX6R_PROF=1, timing experiments: shader-cycle counters of a wave's steps -- [0] waiting for its fragment reads, [1] for its
//  DMA pieces, [2] in the barrier, [3] in the MFMA stream (up to the next step's entry); s_memtime is an lgkmcnt operation: each
//  stamp costs its own round trip, ~3 x 50 cycles per step)
template <class C, int NL, int LID, int CUR>
__device__ __forceinline__ void x6r_step(X6RState<C>& c, int kt, int nk) {
    unsigned t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (X6R_PROF) {
        t0 = (unsigned)__builtin_readcyclecounter();
        if (c.tprev) c.prof[3] += t0 - c.tprev;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    x6r_fence<C, CUR>(c);
    __builtin_amdgcn_sched_barrier(0);
    if (X6R_PROF) t1 = (unsigned)__builtin_readcyclecounter();
    const bool more = kt + 1 < nk, refill = kt + 2 < nk;
    if (more) {
        if constexpr (LID >= 0) wait_vmcnt_imm<0>();
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
    const int sn = (kt + 1) & 1;
    const unsigned a_next = c.a_rd + (unsigned)(sn * C::STAGE), w_next = c.w_rd + (unsigned)(sn * C::STAGE);
    x6r_mma<C, NL, LID, CUR, 0, 0, 0>(c, refill, more, kt + 2, kt & 1, a_next, w_next);
}

// one tile for this wave's role: ring fill, slab 0 published and read, the K loop (nk >= 2, even)
template <class C, int NL, int LID>
__device__ __forceinline__ void x6r_tile(X6RState<C>& c, int nk) {
    constexpr int ND = LID < 0 ? 0 : (C::PPKI - LID + NL - 1) / NL;
    static_assert(ND < 64, "vmcnt is a 6-bit counter");
    x6r_issue_mine<C, NL, LID>(c, 0, 0);
    x6r_issue_mine<C, NL, LID>(c, 1, 1);
    if constexpr (LID >= 0) wait_vmcnt_imm<ND>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    x6r_reads<C, 0, 0, C::NREAD>(c, c.a_rd, c.w_rd);  // slab 0: every group, A h into set 0
    for (int kt = 0; kt < nk; kt += 2) {
        x6r_step<C, NL, LID, 0>(c, kt, nk);
        x6r_step<C, NL, LID, 1>(c, kt + 1, nk);
    }
}

}  // namespace
}  // namespace after
