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
          int CONV_ = 0>
struct X6Cfg {
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
    static constexpr int GA = 3 * AB, GW = 3 * NBK;     // 1-KB pieces (16 rows of one plane) per k-part: A, W
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

// source address of DMA piece i of slab `slab`: consecutive K blocks of a row group are 3 KB apart; CONV: the A pieces
// of tap t come from the row blocks doff[t] further on, their K block is the slab's channel block
template <class C>
__device__ __forceinline__ unsigned long long x6_src(const X6State<C>& c, int i, int slab) {
    if constexpr (C::CONV) {
        const int t = x6_tap<C>(c, slab + c.kslab[i]);
        // (sign masks instead of selects: the address must stay in scalar registers)
        const int d = (c.doff[0] + (c.doff[1] & ((0 - t) >> 31)) + (c.doff[2] & ((1 - t) >> 31))) & c.amask[i];
        return c.sb[i] + (unsigned long long)(long long)(slab * 3072 + d);
    } else {
        return c.sb[i] + (unsigned long long)((unsigned)slab * 3072u);
    }
}

template <class C>
__device__ __forceinline__ void x6_dma(const X6State<C>& c, int i, int slab, int stage) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :
                 : "s"(c.lds0 + (unsigned)(stage * C::STAGE + (c.wid + C::NW * i) * 1024)), "v"(c.voff),
                   "s"(x6_src<C>(c, i, slab))
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
                             : "v"(a_next), "i"((pl * C::AB * 16 + i * 16) * 64));
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

}  // namespace
}  // namespace after
