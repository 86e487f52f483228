// Software pipeline of the LDS-DMA GEMM kernels (gemm.hip) and of the time-major conv kernel
// (conv_tm.hip), as file-scope macros: hipcc rejects asm operands that are lambda captures.
// See gemm.hip for the design notes (one wave cannot hide its own side work behind its own
// MFMAs; counted vmcnt; one raw s_barrier per 32-deep slab; source-side XOR swizzle).
#pragma once

namespace after {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
// One LDS-DMA piece (global_load_lds_dwordx4): every lane moves the 16 bytes at base + voff -- base wave-uniform (scalar registers:
// the instruction's saddr), voff per lane -- to LDS address dst + lane x 16, dst wave-uniform (M0).  The builtin lets the COMPILER
// write M0 (rounds 1 - 5 wrote it in inline asm, `s_mov_b32 m0`, which cannot declare the clobber: safe on the pinned hipcc only).
// AUX: cache policy bits (16 = sc1: miss the vector L1, served by the XCD's L2).
template <int AUX = 0>
__device__ __forceinline__ void lds_dma16(unsigned dst, unsigned voff, unsigned long long base) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(reinterpret_cast<const char*>(base) + voff),
                                     (lds_ptr_t)(__attribute__((address_space(3))) char*)(unsigned long)dst, 16, 0, AUX);
}
}  // namespace after

// ---- the software pipeline of the DMA kernels as file-scope macros (hipcc rejects asm
// operands that are lambda captures).  They expect in scope: NS, STAGE, KK, MT, NT, BK, LPS,
// nk, a_c[], w_c[], fa[][][], fb[][][], acc[][], issue(slab, slot).
namespace after {
template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
}  // namespace after
#define AFTER_GEMM_LOAD_FRAGS(p, slab)                                                             \
    {                                                                                              \
        const unsigned so__ = (unsigned)((slab) % NS) * (STAGE * 4);                               \
        _Pragma("unroll") for (int kk = 0; kk < KK; ++kk) {                                        \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                         \
                asm volatile("ds_read_b128 %0, %1" : "=v"(fa[kk][p][i]) : "v"(a_c[kk] + so__ + i * 16 * BK * 4)); \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                         \
                asm volatile("ds_read_b128 %0, %1" : "=v"(fb[kk][p][j]) : "v"(w_c[kk] + so__ + j * 16 * BK * 4)); \
        }                                                                                          \
    }
#define AFTER_GEMM_FENCE_FRAGS(p)                                            \
    {                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   \
        _Pragma("unroll") for (int kk = 0; kk < KK; ++kk) {                  \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[kk][p][i])); \
            _Pragma("unroll") for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(fb[kk][p][j])); \
        }                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                   \
    }
#define AFTER_GEMM_MFMA_STEP(A_, B_, comp)                                                      \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[i][comp], B_[j][comp], acc[i][j], 0, 0, 0);
#define AFTER_GEMM_MMA(p)                                 \
    _Pragma("unroll") for (int kk = 0; kk < KK; ++kk) {   \
        AFTER_GEMM_MFMA_STEP(fa[kk][p], fb[kk][p], 0)     \
        AFTER_GEMM_MFMA_STEP(fa[kk][p], fb[kk][p], 1)     \
        AFTER_GEMM_MFMA_STEP(fa[kk][p], fb[kk][p], 2)     \
        AFTER_GEMM_MFMA_STEP(fa[kk][p], fb[kk][p], 3)     \
    }
// wait until slab s_ has landed: slabs s_+1 .. min(last_, nk-1) may stay in flight
#define AFTER_GEMM_WAIT_SLAB(s_, last_)                                                 \
    {                                                                                   \
        const int rem__ = ((last_) < nk - 1 ? (last_) : nk - 1) - (s_);                 \
        if (rem__ >= 3 && NS >= 4) wait_vmcnt_imm<(NS >= 4 ? 3 : 0) * LPS>();          \
        else if (rem__ >= 2 && NS >= 3) wait_vmcnt_imm<(NS >= 3 ? 2 : 0) * LPS>();     \
        else if (rem__ >= 1 && NS >= 2) wait_vmcnt_imm<(NS >= 2 ? 1 : 0) * LPS>();     \
        else wait_vmcnt_imm<0>();                                                       \
    }
// one slab: retire slab kt's fragment reads, publish slab kt+1 (one barrier), refill the
// freed ring slot with slab kt+NS, start reading slab kt+1's fragments, then slab kt's MFMAs
#define AFTER_GEMM_STEP(pc, pn, kt_)                                                        \
    {                                                                                       \
        const int kt__ = (kt_);                                                             \
        unsigned long long p0__ = 0, p1__ = 0, p2__ = 0, p3__ = 0;                          \
        if (g.dbg) p0__ = __builtin_readcyclecounter();                                     \
        AFTER_GEMM_FENCE_FRAGS(pc)                                                          \
        if (g.dbg) p1__ = __builtin_readcyclecounter();                                     \
        if (kt__ + 1 < nk) {                                                                \
            if (kt__ + NS - 1 <= nk - 1) {                                                  \
                wait_vmcnt_imm<(NS - 2) * LPS>();                                           \
            } else {                                                                        \
                AFTER_GEMM_WAIT_SLAB(kt__ + 1, kt__ + NS - 1)                               \
            }                                                                               \
            if (g.dbg) p2__ = __builtin_readcyclecounter();                                 \
            __builtin_amdgcn_s_barrier();                                                   \
            asm volatile("" ::: "memory");                                                  \
            if (g.dbg) p3__ = __builtin_readcyclecounter();                                 \
            if (kt__ + NS < nk) issue(kt__ + NS, kt__ % NS);                                \
            AFTER_GEMM_LOAD_FRAGS(pn, kt__ + 1)                                             \
        }                                                                                   \
        if (g.dbg && kt__ + 1 < nk) {                                                       \
            ph_fence += p1__ - p0__;                                                        \
            ph_vm += p2__ - p1__;                                                           \
            ph_bar += p3__ - p2__;                                                          \
        }                                                                                   \
        AFTER_GEMM_MMA(pc)                                                                  \
    }

// Interleaved slab step for kernels that run ONE wave per SIMD: nothing else can feed the matrix
// pipe while this wave issues the next slab's DMA and fragment reads, so those instructions are
// dealt out one by one behind individual MFMAs (each MFMA occupies the pipe for 32 cycles; the
// wave's issue slot is free meanwhile).  Work item w of the slab: w < LPS -> DMA piece w of slab
// kt+NS into the ring slot just retired; then the KK*(MT+NT) fragment reads of slab kt+1.
// Expects additionally: src[], wid, smem, RPP.
#define AFTER_GEMM_STEP_IL(pc, pn, kt_, steady_)                                                     \
    {                                                                                         \
        const int kt__ = (kt_);                                                               \
        constexpr int NMMA__ = KK * 4 * MT * NT, NWORK__ = LPS + KK * (MT + NT);              \
        constexpr int SP__ = NMMA__ / NWORK__ > 0 ? NMMA__ / NWORK__ : 1;                     \
        static_assert(NWORK__ <= NMMA__, "more side work than MFMA slots");                   \
        unsigned long long p0__ = 0, p1__ = 0, p2__ = 0, p3__ = 0;                            \
        if (g.dbg) p0__ = __builtin_readcyclecounter();                                       \
        AFTER_GEMM_FENCE_FRAGS(pc)                                                            \
        if (g.dbg) p1__ = __builtin_readcyclecounter();                                       \
        const bool more__ = (steady_) || kt__ + 1 < nk, refill__ = (steady_) || kt__ + NS < nk; \
        if (more__) {                                                                         \
            if ((steady_) || kt__ + NS - 1 <= nk - 1) {                                       \
                wait_vmcnt_imm<(NS - 2) * LPS>();                                             \
            } else {                                                                          \
                AFTER_GEMM_WAIT_SLAB(kt__ + 1, kt__ + NS - 1)                                 \
            }                                                                                 \
            if (g.dbg) p2__ = __builtin_readcyclecounter();                                   \
            __builtin_amdgcn_s_barrier();                                                     \
            asm volatile("" ::: "memory");                                                    \
            if (g.dbg) p3__ = __builtin_readcyclecounter();                                   \
        }                                                                                     \
        if (g.dbg && more__) {                                                                \
            ph_fence += p1__ - p0__;                                                          \
            ph_vm += p2__ - p1__;                                                             \
            ph_bar += p3__ - p2__;                                                            \
        }                                                                                     \
        const unsigned so__ = (unsigned)((kt__ + 1) % NS) * (STAGE * 4);                      \
        const int rs__ = kt__ % NS;                                                           \
        _Pragma("unroll") for (int kk = 0; kk < KK; ++kk)                                     \
        _Pragma("unroll") for (int comp = 0; comp < 4; ++comp)                                \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                        \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                      \
            /* W fragment as srcA: the accumulator holds C^T, i.e. four consecutive output      \
               columns of one row per lane -> float4 stores / residual loads in the epilogue */ \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[kk][pc][j][comp], fa[kk][pc][i][comp], \
                                                             acc[i][j], 0, 0, 0);             \
            const int slot__ = ((kk * 4 + comp) * MT + i) * NT + j;                           \
            if (slot__ % SP__ == 0 && slot__ / SP__ < NWORK__) {                              \
                const int w__ = slot__ / SP__;                                                \
                if (w__ < LPS) {                                                              \
                    if (refill__) AFTER_BAL_DMA((w__ < LPS ? w__ : 0), kt__ + NS, rs__)       \
                } else if (more__) {                                                          \
                    const int r__ = w__ - LPS, k2__ = r__ / (MT + NT), q__ = r__ % (MT + NT); \
                    if (q__ < MT) {                                                           \
                        asm volatile("ds_read_b128 %0, %1"                                    \
                                     : "=v"(fa[k2__ < KK ? k2__ : 0][pn][q__ < MT ? q__ : 0]) \
                                     : "v"(a_c[k2__ < KK ? k2__ : 0] + so__ + q__ * 16 * BK * 4)); \
                    } else {                                                                  \
                        asm volatile("ds_read_b128 %0, %1"                                    \
                                     : "=v"(fb[k2__ < KK ? k2__ : 0][pn][q__ >= MT ? q__ - MT : 0]) \
                                     : "v"(w_c[k2__ < KK ? k2__ : 0] + so__ + (q__ - MT) * 16 * BK * 4)); \
                    }                                                                         \
                }                                                                             \
            }                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                \
        }                                                                                     \
    }

