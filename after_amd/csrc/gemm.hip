// fp32 GEMM on the CDNA4 matrix cores: C[M,N] = epi(A[M,K] * W[N,K]^T + bias).
//
// Used for every Linear of the denoiser (reference transformerv2.py:251,275-283,
// 330-334,387-398,430,488-492 -- all `nn.Linear`, i.e. x @ W^T + b with W stored
// [out, in]) so activations and weights are both K-contiguous ("B^T input").  Everything
// accumulates with v_mfma_f32_16x16x4_f32 (exact fp32: the reference computes in fp32,
// SURVEY.md 2.1).  Four kernel families, chosen per shape in launch_gemm_cfg:
//
//   gemm_f32_bal_kernel     the workhorse: balanced split-K / row-split tiles of 16x16 blocks
//                           sized so that M = 768 B tokens fills 256 CUs with equal work, >= 2
//                           waves per SIMD, operands by asm LDS-DMA, deterministic LDS
//                           reduction of the k-parts; MODE 1 fuses out_proj + CFG + Euler
//   gemm_f32_skinny_kernel  <= 48 tokens (streaming): column-owning workgroups, 8-way K split
//                           inside the workgroup, fragments straight from global memory
//   gemm_f32_dma_kernel     first-generation 2x2-wave tiles with an LDS-DMA ring (small
//                           Linears, K % 64 != 0)
//   gemm_f32_kernel         register-staged fallback for K % 32 != 0
//
// Common ground (gfx950): K is walked in 32-deep slabs staged through LDS; a tile row in LDS is
// the plain 128-byte k-slab of that row and bank conflicts of the ds_read_b128 fragment reads
// are removed by an XOR swizzle applied on the DMA SOURCE address; one raw s_barrier per slab
// with counted vmcnt; workgroup ids are remapped so that the 8 XCDs (private 4 MiB L2s) each
// own a compact part of the tile matrix.  DESIGN.md section 4 has the measurements behind the
// choices (matrix-pipe co-issue, SIMD quantisation, XCD traffic).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_pipe.h"

namespace after {

namespace {

// LDS row = BK + 8 floats: == 40 (mod 64) for BK = 32 and == 8 (mod 64)... see lds_ld()
template <int BK>
constexpr int lds_ld() { return BK == 32 ? 40 : BK + 40; }  // (LD/4) == 10 (mod 16): conflict-free b128 fragments

template <int NL>
__device__ __forceinline__ void tile_gload(float4 (&r)[NL], const float* __restrict__ base,
                                           const size_t (&off)[NL], const bool (&ok)[NL], int k0,
                                           bool kok) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
        r[i] = (ok[i] && kok) ? *reinterpret_cast<const float4*>(base + off[i] + k0)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
}

// rows lrow + 32*i of one LDS tile buffer (dst already points at [lrow][lc4])
template <int NL, int ROWS, int LD>
__device__ __forceinline__ void tile_sstore(const float4 (&r)[NL], float* dst) {
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<float4*>(dst + ROWS * i * LD) = r[i];
}

template <int MT, int NT, int BK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int BM = 32 * MT, BN = 32 * NT;  // workgroup tile
    constexpr int LDS_LD = lds_ld<BK>();
    constexpr int TPR = BK / 4;         // threads per tile row (one float4 each)
    constexpr int RPP = 256 / TPR;      // tile rows per pass of the 256 threads
    constexpr int A_LOADS = BM / RPP, W_LOADS = BN / RPP;  // float4 per thread per slab
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                    // [2][BM][LDS_LD]
    float* Ws = smem + 2 * BM * LDS_LD;  // [2][BN][LDS_LD]

    // XCD-aware bijective remap (cdna guide T1): block b runs on XCD b % 8.
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tn = bid / tiles_m, tm = bid - tn * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid >> 1) * (16 * MT), wn0 = (wid & 1) * (16 * NT);
    const int lrow = tid / TPR, lc4 = (tid % TPR) * 4;

    const float* __restrict__ gA = g.A;
    const float* __restrict__ gW = g.W;
    const int M = g.M, N = g.N, K = g.K, lda = g.lda, ldw = g.ldw;

    size_t Aoff[A_LOADS], Woff[W_LOADS];
    bool Aok[A_LOADS], Wok[W_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int gm = m0 + lrow + RPP * i;
        Aok[i] = gm < M;
        Aoff[i] = (size_t)(Aok[i] ? gm : 0) * lda + lc4;
    }
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) {
        const int gn = n0 + lrow + RPP * i;
        Wok[i] = gn < N;
        Woff[i] = (size_t)(Wok[i] ? gn : 0) * ldw + lc4;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 ra[A_LOADS], rw[W_LOADS];
    const int nk = (K + BK - 1) / BK;
    tile_gload<A_LOADS>(ra, gA, Aoff, Aok, 0, lc4 < K);
    tile_gload<W_LOADS>(rw, gW, Woff, Wok, 0, lc4 < K);
    tile_sstore<A_LOADS, RPP, LDS_LD>(ra, As + lrow * LDS_LD + lc4);
    tile_sstore<W_LOADS, RPP, LDS_LD>(rw, Ws + lrow * LDS_LD + lc4);
    __syncthreads();
    // fragment addressing of v_mfma_f32_16x16x4_f32: lane l supplies A[i = l&15][k = l>>4]
    // and B[k = l>>4][j = l&15]; one b128 read = k-quad (l>>4) of a 16-deep k block.
    const int frow = lane & 15, fk = (lane >> 4) * 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            const int k0 = (kt + 1) * BK;
            tile_gload<A_LOADS>(ra, gA, Aoff, Aok, k0, (k0 + lc4) < K);
            tile_gload<W_LOADS>(rw, gW, Woff, Wok, k0, (k0 + lc4) < K);
        }
        const float* Ab = &As[(cur * BM + wm0 + frow) * LDS_LD + fk];
        const float* Wb = &Ws[(cur * BN + wn0 + frow) * LDS_LD + fk];
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            float4 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                a[i] = *reinterpret_cast<const float4*>(Ab + i * 16 * LDS_LD + kk * 16);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                b[j] = *reinterpret_cast<const float4*>(Wb + j * 16 * LDS_LD + kk * 16);
            // k-step outermost: consecutive MFMAs hit different accumulators, so the
            // 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 hides behind the
            // 32-cycle issue interval
#define AFTER_MFMA_STEP(comp)                                                                   \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].comp, b[j].comp, acc[i][j], 0, 0, 0);
            AFTER_MFMA_STEP(x)
            AFTER_MFMA_STEP(y)
            AFTER_MFMA_STEP(z)
            AFTER_MFMA_STEP(w)
#undef AFTER_MFMA_STEP
        }
        if (kt + 1 < nk) {
            tile_sstore<A_LOADS, RPP, LDS_LD>(ra, As + ((cur ^ 1) * BM + lrow) * LDS_LD + lc4);
            tile_sstore<W_LOADS, RPP, LDS_LD>(rw, Ws + ((cur ^ 1) * BN + lrow) * LDS_LD + lc4);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
    const int ccol = lane & 15, crow0 = 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int gn = n0 + wn0 + j * 16 + ccol;
        if (gn >= N) continue;
        const float bv = g.bias ? g.bias[gn] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm0 + i * 16 + crow0 + r;
                if (gm >= M) continue;
                float v = acc[i][j][r] + bv;
                if (g.epilogue == EPI_GELU) v = gelu_erf(v);
                if (g.epilogue == EPI_RESIDUAL) v += g.R[(size_t)gm * g.ldr + gn];
                if (g.epilogue == EPI_RELU) v = fmaxf(v, 0.f);
                if (g.epilogue == EPI_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                g.C[(size_t)gm * g.ldc + gn] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// LDS-DMA variant (the fast path when K is a multiple of 32): the K slabs are streamed
// straight from global memory into an NS-deep LDS ring with global_load_lds_dwordx4
// (no VGPR round trip, no ds_write), NS-1 slabs ahead of the MFMAs, and the ring is
// guarded by ONE raw s_barrier per slab plus a counted s_waitcnt vmcnt(N) -- the loads
// of the following slabs stay in flight across the barrier (cdna guide section 5,
// "Pipelining across barriers").  The DMA writes lane-linear (wave base + lane*16 B),
// so a tile row is the plain 128-byte k-slab of that row; bank conflicts of the
// ds_read_b128 fragment reads are removed by an XOR swizzle applied on the SOURCE
// address: 16-byte chunk c of row r is fetched into chunk position c ^ (r & 7), and
// the fragment read of chunk c of row r looks at position c ^ (r & 7) (conflict free
// for every 16-lane service group; derivation in DESIGN.md).  M/N edges read a clamped
// row (its results are never stored); there is no K tail on this path.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else static_assert(N < 0, "add the vmcnt literal");
}

template <int MT, int NT, int NS, int BK>
__global__ __launch_bounds__(256) void gemm_f32_dma_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    constexpr int CPR = BK / 4;                    // 16-byte chunks per tile row
    constexpr int RPP = 64 / CPR;                  // rows per 1 KiB DMA piece (one wave instruction)
    constexpr int A_PW = BM / RPP / 4, W_PW = BN / RPP / 4;  // DMA pieces per wave per slab
    constexpr int LPS = A_PW + W_PW;               // DMA instructions per wave per slab
    constexpr int STAGE = (BM + BN) * BK;          // floats per ring slot
    constexpr int KK = BK / 16;                    // 16-deep k blocks per slab
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tn = bid / tiles_m, tm = bid - tn * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wid >> 1) * (16 * MT), wn0 = (wid & 1) * (16 * NT);
    const int M = g.M, N = g.N, K = g.K;

    // ---- DMA source pointers (per lane) and ring offsets (wave uniform).  Source-side
    // swizzle: ring position `pos` of row r receives global chunk pos ^ (r & (CPR-1)).
    const int rsub = lane / CPR, pos = lane % CPR;
    const float* asrc[A_PW];
    const float* wsrc[W_PW];
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
        const int row = (wid * A_PW + i) * RPP + rsub;
        const int gm = min(m0 + row, M - 1);
        asrc[i] = g.A + (size_t)gm * g.lda + (pos ^ (row & (CPR - 1))) * 4;
    }
#pragma unroll
    for (int i = 0; i < W_PW; ++i) {
        const int row = (wid * W_PW + i) * RPP + rsub;
        const int gn = min(n0 + row, N - 1);
        wsrc[i] = g.W + (size_t)gn * g.ldw + (pos ^ (row & (CPR - 1))) * 4;
    }
    auto issue = [&](int slab, int slot) {
        float* st = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < A_PW; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(asrc[i] + slab * BK),
                                             (lds_ptr_t)(st + (wid * A_PW + i) * RPP * BK), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < W_PW; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc[i] + slab * BK),
                                             (lds_ptr_t)(st + BM * BK + (wid * W_PW + i) * RPP * BK), 16, 0, 0);
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    unsigned long long t_start = 0, t_loop = 0, t_end = 0, r_start = 0;
    if (g.dbg) {
        t_start = __builtin_readcyclecounter();
        r_start = wall_clock64();  // 100 MHz device-wide counter
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < nk) issue(s, s);

    // fragment addressing: lane l -> row i = l & 15, k-quad kq = l >> 4; chunk (kq + 4 kk) of
    // row r sits at position (kq + 4 kk) ^ (r & 7)
    const int frow = lane & 15, kq = lane >> 4, sw = frow & (CPR - 1);
    const int aoff = (wm0 + frow) * BK, woff = BM * BK + (wn0 + frow) * BK;

    // Software pipeline: while the MFMAs of slab kt run, the fragments of slab kt+1 are
    // already being read from LDS into the other register set (F0 / F1 alternate, the loop
    // is unrolled by two so that both are statically named), and slabs kt+2 .. kt+NS are in
    // flight from global memory.  One barrier per slab: it publishes slab kt+1 (every wave
    // has waited for ITS share of that slab's DMA) and retires ring slot kt % NS (every wave
    // has finished reading slab kt's fragments), which is then refilled with slab kt+NS.
    // The fragment reads are inline asm: a compiler-visible ds_read after an LDS-DMA makes
    // hipcc insert s_waitcnt vmcnt(0) in front of it (it cannot prove the DMA targets a
    // different ring slot), which would drain the whole prefetch pipeline every slab.  Their
    // completion is therefore waited for by hand (lgkmcnt(0) + register fence, cdna guide
    // 5.7) right before the MFMAs that consume them.
    f32x4 fa[KK][2][MT], fb[KK][2][NT];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    unsigned a_c[KK], w_c[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const int c = ((kq + 4 * kk) ^ sw) * 4;
        a_c[kk] = lds0 + (aoff + c) * 4;
        w_c[kk] = lds0 + (woff + c) * 4;
    }
    // (macros, not lambdas: hipcc rejects asm operands that are lambda captures)
#define AFTER_LOAD_FRAGS(p, slab)                                                                  \
    {                                                                                              \
        const unsigned so__ = (unsigned)((slab) % NS) * (STAGE * 4);                               \
        _Pragma("unroll") for (int kk = 0; kk < KK; ++kk) {                                        \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                         \
                asm volatile("ds_read_b128 %0, %1" : "=v"(fa[kk][p][i]) : "v"(a_c[kk] + so__ + i * 16 * BK * 4)); \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                         \
                asm volatile("ds_read_b128 %0, %1" : "=v"(fb[kk][p][j]) : "v"(w_c[kk] + so__ + j * 16 * BK * 4)); \
        }                                                                                          \
    }
#define AFTER_FENCE_FRAGS(p)                                                 \
    {                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   \
        _Pragma("unroll") for (int kk = 0; kk < KK; ++kk) {                  \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[kk][p][i])); \
            _Pragma("unroll") for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(fb[kk][p][j])); \
        }                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                   \
    }
#define AFTER_MFMA_STEP(A_, B_, comp)                                                           \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[i][comp], B_[j][comp], acc[i][j], 0, 0, 0);
#define AFTER_MMA(p)                                      \
    _Pragma("unroll") for (int kk = 0; kk < KK; ++kk) {   \
        AFTER_MFMA_STEP(fa[kk][p], fb[kk][p], 0)          \
        AFTER_MFMA_STEP(fa[kk][p], fb[kk][p], 1)          \
        AFTER_MFMA_STEP(fa[kk][p], fb[kk][p], 2)          \
        AFTER_MFMA_STEP(fa[kk][p], fb[kk][p], 3)          \
    }
    // wait until slab s_ has landed: slabs s_+1 .. min(last_, nk-1) may stay in flight
#define AFTER_WAIT_SLAB(s_, last_)                                                      \
    {                                                                                   \
        const int rem__ = ((last_) < nk - 1 ? (last_) : nk - 1) - (s_);                 \
        if (rem__ >= 3 && NS >= 4) wait_vmcnt<(NS >= 4 ? 3 : 0) * LPS>();              \
        else if (rem__ >= 2 && NS >= 3) wait_vmcnt<(NS >= 3 ? 2 : 0) * LPS>();         \
        else if (rem__ >= 1 && NS >= 2) wait_vmcnt<(NS >= 2 ? 1 : 0) * LPS>();         \
        else wait_vmcnt<0>();                                                           \
    }
    // Software pipeline: while the MFMAs of slab kt run, the fragments of slab kt+1 are
    // already being read from LDS into the other register set (sets 0 / 1 alternate, the
    // loop is unrolled by two so that both are statically named), and slabs kt+2 .. kt+NS
    // are in flight from global memory.  One barrier per slab: it publishes slab kt+1
    // (every wave has waited for ITS share of that slab's DMA) and retires ring slot
    // kt % NS (every wave has finished reading slab kt's fragments), which is then
    // refilled with slab kt+NS.
#define AFTER_STEP(pc, pn, kt_)                                                             \
    {                                                                                       \
        const int kt__ = (kt_);                                                             \
        unsigned long long p0__ = 0, p1__ = 0, p2__ = 0, p3__ = 0;                          \
        if (g.dbg) p0__ = __builtin_readcyclecounter();                                     \
        AFTER_FENCE_FRAGS(pc) /* slab kt's fragments are in registers, its LDS reads retired */ \
        if (g.dbg) p1__ = __builtin_readcyclecounter();                                     \
        if (kt__ + 1 < nk) {                                                                \
            if (kt__ + NS - 1 <= nk - 1) {                                                  \
                wait_vmcnt<(NS - 2) * LPS>(); /* steady state: kt+2 .. kt+NS-1 in flight */  \
            } else {                                                                        \
                AFTER_WAIT_SLAB(kt__ + 1, kt__ + NS - 1)                                    \
            }                                                                               \
            if (g.dbg) p2__ = __builtin_readcyclecounter();                                 \
            __builtin_amdgcn_s_barrier();                                                   \
            asm volatile("" ::: "memory");                                                  \
            if (g.dbg) p3__ = __builtin_readcyclecounter();                                 \
            if (kt__ + NS < nk) issue(kt__ + NS, kt__ % NS);                                \
            AFTER_LOAD_FRAGS(pn, kt__ + 1)                                                  \
        }                                                                                   \
        if (g.dbg && kt__ + 1 < nk) {                                                       \
            ph_fence += p1__ - p0__;                                                        \
            ph_vm += p2__ - p1__;                                                           \
            ph_bar += p3__ - p2__;                                                          \
        }                                                                                   \
        AFTER_MMA(pc)                                                                       \
    }
    unsigned long long ph_fence = 0, ph_vm = 0, ph_bar = 0;
    AFTER_WAIT_SLAB(0, NS - 1)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (g.dbg) t_loop = __builtin_readcyclecounter();
    AFTER_LOAD_FRAGS(0, 0)
    for (int kt = 0; kt < nk; kt += 2) {
        AFTER_STEP(0, 1, kt)
        if (kt + 1 < nk) AFTER_STEP(1, 0, kt + 1)
    }
#undef AFTER_STEP
#undef AFTER_WAIT_SLAB
#undef AFTER_MMA
#undef AFTER_MFMA_STEP
#undef AFTER_FENCE_FRAGS
#undef AFTER_LOAD_FRAGS
    if (g.dbg) {
        asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
        t_end = __builtin_readcyclecounter();
    }

    const int ccol = lane & 15, crow0 = 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int gn = n0 + wn0 + j * 16 + ccol;
        if (gn >= N) continue;
        const float bv = g.bias ? g.bias[gn] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm0 + i * 16 + crow0 + r;
                if (gm >= M) continue;
                float v = acc[i][j][r] + bv;
                if (g.epilogue == EPI_GELU) v = gelu_erf(v);
                if (g.epilogue == EPI_RESIDUAL) v += g.R[(size_t)gm * g.ldr + gn];
                if (g.epilogue == EPI_RELU) v = fmaxf(v, 0.f);
                if (g.epilogue == EPI_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                g.C[(size_t)gm * g.ldc + gn] = v;
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
        d[7] = (ph_fence & 0xFFFFF) | ((ph_vm & 0xFFFFF) << 20) | ((ph_bar & 0xFFFFF) << 40);
    }
}

// ---------------------------------------------------------------------------------
// Balanced split-K variant for the few-token (B = 1..4) shapes.  With M = 768 tokens the
// 768 x 1536 output is 4.5 16x16 blocks per SIMD and the 768 x 512 one 1.5: whatever the
// tile, whole blocks quantise to 5 / 2 rounds (10 % / 25 % of the matrix pipe idle) and the
// finer the tile, the more single-accumulator waves whose dependent MFMAs cannot overlap.
// Here each workgroup owns a (16 MB) x (32 NB) output tile and BOTH halves of K: wave w
// accumulates k-half (w & 1) of column part (w >> 1), i.e. MB x NB blocks over K/2 -- with
// (MB, NB) = (3, 3) / (3, 1) that is exactly 9 / 3 half-blocks per SIMD and ONE workgroup
// per CU at M = 768 (256 workgroups; 256 B at batch B), every wave carrying MB*NB
// independent accumulators (72 / 24 MFMAs per 32-deep slab between barriers).  The two
// k-halves are summed through LDS in the epilogue: each wave finalises half of the blocks as
// (own + partner), a commutative two-term sum -> bit-deterministic, no atomics.
// Ring slot = [A k-half 0 | A k-half 1 | W k-half 0 | W k-half 1] rows of 32 floats, filled
// by LDS-DMA with the same source-side XOR swizzle as above.  Requires K % 64 == 0.
// MODE 1 (EPI_CFG_EULER, MB = 3): the tile's three row blocks are the three CFG branches of the
// same 16 tokens (tile row 16 r + tt <-> A row r * BT + 16 tm + tt), so after the k-part reduction a
// lane holds d_full, d_mid, d_none of one token and four channels and the CFG combination, the
// Euler update and both output layouts are finished in registers.
template <int MB, int NB, int KS, int NS, int RS, int MODE = 0>
__global__ __launch_bounds__(128 * KS * RS) void gemm_f32_bal_kernel(GemmArgs g, int tiles_m, int tiles_n, int xcd_pm) {
    static_assert(MODE == 0 || (MB == 3 && RS == 1), "MODE 1: three row blocks = the three CFG branches");
    constexpr int BK = 32, CPR = 8, RPP = 8, KK = 2;
    static_assert(MB % RS == 0, "row parts must divide the tile's block rows");
    constexpr int MT = MB / RS, NT = NB;          // blocks per wave (names used by the macros)
    constexpr int BM = 16 * MB, BN = 32 * NB;
    constexpr int NW = 2 * KS * RS;               // waves: KS k-parts x RS row parts x 2 column parts
    constexpr int ROWS = KS * (BM + BN);          // ring slot: [A k-part 0..KS-1 | W k-part 0..KS-1]
    constexpr int LPS = ROWS / RPP / NW;          // DMA instructions per wave per slab
    static_assert(ROWS % (RPP * NW) == 0, "ring slot must split evenly over the waves");
    constexpr int STAGE = ROWS * BK;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // Workgroup -> tile map.  Block b runs on XCD b % 8 (round-robin dispatch) and every XCD has
    // a private L2, so the 8 XCDs are laid out as a pm x pn grid over the tile matrix: an XCD
    // then fetches 1/pm of A and 1/pn of W (launch_bal picks pm minimising A/pm + W/pn; a 1 x 8
    // grid makes every XCD re-read all of A: 44 MB per MLP-down launch against 9.4 MB
    // algorithmic).  Needs tiles_m % pm == 0 and tiles_n % pn == 0, else pm arrives as 0 and
    // XCDs take contiguous ranges of the tm-fastest order.
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
    const int kh = wid % KS, rp = (wid / KS) % RS, part = wid / (KS * RS);  // k-part, row part, column part
    const int M = g.M, N = g.N, Kh = g.K / KS;

    // DMA addressing: wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset, so that
    // a piece costs one s_mov m0 + one global_load_lds (no per-slab VALU address arithmetic).
    const int rsub = lane / CPR, pos = lane % CPR;
    unsigned voff[LPS];
    const float* sbase[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const int row0 = (wid * LPS + i) * RPP;  // first row of the piece: wave uniform
        const int row = row0 + rsub;
        if (row0 < KS * BM) {
            const int half = row0 / BM;
            int gm = min(m0 + row - half * BM, M - 1);
            if constexpr (MODE == 1) {
                const int rin = row - half * BM, bt = M / 3;  // branch rin / 16, token 16 tm + rin % 16
                gm = (rin >> 4) * bt + min(tm * 16 + (rin & 15), bt - 1);
            }
            sbase[i] = g.A + half * Kh;
            voff[i] = ((unsigned)gm * (unsigned)g.lda + (unsigned)((pos ^ (row & (CPR - 1))) * 4)) * 4u;
        } else {
            const int rr0 = row0 - KS * BM;
            const int half = rr0 / BN;
            const int gn = min(n0 + (row - KS * BM) - half * BN, N - 1);
            sbase[i] = g.W + half * Kh;
            voff[i] = ((unsigned)gn * (unsigned)g.ldw + (unsigned)((pos ^ (row & (CPR - 1))) * 4)) * 4u;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
#define AFTER_BAL_DMA(w_, slab_, slot_)                                                                         \
    lds_dma16(lds0 + (unsigned)(((slot_) * STAGE + (wid * LPS + (w_)) * RPP * BK) * 4), voff[w_],                 \
              (unsigned long long)(uintptr_t)(sbase[w_] + (slab_) * BK)); /* (M0 is the compiler's: gemm_pipe.h) */

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = Kh / BK;
    unsigned long long t_start = 0, t_loop = 0, t_end = 0, r_start = 0;
    unsigned long long ph_fence = 0, ph_vm = 0, ph_bar = 0;
    if (g.dbg) {
        t_start = __builtin_readcyclecounter();
        r_start = wall_clock64();
    }
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
    AFTER_GEMM_WAIT_SLAB(0, NS - 1)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (g.dbg) t_loop = __builtin_readcyclecounter();
    AFTER_GEMM_LOAD_FRAGS(0, 0)
    int kt = 0;
    for (; kt + 1 + NS < nk; kt += 2) {  // steady state: no tail predicates in the MFMA stream
        AFTER_GEMM_STEP_IL(0, 1, kt, true)
        AFTER_GEMM_STEP_IL(1, 0, kt + 1, true)
    }
    for (; kt < nk; kt += 2) {
        AFTER_GEMM_STEP_IL(0, 1, kt, false)
        if (kt + 1 < nk) AFTER_GEMM_STEP_IL(1, 0, kt + 1, false)
    }
    if (g.dbg) {
        asm volatile("s_nop 0" ::"v"(acc[MT - 1][NT - 1][0]));
        t_end = __builtin_readcyclecounter();
    }

    // ---- split-K reduction through LDS (the ring is free: every wave is past its last read).
    // Wave (kh, rp, part) finalises the blocks b with b % KS == kh as p0 + p1 + ... in k-part
    // order: a fixed summation order, i.e. bit-deterministic.  KS = 1: nothing to reduce.
    float* red = smem;  // [wave][block][lane][4]
    if constexpr (KS > 1) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                *reinterpret_cast<f32x4*>(red + ((wid * MT * NT + i * NT + j) * 64 + lane) * 4) = acc[i][j];
        __syncthreads();
    }
    unsigned long long t_red = 0, t_st = 0;
    if (g.dbg) t_red = __builtin_readcyclecounter();
    if constexpr (MODE == 1) {
        const int bt = M / 3, T = g.T;
        const float total = g.cfg[0], factor = g.cfg[1], dt = g.cfg[2];
        const int tok = tm * 16 + (lane & 15);
        const int bq = tok / T, t = tok - bq * T;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j % KS != kh) continue;  // wave (kh, part) finishes column blocks j = kh (mod KS), all branches
            const int gn = n0 + part * 16 * NB + j * 16 + 4 * (lane >> 4);
            f32x4 d[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int w0 = part * KS;
                d[i] = *reinterpret_cast<const f32x4*>(red + ((w0 * MT * NT + i * NT + j) * 64 + lane) * 4);
#pragma unroll
                for (int q = 1; q < KS; ++q)
                    d[i] += *reinterpret_cast<const f32x4*>(red + (((w0 + q) * MT * NT + i * NT + j) * 64 + lane) * 4);
            }
            if (tok >= bt) continue;
            f32x4 xn = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = gn + r;
                if (n >= N) continue;
                const float bo = g.bias ? g.bias[n] : 0.f;
                const float dfull = d[0][r] + bo, dmid = d[1][r] + bo, dnone = d[2][r] + bo;
                const float v = dnone + total * (dmid + factor * (dfull - dmid) - dnone);
                const size_t o = ((size_t)bq * N + n) * T + t;
                xn[r] = (g.xin ? g.xin[o] : 0.f) + v * dt;
                g.xout[o] = xn[r];
            }
            if (g.xt) {
                float* xp = g.xt + (size_t)tok * g.xt_ld + gn;
                if (gn + 3 < g.xt_ld) *reinterpret_cast<f32x4*>(xp) = xn;  // columns >= N carry zeros (padding)
                else
                    for (int r = 0; r < 4; ++r)
                        if (gn + r < g.xt_ld) xp[r] = xn[r];
            }
        }
        return;
    }
    // accumulator layout (transposed MFMA): lane l holds C[row = l & 15][col = 4 (l >> 4) + r]
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
                const int w0 = (part * RS + rp) * KS;  // first wave of this (row part, column part)
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
                for (int r = 0; r < 4; ++r) o[r] = gelu_erf(o[r]);
            } else if (g.epilogue == EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
            } else if (g.epilogue == EPI_SIGMOID) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = 1.0f / (1.0f + expf(-o[r]));
            }
            float* cp = g.C + (size_t)gm * g.ldc + gn;
            if (vec_ok && gn + 3 < N) {
                if (g.epilogue == EPI_RESIDUAL)
                    o += *reinterpret_cast<const f32x4*>(g.R + (size_t)gm * g.ldr + gn);
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
    if (g.dbg) t_st = __builtin_readcyclecounter();
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
        d[7] = (ph_fence & 0xFFFFF) | ((ph_vm & 0xFFFFF) << 20) | ((ph_bar & 0xFFFFF) << 40);
        if (g.epilogue & 0x100) d[7] = ((t_red - t_end) & 0xFFFFF) | (((t_st - t_red) & 0xFFFFF) << 20);
    }
}

#undef AFTER_BAL_DMA

template <int MB, int NB, int KS, int NS, int RS = 1>
int launch_bal(const GemmArgs& g, hipStream_t stream) {
    constexpr int BM = 16 * MB, BN = 32 * NB;
    const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
    const size_t ring = size_t(NS) * KS * (BM + BN) * 32 * sizeof(float);
    const size_t red = KS > 1 ? size_t(2 * KS) * MB * NB * 256 * sizeof(float) : 0;
    const size_t lds = ring > red ? ring : red;
    static_assert(size_t(NS) * KS * (BM + BN) * 32 * sizeof(float) <= 160 * 1024, "ring exceeds the LDS");
    static LdsAttr attr_set;
    AFTER_TRY(ensure_lds_attr(attr_set, reinterpret_cast<const void*>(gemm_f32_bal_kernel<MB, NB, KS, NS, RS>), lds));
    // XCD grid pm x (8 / pm): minimise the per-XCD operand footprint A / pm + W / pn
    static int xcd2d = -1;
    if (xcd2d < 0) {
        const char* e = getenv("AFTER_GEMM_XCD2D");
        xcd2d = e ? atoi(e) : 1;
    }
    int pm = 0;
    if (xcd2d) {
        double best = 0;
        for (int c = 1; c <= 8; c *= 2) {
            if (tiles_m % c || tiles_n % (8 / c)) continue;
            const double cost = (double)g.M / c + (double)g.N / (8 / c);  // x K x 4 bytes
            if (pm == 0 || cost < best) {
                pm = c;
                best = cost;
            }
        }
    }
    hipLaunchKernelGGL((gemm_f32_bal_kernel<MB, NB, KS, NS, RS>), dim3(tiles_m * tiles_n), dim3(128 * KS * RS),
                       lds, stream, g, tiles_m, tiles_n, pm);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

// ---------------------------------------------------------------------------------
// Skinny GEMM for the streaming path (M <= 96 tokens: 3 CFG rows x 4 frames x 1..8 streams).
// Such a launch is pure weight streaming (3 MB of W against <= 0.6 MB of A), so the tile is
// turned around: one workgroup owns 16 output COLUMNS (16 rows of W) for all M rows, its 8 waves
// split K eight ways and load their fragments straight from global memory into registers (no
// LDS staging: every W element is used by exactly one wave), several k-blocks in flight per
// wave; the eight partial tiles are summed through LDS in wave order (deterministic).
// N / 16 workgroups pull W at ~100 KB per CU, i.e. one memory round trip.
template <int MB>
__global__ __launch_bounds__(512) void gemm_f32_skinny_kernel(GemmArgs g) {
    constexpr int U = MB <= 3 ? 4 : 2;  // k-blocks (16 deep) in flight per wave
    __shared__ __attribute__((aligned(16))) float red[8 * MB * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int M = g.M, N = g.N, Kw = g.K >> 3;
    const float* wp = g.W + (size_t)min(n0 + row, N - 1) * g.ldw + w * Kw + kq * 4;
    const float* ap[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) ap[i] = g.A + (size_t)min(i * 16 + row, M - 1) * g.lda + w * Kw + kq * 4;
    f32x4 acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < Kw; kb += 16 * U) {
        f32x4 bw[U], a[U][MB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = kb + 16 * u;
            if (k < Kw) {
                bw[u] = *reinterpret_cast<const f32x4*>(wp + k);
#pragma unroll
                for (int i = 0; i < MB; ++i) a[u][i] = *reinterpret_cast<const f32x4*>(ap[i] + k);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kb + 16 * u < Kw) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < MB; ++i)  // W fragment as srcA: the accumulator holds C^T
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u][c], a[u][i][c], acc[i], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) *reinterpret_cast<f32x4*>(red + ((w * MB + i) * 64 + lane) * 4) = acc[i];
    __syncthreads();
    // wave w finalises row blocks w, w + 8, ...: lane owns row m = 16 i + (lane & 15), columns
    // n0 + 4 (lane >> 4) + r
    const int gn = n0 + 4 * kq;
    for (int i = w; i < MB; i += 8) {
        f32x4 o = *reinterpret_cast<const f32x4*>(red + (i * 64 + lane) * 4);
#pragma unroll
        for (int q = 1; q < 8; ++q) o += *reinterpret_cast<const f32x4*>(red + ((q * MB + i) * 64 + lane) * 4);
        const int gm = i * 16 + row;
        if (gm >= M) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (gn + r >= N) continue;
            float v = o[r] + (g.bias ? g.bias[gn + r] : 0.f);
            if (g.epilogue == EPI_GELU) v = gelu_erf(v);
            if (g.epilogue == EPI_RESIDUAL) v += g.R[(size_t)gm * g.ldr + gn + r];
            if (g.epilogue == EPI_RELU) v = fmaxf(v, 0.f);
            if (g.epilogue == EPI_SIGMOID) v = 1.0f / (1.0f + expf(-v));
            g.C[(size_t)gm * g.ldc + gn + r] = v;
        }
    }
}

template <int MB>
int launch_skinny(const GemmArgs& g, hipStream_t stream) {
    hipLaunchKernelGGL((gemm_f32_skinny_kernel<MB>), dim3(cdiv(g.N, 16)), dim3(512), 0, stream, g);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

template <int MT, int NT, int NS, int BK>
int launch_dma(const GemmArgs& g, hipStream_t stream) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
    const size_t lds = size_t(NS) * (BM + BN) * BK * sizeof(float);
    static LdsAttr attr_set;
    AFTER_TRY(ensure_lds_attr(attr_set, reinterpret_cast<const void*>(gemm_f32_dma_kernel<MT, NT, NS, BK>), lds));
    hipLaunchKernelGGL((gemm_f32_dma_kernel<MT, NT, NS, BK>), dim3(tiles_m * tiles_n), dim3(256), lds,
                       stream, g, tiles_m, tiles_n);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

int g_lds_min = -1, g_bk = -1;  // debug knobs: AFTER_GEMM_LDS_MIN (bytes), AFTER_GEMM_BK (32|64)

template <int MT, int NT, int BK>
int launch_cfg_bk(const GemmArgs& g, hipStream_t stream) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
    size_t lds = size_t(2) * (BM + BN) * lds_ld<BK>() * sizeof(float);
    if (g_lds_min < 0) {
        const char* e = getenv("AFTER_GEMM_LDS_MIN");
        g_lds_min = e ? atoi(e) : 0;
    }
    if ((size_t)g_lds_min > lds) lds = g_lds_min;
    static LdsAttr attr;
    AFTER_TRY(ensure_lds_attr(attr, reinterpret_cast<const void*>(gemm_f32_kernel<MT, NT, BK>), lds));
    hipLaunchKernelGGL((gemm_f32_kernel<MT, NT, BK>), dim3(tiles_m * tiles_n), dim3(256), lds, stream, g,
                       tiles_m, tiles_n);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

int g_dma = -1;  // AFTER_GEMM_DMA=0 disables the LDS-DMA path (debug)

template <int MT, int NT>
int launch_cfg(const GemmArgs& g, hipStream_t stream) {
    if (g_dma < 0) {
        const char* e = getenv("AFTER_GEMM_DMA");
        g_dma = e ? atoi(e) : 1;
    }
    if (g_bk < 0) {
        const char* e = getenv("AFTER_GEMM_BK");
        g_bk = e ? atoi(e) : 32;
    }
    if (g_dma && (g.K % 32) == 0) {
        // measured on MI355X (scripts/bench_gemm.py): a 2-deep ring is as fast as deeper
        // ones (co-resident workgroups already cover the L2 latency) and leaves LDS for more
        // workgroups per CU; 64-deep slabs pay off for the long-K, few-workgroup MLP-down GEMM
        if constexpr (MT * NT <= 2) {
            if ((g_bk == 64 || (g_bk == 32 && g.K >= 1024)) && (g.K % 64) == 0) {
                if (g_dma == 3) return launch_dma<MT, NT, 3, 64>(g, stream);
                return launch_dma<MT, NT, 2, 64>(g, stream);
            }
        }
        if (g_dma == 3) return launch_dma<MT, NT, 3, 32>(g, stream);
        if (g_dma == 4 && MT * NT <= 4) return launch_dma<MT, NT, 4, 32>(g, stream);
        return launch_dma<MT, NT, 2, 32>(g, stream);
    }
    if (g_bk == 64 && MT * NT <= 4) return launch_cfg_bk<MT, NT, 64>(g, stream);
    return launch_cfg_bk<MT, NT, 32>(g, stream);
}

}  // namespace

int launch_gemm(const GemmArgs& g, hipStream_t stream) { return launch_gemm_cfg(g, 0, 0, stream); }

int launch_gemm_cfg_euler(const GemmArgs& g, hipStream_t stream) {
    constexpr int MB = 3, NB = 1, KS = 4, NS = 2;  // 16 tokens x 32 channels per workgroup, 4 k-parts (8 waves)
    AFTER_REQUIRE(g.M > 0 && g.M % 3 == 0 && g.N > 0 && g.T > 0 && (g.M / 3) % g.T == 0, AFTER_E_INVALID,
                  "cfg_euler gemm: M = 3 * B * T expected (M=%d T=%d)", g.M, g.T);
    AFTER_REQUIRE((g.K % (32 * KS)) == 0 && (g.lda % 4) == 0 && (g.ldw % 4) == 0 && g.xout && g.cfg,
                  AFTER_E_INVALID, "cfg_euler gemm: K %% %d != 0 or missing operands", 32 * KS);
    AFTER_REQUIRE((size_t)g.M * g.lda < (1u << 30) && (size_t)g.N * g.ldw < (1u << 30), AFTER_E_INVALID,
                  "cfg_euler gemm: operand too large for 32-bit DMA offsets");
    AFTER_REQUIRE(!g.xt || (g.xt_ld >= g.N && (g.xt_ld % 4) == 0), AFTER_E_INVALID, "cfg_euler gemm: bad xt_ld");
    const int bt = g.M / 3;
    const int tiles_m = cdiv(bt, 16), tiles_n = cdiv(g.N, 32 * NB);
    const size_t ring = size_t(NS) * KS * (16 * MB + 32 * NB) * 32 * sizeof(float);
    const size_t red = size_t(2 * KS) * MB * NB * 256 * sizeof(float);
    const size_t lds = ring > red ? ring : red;
    static LdsAttr attr_set;
    AFTER_TRY(ensure_lds_attr(attr_set, reinterpret_cast<const void*>(gemm_f32_bal_kernel<MB, NB, KS, NS, 1, 1>), lds));
    hipLaunchKernelGGL((gemm_f32_bal_kernel<MB, NB, KS, NS, 1, 1>), dim3(tiles_m * tiles_n), dim3(128 * KS), lds,
                       stream, g, tiles_m, tiles_n, 0);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

int launch_gemm_cfg(const GemmArgs& g, int mt, int nt, hipStream_t stream) {
    AFTER_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, AFTER_E_INVALID, "gemm: empty problem %dx%dx%d",
                  g.M, g.N, g.K);
    AFTER_REQUIRE((g.K % 4) == 0 && (g.lda % 4) == 0 && (g.ldw % 4) == 0, AFTER_E_INVALID,
                  "gemm: K/lda/ldw must be multiples of 4 (K=%d lda=%d ldw=%d)", g.K, g.lda, g.ldw);
    AFTER_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.W % 16) == 0, AFTER_E_INVALID,
                  "gemm: operands must be 16-byte aligned");
    AFTER_REQUIRE(g.epilogue != EPI_RESIDUAL || g.R != nullptr, AFTER_E_INVALID,
                  "gemm: residual epilogue without R");
    if (mt >= 400) {  // skinny kernels: mt = 400 + MB
        AFTER_REQUIRE((g.K % 128) == 0 && g.M <= 16 * (mt - 400), AFTER_E_INVALID,
                      "gemm: skinny tiles need K %% 128 == 0 and M <= 16 MB");
        if (mt == 401) return launch_skinny<1>(g, stream);
        if (mt == 403) return launch_skinny<3>(g, stream);
        if (mt == 406) return launch_skinny<6>(g, stream);
        set_error("gemm: no skinny configuration MB=%d", mt - 400);
        return AFTER_E_INVALID;
    }
    if (mt >= 100) {  // balanced split-K kernels: mt = 100 + MB (2 k-parts) / 200 + MB (4), nt = 10 NS + NB
        if (mt >= 300) {  // no K split: 2 row parts x 2 column parts (mt = 300 + MB)
            const int mb3 = mt % 100, nb3 = nt % 10, ns3 = nt / 10;
            AFTER_REQUIRE((g.K % 32) == 0, AFTER_E_INVALID, "gemm: K %% 32 != 0");
            AFTER_REQUIRE((size_t)g.M * g.lda < (1u << 30) && (size_t)g.N * g.ldw < (1u << 30), AFTER_E_INVALID,
                          "gemm: operand too large for 32-bit DMA offsets");
            if (mb3 == 4 && nb3 == 3 && ns3 == 2) return launch_bal<4, 3, 1, 2, 2>(g, stream);
            if (mb3 == 4 && nb3 == 3 && ns3 == 3) return launch_bal<4, 3, 1, 3, 2>(g, stream);
            if (mb3 == 4 && nb3 == 2 && ns3 == 2) return launch_bal<4, 2, 1, 2, 2>(g, stream);
            if (mb3 == 4 && nb3 == 2 && ns3 == 3) return launch_bal<4, 2, 1, 3, 2>(g, stream);
            if (mb3 == 6 && nb3 == 3 && ns3 == 2) return launch_bal<6, 3, 1, 2, 2>(g, stream);
            if (mb3 == 6 && nb3 == 2 && ns3 == 2) return launch_bal<6, 2, 1, 2, 2>(g, stream);
            set_error("gemm: no row-split configuration MB=%d NB=%d NS=%d", mb3, nb3, ns3);
            return AFTER_E_INVALID;
        }
        const int ks = mt >= 200 ? 4 : 2, mb = mt % 100, nb = nt % 10, ns = nt / 10;
        AFTER_REQUIRE((g.K % (32 * ks)) == 0, AFTER_E_INVALID, "gemm: %d-way split-K needs K %% %d == 0", ks, 32 * ks);
        AFTER_REQUIRE((size_t)g.M * g.lda < (1u << 30) && (size_t)g.N * g.ldw < (1u << 30), AFTER_E_INVALID,
                      "gemm: operand too large for 32-bit DMA offsets");
#define AFTER_BAL_CASE(MB_, NB_, KS_, NS_) \
    if (mb == MB_ && nb == NB_ && ks == KS_ && ns == NS_) return launch_bal<MB_, NB_, KS_, NS_>(g, stream);
        AFTER_BAL_CASE(3, 3, 2, 2) AFTER_BAL_CASE(3, 3, 2, 3) AFTER_BAL_CASE(3, 1, 2, 2) AFTER_BAL_CASE(3, 1, 2, 4)
        AFTER_BAL_CASE(3, 3, 4, 2) AFTER_BAL_CASE(3, 1, 4, 2) AFTER_BAL_CASE(3, 1, 4, 3) AFTER_BAL_CASE(3, 1, 4, 4)
        AFTER_BAL_CASE(3, 2, 4, 2) AFTER_BAL_CASE(2, 2, 4, 2) AFTER_BAL_CASE(2, 2, 4, 3)
        AFTER_BAL_CASE(2, 2, 2, 2) AFTER_BAL_CASE(3, 2, 2, 2) AFTER_BAL_CASE(2, 3, 2, 2) AFTER_BAL_CASE(2, 2, 2, 3)
        AFTER_BAL_CASE(4, 2, 2, 2) AFTER_BAL_CASE(4, 3, 2, 2)
#undef AFTER_BAL_CASE
        set_error("gemm: no split-K configuration MB=%d NB=%d KS=%d NS=%d", mb, nb, ks, ns);
        return AFTER_E_INVALID;
    }
    if (mt > 0 || nt > 0) {
        if (mt == 4 && nt == 4) return launch_cfg<4, 4>(g, stream);
        if (mt == 4 && nt == 2) return launch_cfg<4, 2>(g, stream);
        if (mt == 2 && nt == 2) return launch_cfg<2, 2>(g, stream);
        if (mt == 1 && nt == 2) return launch_cfg<1, 2>(g, stream);
        if (mt == 1 && nt == 1) return launch_cfg<1, 1>(g, stream);
        set_error("gemm: no tile configuration %dx%d", mt, nt);
        return AFTER_E_INVALID;
    }
    // Tile choice (measured on MI355X, scripts/bench_gemm.py).  The balanced
    // split-K kernels win whenever K allows the split: they quantise to whole CUs (48-row tiles:
    // 768 B tokens -> 256 B workgroups per 96 / 32 columns), keep >= 2 waves per SIMD busy and
    // have no ragged last round.  A wave cannot overlap its own non-MFMA instructions with its
    // MFMAs (scripts/ubench/mfma_coissue.hip), so a second wave per SIMD is what hides the
    // DMA / LDS / barrier work of the first.
    static int use_bal = -1;
    if (use_bal < 0) {
        const char* e = getenv("AFTER_GEMM_BAL");
        use_bal = e ? atoi(e) : 1;
    }
    static int use_skinny = -1;
    if (use_skinny < 0) {
        const char* e = getenv("AFTER_GEMM_SKINNY");
        use_skinny = e ? atoi(e) : 1;
    }
    if (use_skinny && g.M <= (use_skinny > 1 ? 96 : 48) && (g.K % 128) == 0 && g.N >= 64) {
        if (g.M <= 16) return launch_skinny<1>(g, stream);
        if (g.M <= 48) return launch_skinny<3>(g, stream);
        return launch_skinny<6>(g, stream);  // AFTER_GEMM_SKINNY=2 only: slower than split-K at M = 96
    }
    const bool fits32 = (size_t)g.M * g.lda < (1u << 30) && (size_t)g.N * g.ldw < (1u << 30);
    if (use_bal && fits32 && g.M >= 64 && g.N >= 128) {
        const bool long_k = g.K >= 2 * g.N;  // "down" projections: narrow N, long K
        if ((long_k || g.M < 192) && (g.K % 128) == 0 && g.M <= 1024) return launch_bal<3, 1, 4, 2>(g, stream);
        if ((g.K % 64) == 0) {
            if (g.M <= 1024) return launch_bal<3, 1, 2, 2>(g, stream);
            if (long_k)  // K is long enough to keep splitting it
                return g.M < 4096 ? launch_bal<3, 1, 2, 2>(g, stream) : launch_bal<4, 2, 2, 2>(g, stream);
            // wide-N, short-K GEMMs with many tokens: enough tiles to balance without a K split,
            // so the waves split the tile's rows instead (no reduction, half the ring per
            // workgroup -> more co-resident workgroups).  M = 6144: 129 TFLOP/s vs 115 split.
            return g.M < 2048 ? launch_bal<6, 3, 1, 2, 2>(g, stream) : launch_bal<4, 3, 1, 2, 2>(g, stream);
        }
    }
    // Classic 2x2-wave tiles: the largest workgroup tile that still yields >= 2 workgroups per CU
    // (co-residency is what hides the LDS/barrier latency of this MFMA-bound loop).
    auto wgs = [&](int bm, int bn) { return (long long)cdiv(g.M, bm) * cdiv(g.N, bn); };
    const long long want = 2 * 256;
    // (measured, M = 6144: 64x64 tiles reach 93-105 TFLOP/s, 128x64 87, 128x128 69-82:
    // the larger tiles hold too few waves per SIMD to cover their own LDS latency)
    if (wgs(64, 64) >= want) return launch_cfg<2, 2>(g, stream);
    // below ~3 workgroups per CU the finer 32x32 tile balances the 1024 SIMDs better
    if (wgs(32, 64) >= 3 * 256) return launch_cfg<1, 2>(g, stream);
    return launch_cfg<1, 1>(g, stream);
}

}  // namespace after

// Diagnostic / unit-test entry point (not on the reference's surface): the GEMM used by
// every Linear of the denoiser, callable on its own for parity and roofline tests.
// force_mt/force_nt > 0 pin the tile configuration (workgroup tile 32*mt x 32*nt).
static unsigned long long* g_gemm_dbg = nullptr;
extern "C" void after_gemm_set_debug(unsigned long long* dbg) { g_gemm_dbg = dbg; }

extern "C" int after_gemm_f32(const float* A, int lda, const float* W, int ldw, const float* bias,
                              const float* R, int ldr, float* C, int ldc, int M, int N, int K,
                              int epilogue, int force_mt, int force_nt, void* stream) {
    after::GemmArgs g{A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, epilogue};
    g.dbg = g_gemm_dbg;
    return after::launch_gemm_cfg(g, force_mt, force_nt, (hipStream_t)stream);
}
