// fp32 GEMM on the CDNA4 matrix cores: C[M,N] = epi(A[M,K] * W[N,K]^T + bias).
//
// Used for every Linear of the denoiser (reference transformerv2.py:251,275-283,
// 330-334,387-398,430,488-492 -- all `nn.Linear`, i.e. x @ W^T + b with W stored
// [out, in]) so activations and weights are both K-contiguous ("B^T input").
//
// Design (gfx950): 256-thread workgroup = 4 waves in a 2x2 arrangement, each wave
// owning TM x TN tiles of 32x32 accumulated with v_mfma_f32_32x32x2_f32 (exact
// fp32: the reference computes in fp32).  K is walked in BK=32
// slabs staged through LDS: coalesced 16-byte global loads (8 lanes cover one
// 128-byte row segment) -> registers -> ds_write_b128 into rows padded to 36
// floats (144 B) so that the fragment reads -- one ds_read_b128 per lane giving
// four consecutive k for one row, i.e. operands of four MFMAs -- are bank-conflict
// free for every 16-lane service group.  The next slab's global loads are issued
// before the current slab's MFMAs (register double buffering + two LDS buffers,
// one barrier per slab).  Workgroup ids are remapped so that each XCD (private
// 4 MiB L2) owns a contiguous range of N-tiles, i.e. streams only 1/8 of W.
#include "common.h"

namespace after {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = 36;  // floats per LDS row (32 + 4 pad)

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

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
template <int NL>
__device__ __forceinline__ void tile_sstore(const float4 (&r)[NL], float* dst) {
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<float4*>(dst + 32 * i * LDS_LD) = r[i];
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int WM = BM / 2, WN = BN / 2;  // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_LOADS = BM / 32, W_LOADS = BN / 32;  // float4 per thread per slab
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM][LDS_LD]
    float* Ws = smem + 2 * BM * LDS_LD;     // [2][BN][LDS_LD]

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
    const int wm0 = (wid >> 1) * WM, wn0 = (wid & 1) * WN;
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;

    const float* __restrict__ gA = g.A;
    const float* __restrict__ gW = g.W;
    const int M = g.M, N = g.N, K = g.K, lda = g.lda, ldw = g.ldw;

    size_t Aoff[A_LOADS], Woff[W_LOADS];
    bool Aok[A_LOADS], Wok[W_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int gm = m0 + lrow + 32 * i;
        Aok[i] = gm < M;
        Aoff[i] = (size_t)(Aok[i] ? gm : 0) * lda + lc4;
    }
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) {
        const int gn = n0 + lrow + 32 * i;
        Wok[i] = gn < N;
        Woff[i] = (size_t)(Wok[i] ? gn : 0) * ldw + lc4;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[A_LOADS], rw[W_LOADS];
    const int nk = (K + BK - 1) / BK;
    tile_gload<A_LOADS>(ra, gA, Aoff, Aok, 0, lc4 < K);
    tile_gload<W_LOADS>(rw, gW, Woff, Wok, 0, lc4 < K);
    tile_sstore<A_LOADS>(ra, As + lrow * LDS_LD + lc4);
    tile_sstore<W_LOADS>(rw, Ws + lrow * LDS_LD + lc4);
    __syncthreads();
    const int frow = lane & 31, fk = (lane >> 5) * 4;
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
        for (int kk = 0; kk < BK / 8; ++kk) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const float4*>(Wb + j * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) {
            tile_sstore<A_LOADS>(ra, As + ((cur ^ 1) * BM + lrow) * LDS_LD + lc4);
            tile_sstore<W_LOADS>(rw, Ws + ((cur ^ 1) * BN + lrow) * LDS_LD + lc4);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int ccol = lane & 31, crow0 = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gn = n0 + wn0 + j * 32 + ccol;
        if (gn >= N) continue;
        const float bv = g.bias ? g.bias[gn] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + crow0;
                if (gm >= M) continue;
                float v = acc[i][j][r] + bv;
                if (g.epilogue == EPI_GELU) v = gelu_erf(v);
                if (g.epilogue == EPI_RESIDUAL) v += g.R[(size_t)gm * g.ldr + gn];
                g.C[(size_t)gm * g.ldc + gn] = v;
            }
        }
    }
}

template <int BM, int BN>
int launch_cfg(const GemmArgs& g, hipStream_t stream) {
    const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
    const size_t lds = size_t(2) * (BM + BN) * LDS_LD * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        AFTER_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<BM, BN>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN>), dim3(tiles_m * tiles_n), dim3(256), lds, stream, g,
                       tiles_m, tiles_n);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

}  // namespace

int launch_gemm(const GemmArgs& g, hipStream_t stream) {
    AFTER_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, AFTER_E_INVALID, "gemm: empty problem %dx%dx%d",
                  g.M, g.N, g.K);
    AFTER_REQUIRE((g.K % 4) == 0 && (g.lda % 4) == 0 && (g.ldw % 4) == 0, AFTER_E_INVALID,
                  "gemm: K/lda/ldw must be multiples of 4 (K=%d lda=%d ldw=%d)", g.K, g.lda, g.ldw);
    AFTER_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.W % 16) == 0, AFTER_E_INVALID,
                  "gemm: operands must be 16-byte aligned");
    const long long t128 = (long long)cdiv(g.M, 128) * cdiv(g.N, 128);
    const long long t12864 = (long long)cdiv(g.M, 128) * cdiv(g.N, 64);
    if (t128 >= 512) return launch_cfg<128, 128>(g, stream);
    if (t12864 >= 384) return launch_cfg<128, 64>(g, stream);
    return launch_cfg<64, 64>(g, stream);
}

}  // namespace after
