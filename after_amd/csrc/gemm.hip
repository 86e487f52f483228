// fp32 GEMM on the CDNA4 matrix cores: C[M,N] = epi(A[M,K] * W[N,K]^T + bias).
//
// Used for every Linear of the denoiser (reference transformerv2.py:251,275-283,
// 330-334,387-398,430,488-492 -- all `nn.Linear`, i.e. x @ W^T + b with W stored
// [out, in]) so activations and weights are both K-contiguous ("B^T input").
//
// Design (gfx950): 256-thread workgroup = 4 waves in a 2x2 arrangement, each wave
// owning MT x NT tiles of 16x16 accumulated with v_mfma_f32_16x16x4_f32 (exact
// fp32: the reference computes in fp32).  The 16x16 granule (rather
// than 32x32) is what lets the B=1 shapes (M = 768 tokens) fill 256 CUs with >= 2
// co-resident workgroups each: the fp32 matrix pipe is slow (32 cycles per
// instruction), so LDS/L2 bandwidth is idle and the only enemy is exposed latency
// -- which co-resident workgroups hide for each other.  K is walked in BK=32 slabs
// staged through LDS: coalesced 16-byte global loads (8 lanes cover one 128-byte row
// segment) -> registers -> ds_write_b128 into rows padded to 40 floats (160 B), the
// stride for which the fragment reads -- one ds_read_b128 per lane = four
// consecutive k of one row = operands of four MFMAs -- are bank-conflict free in
// every 16-lane service group of ds_read_b128.  The next slab's global loads are
// issued before the current slab's MFMAs (register double buffering + two LDS
// buffers, one barrier per slab).  Workgroup ids are remapped so that each XCD
// (private 4 MiB L2) owns a contiguous range of N-tiles, i.e. streams 1/8 of W.
#include "common.h"

namespace after {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = 40;  // floats per LDS row (32 + 8 pad): conflict-free b128 fragment reads

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

template <int MT, int NT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int BM = 32 * MT, BN = 32 * NT;            // workgroup tile
    constexpr int A_LOADS = BM / 32, W_LOADS = BN / 32;  // float4 per thread per slab
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

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 ra[A_LOADS], rw[W_LOADS];
    const int nk = (K + BK - 1) / BK;
    tile_gload<A_LOADS>(ra, gA, Aoff, Aok, 0, lc4 < K);
    tile_gload<W_LOADS>(rw, gW, Woff, Wok, 0, lc4 < K);
    tile_sstore<A_LOADS>(ra, As + lrow * LDS_LD + lc4);
    tile_sstore<W_LOADS>(rw, Ws + lrow * LDS_LD + lc4);
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
            tile_sstore<A_LOADS>(ra, As + ((cur ^ 1) * BM + lrow) * LDS_LD + lc4);
            tile_sstore<W_LOADS>(rw, Ws + ((cur ^ 1) * BN + lrow) * LDS_LD + lc4);
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

template <int MT, int NT>
int launch_cfg(const GemmArgs& g, hipStream_t stream) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
    const size_t lds = size_t(2) * (BM + BN) * LDS_LD * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        AFTER_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<MT, NT>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<MT, NT>), dim3(tiles_m * tiles_n), dim3(256), lds, stream, g,
                       tiles_m, tiles_n);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

}  // namespace

int launch_gemm(const GemmArgs& g, hipStream_t stream) { return launch_gemm_cfg(g, 0, 0, stream); }

int launch_gemm_cfg(const GemmArgs& g, int mt, int nt, hipStream_t stream) {
    AFTER_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, AFTER_E_INVALID, "gemm: empty problem %dx%dx%d",
                  g.M, g.N, g.K);
    AFTER_REQUIRE((g.K % 4) == 0 && (g.lda % 4) == 0 && (g.ldw % 4) == 0, AFTER_E_INVALID,
                  "gemm: K/lda/ldw must be multiples of 4 (K=%d lda=%d ldw=%d)", g.K, g.lda, g.ldw);
    AFTER_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.W % 16) == 0, AFTER_E_INVALID,
                  "gemm: operands must be 16-byte aligned");
    AFTER_REQUIRE(g.epilogue != EPI_RESIDUAL || g.R != nullptr, AFTER_E_INVALID,
                  "gemm: residual epilogue without R");
    if (mt > 0 || nt > 0) {
        if (mt == 4 && nt == 4) return launch_cfg<4, 4>(g, stream);
        if (mt == 4 && nt == 2) return launch_cfg<4, 2>(g, stream);
        if (mt == 2 && nt == 2) return launch_cfg<2, 2>(g, stream);
        if (mt == 1 && nt == 2) return launch_cfg<1, 2>(g, stream);
        if (mt == 1 && nt == 1) return launch_cfg<1, 1>(g, stream);
        set_error("gemm: no tile configuration %dx%d", mt, nt);
        return AFTER_E_INVALID;
    }
    // Tile choice: the largest workgroup tile that still yields >= 2 workgroups per CU
    // (co-residency is what hides the LDS/barrier latency of this MFMA-bound loop).
    auto wgs = [&](int bm, int bn) { return (long long)cdiv(g.M, bm) * cdiv(g.N, bn); };
    const long long want = 2 * 256;
    if (wgs(128, 128) >= want) return launch_cfg<4, 4>(g, stream);
    if (wgs(128, 64) >= want) return launch_cfg<4, 2>(g, stream);
    if (wgs(64, 64) >= want) return launch_cfg<2, 2>(g, stream);
    if (wgs(32, 64) >= want) return launch_cfg<1, 2>(g, stream);
    return launch_cfg<1, 1>(g, stream);
}

}  // namespace after

// Diagnostic / unit-test entry point (not on the reference's surface): the GEMM used by
// every Linear of the denoiser, callable on its own for parity and roofline tests.
// force_mt/force_nt > 0 pin the tile configuration (workgroup tile 32*mt x 32*nt).
extern "C" int after_gemm_f32(const float* A, int lda, const float* W, int ldw, const float* bias,
                              const float* R, int ldr, float* C, int ldc, int M, int N, int K,
                              int epilogue, int force_mt, int force_nt, void* stream) {
    after::GemmArgs g{A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, epilogue};
    return after::launch_gemm_cfg(g, force_mt, force_nt, (hipStream_t)stream);
}
