// Shared host-side helpers of libafter_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/after_hip.h"

namespace after {

void set_error(const char* fmt, ...);

#define AFTER_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            ::after::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),     \
                               __FILE__, __LINE__);                                        \
            return AFTER_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define AFTER_REQUIRE(cond, code, ...)         \
    do {                                       \
        if (!(cond)) {                         \
            ::after::set_error(__VA_ARGS__);   \
            return (code);                     \
        }                                      \
    } while (0)

#define AFTER_TRY(expr)             \
    do {                            \
        int rc__ = (expr);          \
        if (rc__ != AFTER_OK) return rc__; \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }

// Device arena: one hipMalloc per handle, bump-allocated at create time so that
// nothing is allocated on the hot path.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0;
    int init(size_t bytes);
    void release();
    template <typename T>
    T* take(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
        if (off + bytes > cap) return nullptr;
        T* p = reinterpret_cast<T*>(base + off);
        off += bytes;
        return p;
    }
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the function ON A DEVICE: a launcher remembers what it has
// set per device (a process that moves a handle to a second GPU would otherwise launch > 64 KB of dynamic LDS without it)
struct LdsAttr {
    size_t set[16] = {};
};
inline int ensure_lds_attr(LdsAttr& st, const void* fn, size_t lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    dev &= 15;
    if (lds > st.set[dev]) {
        AFTER_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        st.set[dev] = lds;
    }
    return AFTER_OK;
}

// Optional per-kernel-family timing with HIP events on the launch stream.
struct KernelTimer {
    bool enabled = false;
    static constexpr int kMax = 8192;
    hipEvent_t* ev = nullptr;  // 2*kMax
    int n = 0;
    double flops = 0, bytes = 0;  // algorithmic work of the bracketed launches
    int enable(bool on);
    void begin(hipStream_t s) {
        if (enabled && n < kMax) (void)hipEventRecord(ev[2 * n], s);
    }
    void end(hipStream_t s, double fl, double by = 0) {
        if (enabled && n < kMax) {
            (void)hipEventRecord(ev[2 * n + 1], s);
            ++n;
            flops += fl;
            bytes += by;
        }
    }
    int collect(double* ms, long long* launches, double* fl, double* by = nullptr);
    void destroy();
};

// exact-erf GELU of the reference's MLP (transformerv2.py:275-283: nn.GELU()) for the GEMM epilogues:
//   gelu(x) = x/2 (1 + erf(x / sqrt 2)),  erf(t) = 1 - 2^(t q(t)) for t = min(|x| / sqrt 2, 4)   (erf(4) = 1 - 1.5e-8)
// q: degree-8 minimax fit of log2(erfc(t)) / t on [0, 4]; evaluated in fp32 the formula is within 1.3e-7 of erf
// and the GELU within 4.2e-7 of the fp64 value over |x| < 8 -- the same as the fp32 erff formula (4.5e-7) -- in 15
// instructions and one v_exp_f32 instead of libm erff's two divergent branches (~30): the epilogue of the MLP-up
// GEMM was 1.9 us of its 12.3 us at one clip and 8.6 of 60 us at eight.
__device__ __forceinline__ float gelu_erf(float x) {
    const float t = fminf(fabsf(x) * 0.70710678118654752440f, 4.0f);
    float r = 1.038623122e-05f;
    r = fmaf(r, t, -1.402917551e-04f);
    r = fmaf(r, t, 7.946707774e-04f);
    r = fmaf(r, t, -2.155642258e-03f);
    r = fmaf(r, t, -6.602090434e-05f);
    r = fmaf(r, t, 2.783408947e-02f);
    r = fmaf(r, t, -1.483516544e-01f);
    r = fmaf(r, t, -9.184343815e-01f);
    r = fmaf(r, t, -1.627907872e+00f);
    const float e = __builtin_amdgcn_exp2f(r * t);
    const float hx = 0.5f * x;
    return fmaf(hx, copysignf(1.0f - e, x), hx);
}

// ---------------------------------------------------------------- GEMM (gemm.hip)
// C[M,N] = epilogue(A[M,K] * W[N,K]^T): fp32 in, fp32 MFMA (v_mfma_f32_32x32x2_f32),
// fp32 out.  A, W row-major with K contiguous (lda/ldw multiples of 4 floats,
// 16-byte aligned bases).  Arbitrary M, N; K arbitrary (zero-filled past K).
enum GemmEpilogue {
    EPI_NONE = 0,       // C = acc (+ bias)
    EPI_GELU = 1,       // C = gelu_erf(acc + bias)
    EPI_RESIDUAL = 2,   // C = acc + bias + R[m, n]
    EPI_RELU = 3,       // C = max(acc + bias, 0)
    EPI_SIGMOID = 4,    // C = sigmoid(acc + bias)
    EPI_CFG_EULER = 5,  // fused sampler tail (launch_gemm_cfg_euler only)
};
struct GemmArgs {
    const float* A;
    int lda;
    const float* W;
    int ldw;
    const float* bias;  // [N] or nullptr
    const float* R;     // residual [M, ldr] (EPI_RESIDUAL)
    int ldr;
    float* C;
    int ldc;
    int M, N, K;
    int epilogue;
    unsigned long long* dbg = nullptr;  // diagnostics: 4 s_memtime stamps per workgroup
    // ---- EPI_CFG_EULER (launch_gemm_cfg_euler): A = final residual stream [3 * BT, K] with the
    // CFG branches stacked (full | mid | none), W / bias = out_proj.  Per token and channel:
    //   d = d_none + total (d_mid + factor (d_full - d_mid) - d_none);  x' = xin + dt d
    // written to xout [B, N, T] (reference layout) and xt [BT, xt_ld] (token-major, the next
    // step's patchify input).  cfg -> {total, factor, dt} in device memory.
    const float* xin = nullptr;
    float* xout = nullptr;
    float* xt = nullptr;
    const float* cfg = nullptr;
    int T = 0, xt_ld = 0;
};
// out_proj + CFG combine + Euler update of one sampler step in ONE launch; g.M = 3 * B * T
int launch_gemm_cfg_euler(const GemmArgs& g, hipStream_t stream);
int launch_gemm(const GemmArgs& g, hipStream_t stream);
int launch_gemm_cfg(const GemmArgs& g, int force_mt, int force_nt, hipStream_t stream);
// fp32 GEMM through the bf16 matrix pipe (gemm_x6.hip): both operands as three bf16 planes (x = h + m + l
// exactly), six v_mfma_f32_16x16x32_bf16 per 32-deep step, fp32 accumulate, fp32 result -- or, for a producer of
// the next GEMM's A operand, the result's three planes (C3).
//
// Plane storage ("x6 blocks"): a [R][K] matrix (K % 32 == 0, rows padded to a multiple of 16) is a sequence of
// 1-KB blocks [R / 16][K / 32][plane]: 16 rows x 32 k of one plane, row r at byte 64 r, its four 16-byte chunks
// XOR-permuted (chunk c at slot c ^ f((r / 4) % 4), f = {0, 2, 3, 1}).  A block is exactly one LDS-DMA piece of
// the GEMM and already its LDS image: a wave instruction moves 1 KB of contiguous memory (eight full 128-byte
// lines; row-major planes would be sixteen 64-byte segments, half the rate of the CU's load path) and the
// fragment reads (lane l: row l & 15, chunk l >> 4) are bank-conflict free.
__host__ __device__ inline size_t x6_rows_padded(int rows) { return ((size_t)rows + 15) & ~(size_t)15; }
__host__ __device__ inline size_t x6_elems(int rows, int K) { return x6_rows_padded(rows) * 3 * (size_t)K; }
// element (unsigned short) offset of (row r, plane p, column k) -- k % 8 consecutive elements stay contiguous
__host__ __device__ inline size_t x6_offset(int r, int p, int k, int K) {
    const int rr = r & 15, c = (k & 31) >> 3;
    const int slot = c ^ ((0x78 >> (2 * ((rr >> 2) & 3))) & 3);
    return (((size_t)(r >> 4) * (size_t)(K >> 5) + (size_t)(k >> 5)) * 3 + (size_t)p) * 512 + (size_t)(rr * 32 + slot * 8 + (k & 7));
}
struct X6GemmArgs {
    const unsigned short* A3;  // x6 blocks of A [M][K]
    const unsigned short* W3;  // x6 blocks of W [N][K] (padding rows zero)
    const float* bias;
    const float* R;            // residual [M, ldr] (EPI_RESIDUAL, fp32 output only)
    int ldr;
    float* C;                  // fp32 output [M][ldc], or nullptr
    unsigned short* C3;        // x6 blocks of the output [M][N] (N % 32 == 0), or nullptr
    int ldc;
    int M, N, K;
    int epilogue;
    unsigned long long* dbg = nullptr;
};
int gemm_x6_split(const float* W, int ldw, unsigned short* W3, int N, int K, hipStream_t s);
// the same matrix x scale as TWO fp16 pieces in "h3 blocks" (gemm_h3_pipe.h: the x6 block layout with two planes; pad16(N) x 2 x K
// unsigned shorts; scale: an exact power of two with max|W| x scale <= 2^15)
int gemm_h3_split(const float* W, int ldw, unsigned short* W2, int N, int K, float scale, hipStream_t s);
int gemm_x6_pick_tile(int M, int N, int K);
int launch_gemm_x6(const X6GemmArgs& g, int tile, hipStream_t stream);

// four consecutive floats of a row (columns k .. k + 3, k % 4 == 0) -> their bf16 planes, stored into the x6
// blocks at `base`: what the producers of a gemm_x6 A operand do (ln_mod_ln / attention LayerNorm tail in
// denoiser.hip, gemm_x6's own plane-output epilogue)
__device__ __forceinline__ void x6_split4(float x0, float x1, float x2, float x3, uint2& h, uint2& m, uint2& l) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v0 = {x0, x1}, v1 = {x2, x3};
    h.x = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, bf2));
    h.y = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, bf2));
    const f2 r0 = {x0 - __uint_as_float(h.x << 16), x1 - __uint_as_float(h.x & 0xFFFF0000u)};
    const f2 r1 = {x2 - __uint_as_float(h.y << 16), x3 - __uint_as_float(h.y & 0xFFFF0000u)};
    m.x = __builtin_bit_cast(unsigned, __builtin_convertvector(r0, bf2));
    m.y = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf2));
    const f2 t0 = {r0[0] - __uint_as_float(m.x << 16), r0[1] - __uint_as_float(m.x & 0xFFFF0000u)};
    const f2 t1 = {r1[0] - __uint_as_float(m.y << 16), r1[1] - __uint_as_float(m.y & 0xFFFF0000u)};
    l.x = __builtin_bit_cast(unsigned, __builtin_convertvector(t0, bf2));
    l.y = __builtin_bit_cast(unsigned, __builtin_convertvector(t1, bf2));
}
__device__ __forceinline__ void x6_store4(unsigned short* base, int r, int k, int K, float x0, float x1, float x2, float x3) {
    uint2 h, m, l;
    x6_split4(x0, x1, x2, x3, h, m, l);
    unsigned short* p = base + x6_offset(r, 0, k, K);
    *reinterpret_cast<uint2*>(p) = h;
    *reinterpret_cast<uint2*>(p + 512) = m;
    *reinterpret_cast<uint2*>(p + 1024) = l;
}

}  // namespace after
