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
// fp32 GEMM through the bf16 matrix pipe (gemm_x6.hip; experimental, opt-in): W3 = [N][3][K] bf16 planes of W
int gemm_x6_split(const float* W, int ldw, unsigned short* W3, int N, int K, hipStream_t s);
int launch_gemm_x6(const GemmArgs& g, const unsigned short* W3, int tile, hipStream_t stream);

}  // namespace after
