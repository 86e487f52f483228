// The once-per-clip conditioning encoders on gfx950:
//   encoder_time = Encoder1D   (reference after/diffusion/networks/encoder.py:116-322)
//   encoder      = ECAPATDNN   (reference after/diffusion/networks/ecapa_encoder.py:458-666)
// Both are stacks of small 1-D convs on the time-major conv path of conv_tm.hip.  Encoder1D:
// BatchNorm(eval) + SiLU of the NEXT conv applied by the producing conv's second output
// (offline) or by act_pad_tm with the streaming state.  ECAPA: epilogue ReLU + BatchNorm affine
// for TDNNBlock, reflect padding, the Res2Net chain through second outputs, T-parallel pooling
// statistics and one-wave-per-output GEMMs for the per-clip vectors.
#include <new>
#include <vector>

#include "conv.h"

namespace after {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// per-(b, c) statistics over time (ecapa_encoder.py AttentiveStatisticsPooling):
//   w = softmax_t(logits[b,c,:]) (or 1/T when logits == nullptr)
//   mean = sum_t w z ;  std = sqrt(clamp(sum_t w (z - mean)^2, 1e-12))
// out[b, c] = mean, out[b, C + c] = std, each optionally followed by a per-channel
// affine (asp_bn).  One wave per (b, c) row.
__global__ __launch_bounds__(256) void time_stats_kernel(const float* __restrict__ z,
                                                         const float* __restrict__ logits,
                                                         float* __restrict__ out,
                                                         const float* __restrict__ post_scale,
                                                         const float* __restrict__ post_shift,
                                                         int B, int C, int T, int z_bstride,
                                                         int only_mean) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * C) return;
    const int b = row / C, c = row - b * C;
    const float* zr = z + (size_t)b * z_bstride + (size_t)c * T;
    const float* lr = logits ? logits + ((size_t)b * C + c) * T : nullptr;
    float mx = -INFINITY;
    if (lr)
        for (int t = lane; t < T; t += 64) mx = fmaxf(mx, lr[t]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = 0.f, sz = 0.f;
    for (int t = lane; t < T; t += 64) {
        const float w = lr ? expf(lr[t] - mx) : 1.0f;
        se += w;
        sz += w * zr[t];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        se += __shfl_xor(se, o, 64);
        sz += __shfl_xor(sz, o, 64);
    }
    const float inv = 1.0f / se;
    const float mean = sz * inv;
    float sv = 0.f;
    for (int t = lane; t < T; t += 64) {
        const float w = (lr ? expf(lr[t] - mx) : 1.0f) * inv;
        const float d = zr[t] - mean;
        sv += w * d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o, 64);
    if (lane == 0) {
        float m = mean, sd = sqrtf(fmaxf(sv, 1e-12f));
        const int w2 = only_mean ? C : 2 * C;
        if (post_scale) {
            m = m * post_scale[c] + post_shift[c];
            if (!only_mean) sd = sd * post_scale[C + c] + post_shift[C + c];
        }
        out[(size_t)b * w2 + c] = m;
        if (!only_mean) out[(size_t)b * w2 + C + c] = sd;
    }
}

// SEBlock tail + residual: y[b,c,t] = s[b,c] * x[b,c,t] + res[b,c,t]   (ecapa SERes2NetBlock)
__global__ __launch_bounds__(256) void se_scale_add_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ s,
                                                           const float* __restrict__ res,
                                                           float* __restrict__ y, int C, int T,
                                                           int res_bstride, int y_bstride,
                                                           size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int t = idx % T;
    const int c = (idx / T) % C;
    const int b = idx / ((size_t)T * C);
    y[(size_t)b * y_bstride + (size_t)c * T + t] =
        s[(size_t)b * C + c] * x[idx] + res[(size_t)b * res_bstride + (size_t)c * T + t];
}

__global__ __launch_bounds__(256) void copy_slice_kernel(const float* __restrict__ x,
                                                         float* __restrict__ y, int Cs, int T,
                                                         int x_bstride, int y_bstride, size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const size_t per = (size_t)Cs * T;
    const int b = idx / per;
    const size_t r = idx - (size_t)b * per;
    y[(size_t)b * y_bstride + r] = x[(size_t)b * x_bstride + r];
}

__global__ void tanh_kernel(float* x, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = tanhf(x[i]);
}

// ---- time-major ([B][T][C]) variants used by ECAPA on the conv_tm path
// per-(b, c) statistics over time: a block owns 64 consecutive channels (coalesced rows) and
// splits T over its 4 waves; the three passes of time_stats_kernel (max, weighted sums, weighted
// variance) each end in a 4-way LDS combine.  Same arithmetic as time_stats_kernel up to the
// association of the T sums.
__global__ __launch_bounds__(256) void time_stats_tm_kernel(const float* __restrict__ z, int ldz,
                                                            const float* __restrict__ logits, int ldl,
                                                            float* __restrict__ out,
                                                            const float* __restrict__ post_scale,
                                                            const float* __restrict__ post_shift, int C,
                                                            int T, int only_mean) {
    __shared__ float sh[3][4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, b = blockIdx.y;
    const bool ok = c < C;
    const int per = (T + 3) / 4, t_lo = sl * per, t_hi = min(T, t_lo + per);
    const float* zr = z + (size_t)b * T * ldz + (ok ? c : 0);
    const float* lr = logits ? logits + (size_t)b * T * ldl + (ok ? c : 0) : nullptr;
    float mx = -INFINITY;
    if (lr) {
#pragma unroll 8
        for (int t = t_lo; t < t_hi; ++t) mx = fmaxf(mx, lr[(size_t)t * ldl]);
        sh[0][sl][cl] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(sh[0][0][cl], sh[0][1][cl]), fmaxf(sh[0][2][cl], sh[0][3][cl]));
    }
    float se = 0.f, sz = 0.f;
#pragma unroll 8
    for (int t = t_lo; t < t_hi; ++t) {
        const float w = lr ? expf(lr[(size_t)t * ldl] - mx) : 1.0f;
        se += w;
        sz += w * zr[(size_t)t * ldz];
    }
    sh[1][sl][cl] = se;
    sh[2][sl][cl] = sz;
    __syncthreads();
    se = (sh[1][0][cl] + sh[1][1][cl]) + (sh[1][2][cl] + sh[1][3][cl]);
    sz = (sh[2][0][cl] + sh[2][1][cl]) + (sh[2][2][cl] + sh[2][3][cl]);
    const float inv = 1.0f / se;
    const float mean = sz * inv;
    float sv = 0.f;
#pragma unroll 8
    for (int t = t_lo; t < t_hi; ++t) {
        const float w = (lr ? expf(lr[(size_t)t * ldl] - mx) : 1.0f) * inv;
        const float d = zr[(size_t)t * ldz] - mean;
        sv += w * d * d;
    }
    __syncthreads();  // the sums above have been read by every wave
    sh[0][sl][cl] = sv;
    __syncthreads();
    if (sl != 0 || !ok) return;
    sv = (sh[0][0][cl] + sh[0][1][cl]) + (sh[0][2][cl] + sh[0][3][cl]);
    float m = mean, sd = sqrtf(fmaxf(sv, 1e-12f));
    const int w2 = only_mean ? C : 2 * C;
    if (post_scale) {
        m = m * post_scale[c] + post_shift[c];
        if (!only_mean) sd = sd * post_scale[C + c] + post_shift[C + c];
    }
    out[(size_t)b * w2 + c] = m;
    if (!only_mean) out[(size_t)b * w2 + C + c] = sd;
}

// out[m, n] = epi(sum_k a[m, k] w[n, k] + bias[n]) for the handful of per-clip vectors of ECAPA
// (SE squeeze, attention context bias, final fc: M = B rows, K up to 3072): one wave per output.
__global__ __launch_bounds__(256) void rowvec_gemm_kernel(const float* __restrict__ a, int lda,
                                                          const float* __restrict__ w, int ldw,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ out, int ldo, int M, int N,
                                                          int K, int epi) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= M * N) return;
    const int m = o / N, n = o - m * N;
    const float* ar = a + (size_t)m * lda;
    const float* wr = w + (size_t)n * ldw;
    float acc = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {  // K % 4 == 0, rows 16-byte aligned
        const f32x4 av = *reinterpret_cast<const f32x4*>(ar + k);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
        acc += (av[0] * wv[0] + av[1] * wv[1]) + (av[2] * wv[2] + av[3] * wv[3]);
    }
#pragma unroll
    for (int s2 = 32; s2 > 0; s2 >>= 1) acc += __shfl_xor(acc, s2, 64);
    if (lane == 0) {
        float v = acc + (bias ? bias[n] : 0.f);
        if (epi == EPI_RELU) v = fmaxf(v, 0.f);
        else if (epi == EPI_SIGMOID) v = 1.0f / (1.0f + expf(-v));
        out[(size_t)m * ldo + n] = v;
    }
}

// y[b][t][c] = s[b][c] * x[b][t][c] + res[b][t][c] on row-pitched views (4 channels per thread)
__global__ __launch_bounds__(256) void se_scale_add_tm_kernel(const float* __restrict__ x, int ldx,
                                                              const float* __restrict__ sc,
                                                              const float* __restrict__ res, int ldr,
                                                              float* __restrict__ y, int ldy, int C, int T,
                                                              size_t total4) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int q = C / 4;
    const int c = 4 * (idx % q);
    const size_t bt = idx / q;  // b * T + t
    const int b = bt / T;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + bt * ldx + c);
    const f32x4 sv = *reinterpret_cast<const f32x4*>(sc + (size_t)b * C + c);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(res + bt * ldr + c);
    *reinterpret_cast<f32x4*>(y + bt * ldy + c) = sv * xv + rv;
}

// y[b][t][0:Cs] = x[b][t][0:Cs] (row-pitched slice copy)
__global__ __launch_bounds__(256) void copy_slice_tm_kernel(const float* __restrict__ x, int ldx,
                                                            float* __restrict__ y, int ldy, int Cs,
                                                            size_t total4) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int q = Cs / 4;
    const int c = 4 * (idx % q);
    const size_t bt = idx / q;
    *reinterpret_cast<f32x4*>(y + bt * ldy + c) = *reinterpret_cast<const f32x4*>(x + bt * ldx + c);
}

struct WCursor {
    const float* const* w;
    int n, i = 0;
    bool ok = true;
    const float* next() {
        if (i >= n || !w[i]) {
            ok = false;
            ++i;
            return nullptr;
        }
        return w[i++];
    }
};

struct Affine {
    float *scale = nullptr, *shift = nullptr;
};
struct ConvW {
    float *w = nullptr, *bias = nullptr;
    int cin = 0, cout = 0, k = 1;
};

int take(Arena& a, float** p, size_t n) {
    *p = a.take<float>(n);
    AFTER_REQUIRE(*p, AFTER_E_NOMEM, "encoder: weight arena exhausted");
    return AFTER_OK;
}

int load_bn(Arena& a, WCursor& c, Affine& af, int C) {
    const float *w = c.next(), *b = c.next(), *rm = c.next(), *rv = c.next();
    AFTER_REQUIRE(c.ok, AFTER_E_INVALID, "encoder: missing BatchNorm tensors");
    AFTER_TRY(take(a, &af.scale, C));
    AFTER_TRY(take(a, &af.shift, C));
    return launch_bn_affine(w, b, rm, rv, af.scale, af.shift, C, 1, 1e-5f, 0);
}

int load_conv(Arena& a, WCursor& c, ConvW& cw, int cin, int cout, int k, bool weight_norm) {
    cw.cin = cin;
    cw.cout = cout;
    cw.k = k;
    const float* g = weight_norm ? c.next() : nullptr;
    const float* v = c.next();
    const float* b = c.next();
    AFTER_REQUIRE(c.ok, AFTER_E_INVALID, "encoder: missing conv tensors");
    AFTER_TRY(take(a, &cw.w, (size_t)cout * k * pad16(cin)));
    AFTER_TRY(pack_conv_weight(v, g, cw.w, cout, cin, k, pad16(cin), 0));
    AFTER_TRY(take(a, &cw.bias, cout));
    AFTER_HIP_CHECK(hipMemcpy(cw.bias, b, cout * sizeof(float), hipMemcpyDeviceToDevice));
    return AFTER_OK;
}

size_t conv_floats(int cin, int cout, int k) { return (size_t)cout * k * pad16(cin) + cout + 256; }

}  // namespace
}  // namespace after

using namespace after;

// =================================================================== Encoder1D
// Time-major conv path (conv_tm.hip).  BatchNorm(eval) + SiLU have no data dependency on the whole
// tensor, so OFFLINE every conv's epilogue writes the next conv's activated, haloed input itself
// (ConvTmRun::y2): one act_pad launch for the network input, none in between.  STREAMING
// (cached_conv semantics: the left halo is the previous chunk's tail) keeps one act_pad per conv,
// which owns the state hand-over.
struct EncConv {
    ConvW cw;
    ConvDmaPlanIn in;
    ConvTmPlan plan;
    float* wd = nullptr;  // [Cout][taps * Cp] GEMM operand
};
struct V2Block {
    Affine bn0, bn1;
    EncConv c0, c1;
};

struct after_encoder1d {
    after_encoder1d_cfg cfg;
    int max_batch, max_T, cmax = 0;
    Arena wa, ws, sa;
    std::vector<V2Block> blocks;  // n + 1 (last = final V2ConvBlock1D)
    std::vector<EncConv> pools;   // n
    float* buf[3] = {nullptr, nullptr, nullptr};  // raw time-major activations [B][T][C]
    float* xp[2] = {nullptr, nullptr};            // activated + haloed conv inputs
    size_t xp_elems = 0;
    // streaming: one left-context slot [max_batch][halo][Cp(cmax)] per temporal conv
    bool streaming = false;
    float* state = nullptr;
    size_t slot_elems = 0;
    int n_slots = 0, flip = 0;
    int stream_rows = 0;  // rows of the streams, fixed by the first chunk after a reset (a pass updates ITS rows' contexts only)
    int slot = 0;
};

namespace {

int plan_enc(Arena& a, EncConv& e, int stride, bool causal) {
    memset(&e.in, 0, sizeof(e.in));
    e.in.Cin = e.cw.cin;
    e.in.Cout = e.cw.cout;
    e.in.taps = e.cw.k;
    e.in.phases = 1;
    e.in.istride = stride;
    e.in.ostride = 1;
    const int pl = conv_left_pad(e.cw.k, 1, causal);
    for (int t = 0; t < e.cw.k; ++t) e.in.toff[0][t] = t - pl;
    conv_tm_plan(e.in, &e.plan);
    AFTER_REQUIRE(e.plan.ok, AFTER_E_INVALID, "encoder1d: tap pattern outside the conv path");
    e.wd = a.take<float>(e.plan.w_floats);
    AFTER_REQUIRE(e.wd, AFTER_E_NOMEM, "encoder1d: weight arena exhausted");
    return conv_tm_repack(e.cw.w, e.wd, e.in, e.plan, 0);
}

int load_v2(Arena& a, WCursor& c, V2Block& b, int ch, int k) {
    AFTER_TRY(load_bn(a, c, b.bn0, ch));
    AFTER_TRY(load_conv(a, c, b.c0.cw, ch, ch, k, true));
    AFTER_TRY(load_bn(a, c, b.bn1, ch));
    return load_conv(a, c, b.c1.cw, ch, ch, k, true);
}

// what a conv's epilogue prepares for its consumer
struct Next {
    float* y2 = nullptr;       // haloed buffer to fill, or nullptr
    const Affine* bn = nullptr;
    int act = ACT_NONE;
};

// act(affine(x)) -> haloed scratch (+ streaming left context)
int enc_act(after_encoder1d* h, hipStream_t s, const float* x, int x_cm, const Affine* bn, int act, int C,
            int B, int T, float* dst, bool temporal) {
    AFTER_REQUIRE((size_t)B * conv_tm_cp(C) * conv_tm_rows(T) <= h->xp_elems, AFTER_E_CAPACITY,
                  "encoder1d: activation scratch too small");
    ActPadTm p;
    memset(&p, 0, sizeof(p));
    p.x = x;
    p.y = dst;
    p.gamma = bn ? bn->scale : nullptr;
    p.beta = bn ? bn->shift : nullptr;
    p.act = act;
    p.B = B;
    p.C = C;
    p.T = T;
    p.G = 1;
    p.x_cm = x_cm;
    if (h->streaming && temporal) {  // (the context exists twice: read half `flip`, write the other -- one launch per conv)
        const int slot = h->slot++;
        p.state = h->state + ((size_t)h->flip * h->n_slots + slot) * h->slot_elems;
        p.state_out = h->state + ((size_t)(h->flip ^ 1) * h->n_slots + slot) * h->slot_elems;
    }
    return launch_act_pad_tm(p, s);
}

// one conv: src = haloed input (src_raw = 0) or a raw time-major tensor used in place (k = 1)
int enc_conv(after_encoder1d* h, hipStream_t s, const EncConv& e, const float* src, bool src_raw,
             const float* res, int res_cm, float* y, int y_cm, const Next& nx, int B, int Tin) {
    ConvTmRun r;
    memset(&r, 0, sizeof(r));
    r.w = e.wd;
    r.bias = e.cw.bias;
    r.res = res;
    r.res_cm = res_cm;
    r.y = y;
    r.y_cm = y_cm;
    r.G = 1;
    r.B = B;
    r.Tout = Tin / e.in.istride;
    r.Nn = r.Tout;
    if (src_raw) {  // k = 1, stride 1, Cin % 32 == 0: row n of the raw tensor is "haloed row halo + n"
        r.xp = src - (size_t)conv_tm_halo() * e.cw.cin;
        r.Tp = Tin;
        r.x_ld = e.cw.cin;
    } else {
        r.xp = src;
        r.Tp = conv_tm_rows(Tin);
    }
    if (nx.y2) {
        r.y2 = nx.y2;
        r.y2_scale = nx.bn ? nx.bn->scale : nullptr;
        r.y2_shift = nx.bn ? nx.bn->shift : nullptr;
        r.y2_act = nx.act;
    }
    return launch_conv_tm(r, e.in, e.plan, s);
}

bool fusable(int C) { return (C & 31) == 0; }  // y2 rows are written without channel padding

}  // namespace

extern "C" int after_encoder1d_create(const after_encoder1d_cfg* cfg, const float* const* weights,
                                      int n_weights, int max_batch, int max_T,
                                      after_encoder1d** out) {
    AFTER_REQUIRE(cfg && weights && out, AFTER_E_INVALID, "null argument");
    *out = nullptr;
    AFTER_REQUIRE(cfg->n_blocks >= 1 && cfg->n_blocks <= 8 && cfg->kernel_size >= 1 &&
                      cfg->kernel_size <= kMaxTaps && max_batch > 0 && max_T > 0,
                  AFTER_E_INVALID, "encoder1d: bad configuration");
    after_encoder1d* h = new (std::nothrow) after_encoder1d();
    AFTER_REQUIRE(h, AFTER_E_NOMEM, "out of host memory");
    h->cfg = *cfg;
    h->max_batch = max_batch;
    h->max_T = max_T;
    const int n = cfg->n_blocks, k = cfg->kernel_size;
    const bool causal = cfg->causal != 0;
    auto fail = [&](int rc) {
        after_encoder1d_destroy(h);
        return rc;
    };
    size_t wf = 0;
    int cmax = cfg->in_size;
    {
        int c = cfg->in_size;
        for (int i = 0; i < n; ++i) {
            wf += 2 * conv_floats(c, c, k) + 4 * (size_t)c + 1024 +
                  conv_floats(c, cfg->channels[i], 2 * cfg->ratios[i]);
            c = cfg->channels[i];
            cmax = c > cmax ? c : cmax;
        }
        wf += 2 * conv_floats(c, c, k) + 4 * (size_t)c + 1024;
    }
    h->cmax = cmax;
    // packed weights + their GEMM re-layout (channels padded to the 32-deep K slab)
    int rc = h->wa.init(wf * 5 * sizeof(float) + (4 << 20));
    if (rc) return fail(rc);
    WCursor cur{weights, n_weights};
    h->blocks.resize(n + 1);
    h->pools.resize(n);
    int c = cfg->in_size;
    for (int i = 0; i < n; ++i) {
        if ((rc = load_v2(h->wa, cur, h->blocks[i], c, k))) return fail(rc);
        if ((rc = plan_enc(h->wa, h->blocks[i].c0, 1, causal))) return fail(rc);
        if ((rc = plan_enc(h->wa, h->blocks[i].c1, 1, causal))) return fail(rc);
        const int r = cfg->ratios[i];
        if (r < 1 || 2 * r > kMaxTaps) {
            set_error("encoder1d: ratio %d unsupported", r);
            return fail(AFTER_E_INVALID);
        }
        if ((rc = load_conv(h->wa, cur, h->pools[i].cw, c, cfg->channels[i], r == 1 ? 1 : 2 * r, true)))
            return fail(rc);
        if ((rc = plan_enc(h->wa, h->pools[i], r, causal))) return fail(rc);
        c = cfg->channels[i];
    }
    if ((rc = load_v2(h->wa, cur, h->blocks[n], c, k))) return fail(rc);
    if ((rc = plan_enc(h->wa, h->blocks[n].c0, 1, causal))) return fail(rc);
    if ((rc = plan_enc(h->wa, h->blocks[n].c1, 1, causal))) return fail(rc);
    if (!cur.ok || cur.i != n_weights) {
        set_error("encoder1d: expected %d weight tensors, got %d", cur.i, n_weights);
        return fail(AFTER_E_INVALID);
    }
    const size_t elems = (size_t)max_batch * cmax * max_T;
    h->xp_elems = (size_t)max_batch * conv_tm_cp(cmax) * conv_tm_rows(max_T) + 4096;
    if ((rc = h->ws.init((3 * elems + 2 * h->xp_elems) * sizeof(float) + 16384))) return fail(rc);
    for (int i = 0; i < 3; ++i) h->buf[i] = h->ws.take<float>(elems);
    for (int i = 0; i < 2; ++i) h->xp[i] = h->ws.take<float>(h->xp_elems);
    if (!h->buf[2] || !h->xp[1]) return fail(AFTER_E_NOMEM);
    if (hipDeviceSynchronize() != hipSuccess) return fail(AFTER_E_HIP);
    *out = h;
    return AFTER_OK;
}

extern "C" void after_encoder1d_destroy(after_encoder1d* h) {
    if (!h) return;
    h->sa.release();
    h->wa.release();
    h->ws.release();
    delete h;
}

// streaming: `encoder_time.forward_stream` under cc.use_cached_conv(True) (export.py:17,
// 438-441; encoder.py:301-322): every causal conv keeps its k-1 (2r-1 for the strided pool)
// input samples between chunks.
extern "C" int after_encoder1d_enable_streaming(after_encoder1d* h, int enable) {
    AFTER_REQUIRE(h, AFTER_E_INVALID, "null handle");
    if (!enable) {
        h->streaming = false;
        return AFTER_OK;
    }
    AFTER_REQUIRE(h->cfg.causal, AFTER_E_INVALID,
                  "encoder1d: streaming needs causal padding (base.gin:55)");
    if (!h->sa.base) {
        const int slots = 3 * h->cfg.n_blocks + 2;
        h->slot_elems = (size_t)h->max_batch * conv_tm_cp(h->cmax) * conv_tm_halo();
        h->n_slots = slots;
        AFTER_TRY(h->sa.init(2 * (size_t)slots * h->slot_elems * sizeof(float) + 4096));
        h->state = h->sa.take<float>(2 * (size_t)slots * h->slot_elems);
        AFTER_REQUIRE(h->state, AFTER_E_NOMEM, "encoder1d: streaming state allocation failed");
        AFTER_HIP_CHECK(hipMemset(h->sa.base, 0, h->sa.off));
    }
    h->streaming = true;
    h->stream_rows = 0;
    return AFTER_OK;
}

extern "C" int after_encoder1d_reset_state(after_encoder1d* h, void* stream) {
    AFTER_REQUIRE(h && h->sa.base, AFTER_E_INVALID, "encoder1d: streaming was never enabled");
    h->flip = 0;
    h->stream_rows = 0;
    AFTER_HIP_CHECK(hipMemsetAsync(h->sa.base, 0, h->sa.off, (hipStream_t)stream));
    return AFTER_OK;
}

extern "C" int after_encoder1d_forward(after_encoder1d* h, const float* z, float* out, int B, int T,
                                       void* stream) {
    AFTER_REQUIRE(h && z && out, AFTER_E_INVALID, "null argument");
    AFTER_REQUIRE(B > 0 && T > 0, AFTER_E_INVALID, "empty batch");
    AFTER_REQUIRE(B <= h->max_batch && T <= h->max_T, AFTER_E_CAPACITY,
                  "B=%d T=%d exceed max_batch=%d max_T=%d", B, T, h->max_batch, h->max_T);
    if (h->streaming) {
        AFTER_REQUIRE(h->stream_rows == 0 || h->stream_rows == B, AFTER_E_INVALID,
                      "encoder1d: the streams were started with %d rows, the chunk has %d: the conv contexts ping-pong per pass, rows "
                      "outside a pass would keep a two-chunk-old context -- after_encoder1d_reset_state first", h->stream_rows, B);
        h->stream_rows = B;
    }
    hipStream_t s = (hipStream_t)stream;
    const after_encoder1d_cfg& c = h->cfg;
    const int n = c.n_blocks;
    h->slot = 0;
    // x: the block input, raw.  Block 0 reads the caller's [B][C][T] tensor in place (x_cm); later
    // blocks read time-major buffers.  xa: x already activated + haloed by the producer's epilogue.
    const float* x = z;
    int x_cm = 1;
    const float* xa = nullptr;
    int ch = c.in_size;
    float* const R[3] = {h->buf[0], h->buf[1], h->buf[2]};
    int ri = 0;  // next free raw buffer (round robin: a block needs x, Y and the pool output alive)
    for (int i = 0; i <= n; ++i) {
        // V2ConvBlock1D (encoder.py:25-71): y = conv1(silu(bn1(conv0(silu(bn0(x)))))) + x
        const V2Block& b = h->blocks[i];
        const bool fuse = !h->streaming && fusable(ch);
        if (!xa) {
            AFTER_TRY(enc_act(h, s, x, x_cm, &b.bn0, ACT_SILU, ch, B, T, h->xp[0], true));
            xa = h->xp[0];
        }
        const bool last = i == n;
        const int r = last ? 1 : c.ratios[i];
        float* Y = last ? out : R[ri++ % 3];
        float* mid = (xa == h->xp[0]) ? h->xp[1] : h->xp[0];
        if (fuse) {
            Next nx;
            nx.y2 = mid;
            nx.bn = &b.bn1;
            nx.act = ACT_SILU;
            AFTER_TRY(enc_conv(h, s, b.c0, xa, false, nullptr, 0, nullptr, 0, nx, B, T));
        } else {
            float* t = R[ri % 3];  // scratch only until the act_pad below has consumed it
            AFTER_TRY(enc_conv(h, s, b.c0, xa, false, nullptr, 0, t, 0, Next(), B, T));
            AFTER_TRY(enc_act(h, s, t, 0, &b.bn1, ACT_SILU, ch, B, T, mid, true));
        }
        // the strided pool (r > 1) needs Y with a halo: let conv1's epilogue write that copy
        float* pool_in = (xa == h->xp[0]) ? h->xp[0] : h->xp[1];  // xa's buffer: free once conv0 has run
        Next n1;
        if (!last && r > 1 && fuse) n1.y2 = pool_in;
        AFTER_TRY(enc_conv(h, s, b.c1, mid, false, x, x_cm, Y, last ? 1 : 0, n1, B, T));
        if (last) break;
        // V2EncoderBlock1D (encoder.py:74-113): the (strided) "pool" conv c -> channels[i]
        AFTER_REQUIRE(T % r == 0, AFTER_E_INVALID, "encoder1d: T=%d not divisible by ratio %d", T, r);
        const int cn = c.channels[i];
        float* X2 = R[ri++ % 3];
        const V2Block& nb = h->blocks[i + 1];
        Next nx;
        const bool nfuse = !h->streaming && fusable(cn);
        float* na = (pool_in == h->xp[0]) ? h->xp[1] : h->xp[0];
        if (nfuse) {
            nx.y2 = na;
            nx.bn = &nb.bn0;
            nx.act = ACT_SILU;
        }
        if (r == 1 && fusable(ch)) {
            AFTER_TRY(enc_conv(h, s, h->pools[i], Y, true, nullptr, 0, X2, 0, nx, B, T));
        } else {
            if (!n1.y2) AFTER_TRY(enc_act(h, s, Y, 0, nullptr, ACT_NONE, ch, B, T, pool_in, r > 1));
            AFTER_TRY(enc_conv(h, s, h->pools[i], pool_in, false, nullptr, 0, X2, 0, nx, B, T));
        }
        T /= r;
        x = X2;
        x_cm = 0;
        xa = nfuse ? na : nullptr;
        ch = cn;
    }
    if (c.use_tanh) {
        const int tot = B * c.channels[n - 1] * T;
        hipLaunchKernelGGL(tanh_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, s, out, tot);
        AFTER_HIP_CHECK(hipGetLastError());
    }
    if (h->streaming) h->flip ^= 1;  // the chunk is enqueued in full: the next one reads the contexts this one wrote
    return AFTER_OK;
}

// ====================================================================== ECAPA
struct TdnnW {
    ConvW conv;
    Affine bn;
    int dil = 1;
    ConvDmaPlanIn in;  // time-major conv path (conv_tm.hip)
    ConvTmPlan plan;
    float* wd = nullptr;
};
struct SeResW {
    TdnnW tdnn1, tdnn2;
    std::vector<TdnnW> res;
    ConvW se1, se2, shortcut;
    bool has_shortcut = false;
    TdnnW shortcut_t;
};
struct after_ecapa {
    after_ecapa_cfg cfg;
    int max_batch, max_T;
    Arena wa, ws;
    TdnnW first, mfa, asp_tdnn;
    std::vector<SeResW> blocks;
    ConvW asp_conv, fc;
    float* asp_w23 = nullptr;  // asp.tdnn weight columns for the (mean, std) context [A, 2C]
    Affine asp_bn;
    // workspaces
    float *feat = nullptr, *cat = nullptr, *t0 = nullptr, *t1 = nullptr, *t2 = nullptr, *t3 = nullptr;
    float *vecA = nullptr, *vecB = nullptr, *vecC = nullptr;
    float* xp[2] = {nullptr, nullptr};  // haloed conv inputs [B][rows(T)][Cp]
    size_t xp_elems = 0;
    TdnnW asp_conv_t;                   // asp.conv / shortcuts as plain convs on the same path
};

namespace {

int load_tdnn(Arena& a, WCursor& c, TdnnW& t, int cin, int cout, int k, int dil) {
    t.dil = dil;
    AFTER_TRY(load_conv(a, c, t.conv, cin, cout, k, false));
    return load_bn(a, c, t.bn, cout);
}

// geometry + GEMM operand of one conv for the time-major path (reflect / zero 'same' padding:
// pl = (k - 1) dil / 2, ecapa_encoder.py:74-78)
int plan_tdnn(Arena& a, TdnnW& t) {
    memset(&t.in, 0, sizeof(t.in));
    t.in.Cin = t.conv.cin;
    t.in.Cout = t.conv.cout;
    t.in.taps = t.conv.k;
    t.in.phases = 1;
    t.in.istride = 1;
    t.in.ostride = 1;
    const int pl = ((t.conv.k - 1) * t.dil) / 2;
    for (int i = 0; i < t.conv.k; ++i) t.in.toff[0][i] = i * t.dil - pl;
    conv_tm_plan(t.in, &t.plan);
    AFTER_REQUIRE(t.plan.ok, AFTER_E_INVALID, "ecapa: tap pattern outside the conv path");
    t.wd = a.take<float>(t.plan.w_floats);
    AFTER_REQUIRE(t.wd, AFTER_E_NOMEM, "ecapa: weight arena exhausted");
    return conv_tm_repack(t.conv.w, t.wd, t.in, t.plan, 0);
}

// TDNNBlock (ecapa_encoder.py:85-139): BN(ReLU(conv_reflect(x))) as ONE conv launch.  The input is
// either a haloed buffer (mirrored rows already in place) or, for k = 1, a raw row-pitched view.
struct TdnnIo {
    const float* src = nullptr;  // haloed [B][rows(T)][Cp]   (raw_ld == 0)
    int raw_ld = 0;              // > 0: src is raw [B][T][raw_ld], read in place (k = 1)
    float* y = nullptr;          // raw output view [B][T][y_ld] at column y_coff, or nullptr
    int y_ld = 0, y_coff = 0;
    const float* bias = nullptr; // override (per-clip with bias_bs > 0)
    int bias_bs = 0;
    bool plain = false;          // no ReLU / BatchNorm (shortcut, asp.conv)
    // second output: the next conv's haloed input = act2(o [+ add]) for channels [clo, chi)
    float* y2 = nullptr;
    int y2_clo = 0, y2_chi = 0, y2_act = ACT_NONE, y2_reflect = 0;
    const float* y2_add = nullptr;
    int y2_add_ld = 0, y2_add_coff = 0;
};

int run_tdnn(const TdnnW& t, hipStream_t s, const TdnnIo& io, int B, int T) {
    ConvTmRun r;
    memset(&r, 0, sizeof(r));
    r.w = t.wd;
    r.bias = io.bias ? io.bias : t.conv.bias;
    r.bias_bstride = io.bias_bs;
    r.y = io.y;
    r.y_ld = io.y_ld;
    r.y_coff = io.y_coff;
    r.y_bs = io.y ? (long long)T * io.y_ld : 0;
    r.G = 1;
    r.B = B;
    r.Tout = T;
    r.Nn = T;
    if (io.raw_ld > 0) {
        AFTER_REQUIRE(t.conv.k == 1 && (t.conv.cin & 31) == 0 && (io.raw_ld & 3) == 0, AFTER_E_INVALID,
                      "ecapa: in-place input needs k = 1 and 32 | Cin");
        r.xp = io.src - (size_t)conv_tm_halo() * io.raw_ld;
        r.x_ld = io.raw_ld;
        r.x_bs = (long long)T * io.raw_ld;
        r.Tp = T;
    } else {
        r.xp = io.src;
        r.Tp = conv_tm_rows(T);
    }
    if (!io.plain) {
        r.out_act = ACT_RELU;
        r.post_scale = t.bn.scale;
        r.post_shift = t.bn.shift;
    }
    if (io.y2) {
        r.y2 = io.y2;
        r.y2_clo = io.y2_clo;
        r.y2_chi = io.y2_chi;
        r.y2_act = io.y2_act;
        r.y2_reflect = io.y2_reflect;
        r.y2_add = io.y2_add;
        r.y2_add_ld = io.y2_add_ld;
        r.y2_add_coff = io.y2_add_coff;
    }
    return launch_conv_tm(r, t.in, t.plan, s);
}

int gemm_rows(hipStream_t s, const float* A, int lda, const float* W, int ldw, const float* bias,
              float* C, int ldc, int M, int N, int K, int epi) {
    if ((long long)M * N <= 16384 && (K & 3) == 0 && (lda & 3) == 0 && (ldw & 3) == 0 &&
        (epi == EPI_NONE || epi == EPI_RELU || epi == EPI_SIGMOID)) {
        // a few rows x a few hundred outputs: one wave per output beats any tile (the MFMA GEMM needs
        // 16 rows and pays ~8-16 us for M = 1)
        hipLaunchKernelGGL(rowvec_gemm_kernel, dim3(cdiv(M * N, 4)), dim3(256), 0, s, A, lda, W, ldw, bias, C, ldc,
                           M, N, K, epi);
        AFTER_HIP_CHECK(hipGetLastError());
        return AFTER_OK;
    }
    GemmArgs g{A, lda, W, ldw, bias, nullptr, 0, C, ldc, M, N, K, epi};
    return launch_gemm(g, s);
}

}  // namespace

extern "C" int after_ecapa_create(const after_ecapa_cfg* cfg, const float* const* weights,
                                  int n_weights, int max_batch, int max_T, after_ecapa** out) {
    AFTER_REQUIRE(cfg && weights && out, AFTER_E_INVALID, "null argument");
    *out = nullptr;
    const int n = cfg->n_blocks;
    AFTER_REQUIRE(n >= 3 && n <= 8 && max_batch > 0 && max_T > 1, AFTER_E_INVALID,
                  "ecapa: bad configuration");
    const int scale = cfg->res2net_scale;
    AFTER_REQUIRE(scale >= 2 && scale <= 16, AFTER_E_INVALID, "ecapa: res2net_scale %d", scale);
    for (int i = 0; i < n; ++i)
        AFTER_REQUIRE(cfg->kernel_sizes[i] >= 1 && cfg->kernel_sizes[i] <= kMaxTaps &&
                          cfg->kernel_sizes[i] % 2 == 1 && cfg->channels[i] % 4 == 0,
                      AFTER_E_INVALID, "ecapa: unsupported kernel size / channels");
    int ccat = 0;
    for (int i = 1; i < n - 1; ++i) {
        AFTER_REQUIRE(cfg->channels[i] % scale == 0, AFTER_E_INVALID, "ecapa: channels %% scale");
        ccat += cfg->channels[i];
    }
    AFTER_REQUIRE(ccat == cfg->channels[n - 1], AFTER_E_INVALID,
                  "ecapa: channels[-1]=%d must equal the concatenated width %d",
                  cfg->channels[n - 1], ccat);
    after_ecapa* h = new (std::nothrow) after_ecapa();
    AFTER_REQUIRE(h, AFTER_E_NOMEM, "out of host memory");
    h->cfg = *cfg;
    h->max_batch = max_batch;
    h->max_T = max_T;
    auto fail = [&](int rc) {
        after_ecapa_destroy(h);
        return rc;
    };
    const int CL = cfg->channels[n - 1], A = cfg->attention_channels, SE = cfg->se_channels;
    size_t wf = conv_floats(cfg->in_size, cfg->channels[0], cfg->kernel_sizes[0]) + 4096;
    int cmaxc = cfg->channels[0];
    for (int i = 1; i < n - 1; ++i) {
        const int ci = cfg->channels[i - 1], co = cfg->channels[i];
        wf += conv_floats(ci, co, 1) + conv_floats(co, co, 1) +
              (scale - 1) * conv_floats(co / scale, co / scale, cfg->kernel_sizes[i]) +
              conv_floats(co, SE, 1) + conv_floats(SE, co, 1) + conv_floats(ci, co, 1) +
              (size_t)(scale + 4) * 2 * co + 8192;
        cmaxc = co > cmaxc ? co : cmaxc;
    }
    wf += conv_floats(CL, CL, cfg->kernel_sizes[n - 1]) + conv_floats(3 * CL, A, 1) +
          (size_t)A * 2 * CL + conv_floats(A, CL, 1) + conv_floats(2 * CL, cfg->out_dim, 1) +
          12 * (size_t)CL + 8192;
    int rc = h->wa.init(wf * 4 * sizeof(float) + (1 << 20));
    if (rc) return fail(rc);
    WCursor cur{weights, n_weights};
    if ((rc = load_tdnn(h->wa, cur, h->first, cfg->in_size, cfg->channels[0], cfg->kernel_sizes[0],
                        cfg->dilations[0])))
        return fail(rc);
    h->blocks.resize(n - 2);
    for (int i = 1; i < n - 1; ++i) {
        SeResW& b = h->blocks[i - 1];
        const int ci = cfg->channels[i - 1], co = cfg->channels[i], cs = co / scale;
        if ((rc = load_tdnn(h->wa, cur, b.tdnn1, ci, co, 1, 1))) return fail(rc);
        b.res.resize(scale - 1);
        for (int j = 0; j < scale - 1; ++j)
            if ((rc = load_tdnn(h->wa, cur, b.res[j], cs, cs, cfg->kernel_sizes[i], cfg->dilations[i])))
                return fail(rc);
        if ((rc = load_tdnn(h->wa, cur, b.tdnn2, co, co, 1, 1))) return fail(rc);
        if ((rc = load_conv(h->wa, cur, b.se1, co, SE, 1, false))) return fail(rc);
        if ((rc = load_conv(h->wa, cur, b.se2, SE, co, 1, false))) return fail(rc);
        b.has_shortcut = ci != co;
        if (b.has_shortcut && (rc = load_conv(h->wa, cur, b.shortcut, ci, co, 1, false))) return fail(rc);
    }
    if ((rc = load_tdnn(h->wa, cur, h->mfa, CL, CL, cfg->kernel_sizes[n - 1], cfg->dilations[n - 1])))
        return fail(rc);
    {
        // asp.tdnn: Conv1d(3 CL -> A, k = 1).  The (mean, std) context is constant over
        // time (ecapa_encoder.py AttentiveStatisticsPooling.forward), so its 2 CL weight
        // columns become a per-clip bias: split [A, 3 CL] into [A, CL] (conv) + [A, 2 CL].
        const float* w = cur.next();
        const float* b = cur.next();
        if (!cur.ok) {
            set_error("ecapa: missing asp.tdnn tensors");
            return fail(AFTER_E_INVALID);
        }
        float* wz = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&wz), (size_t)A * CL * sizeof(float)) != hipSuccess)
            return fail(AFTER_E_NOMEM);
        h->asp_w23 = h->wa.take<float>((size_t)A * 2 * CL);
        h->asp_tdnn.conv.bias = h->wa.take<float>(A);
        h->asp_tdnn.conv.w = h->wa.take<float>((size_t)A * pad16(CL));
        if (!h->asp_w23 || !h->asp_tdnn.conv.bias || !h->asp_tdnn.conv.w) {
            (void)hipFree(wz);
            return fail(AFTER_E_NOMEM);
        }
        bool okc = hipMemcpy2D(wz, (size_t)CL * 4, w, (size_t)3 * CL * 4, (size_t)CL * 4, A,
                               hipMemcpyDeviceToDevice) == hipSuccess &&
                   hipMemcpy2D(h->asp_w23, (size_t)2 * CL * 4, w + CL, (size_t)3 * CL * 4,
                               (size_t)2 * CL * 4, A, hipMemcpyDeviceToDevice) == hipSuccess &&
                   hipMemcpy(h->asp_tdnn.conv.bias, b, A * sizeof(float), hipMemcpyDeviceToDevice) ==
                       hipSuccess;
        rc = okc ? pack_conv_weight(wz, nullptr, h->asp_tdnn.conv.w, A, CL, 1, pad16(CL), 0)
                 : AFTER_E_HIP;
        (void)hipFree(wz);
        if (rc) return fail(rc);
        h->asp_tdnn.conv.cin = CL;
        h->asp_tdnn.conv.cout = A;
        h->asp_tdnn.conv.k = 1;
        if ((rc = load_bn(h->wa, cur, h->asp_tdnn.bn, A))) return fail(rc);
    }
    if ((rc = load_conv(h->wa, cur, h->asp_conv, A, CL, 1, false))) return fail(rc);
    if ((rc = load_bn(h->wa, cur, h->asp_bn, 2 * CL))) return fail(rc);
    {
        // fc as a plain [out_dim, 2 CL] matrix for the GEMM (K contiguous already)
        const float* w = cur.next();
        const float* b = cur.next();
        if (!cur.ok) {
            set_error("ecapa: missing fc tensors");
            return fail(AFTER_E_INVALID);
        }
        h->fc.w = h->wa.take<float>((size_t)cfg->out_dim * 2 * CL);
        h->fc.bias = h->wa.take<float>(cfg->out_dim);
        if (!h->fc.w || !h->fc.bias) return fail(AFTER_E_NOMEM);
        if (hipMemcpy(h->fc.w, w, (size_t)cfg->out_dim * 2 * CL * 4, hipMemcpyDeviceToDevice) != hipSuccess ||
            hipMemcpy(h->fc.bias, b, cfg->out_dim * 4, hipMemcpyDeviceToDevice) != hipSuccess)
            return fail(AFTER_E_HIP);
    }
    if (!cur.ok || cur.i != n_weights) {
        set_error("ecapa: expected %d weight tensors, got %d", cur.i, n_weights);
        return fail(AFTER_E_INVALID);
    }
    // ---- time-major conv path: GEMM operands of every conv
    if ((rc = plan_tdnn(h->wa, h->first))) return fail(rc);
    for (auto& b : h->blocks) {
        if ((rc = plan_tdnn(h->wa, b.tdnn1)) || (rc = plan_tdnn(h->wa, b.tdnn2))) return fail(rc);
        for (auto& t : b.res)
            if ((rc = plan_tdnn(h->wa, t))) return fail(rc);
        if (b.has_shortcut) {
            b.shortcut_t.conv = b.shortcut;
            if ((rc = plan_tdnn(h->wa, b.shortcut_t))) return fail(rc);
        }
    }
    if ((rc = plan_tdnn(h->wa, h->mfa)) || (rc = plan_tdnn(h->wa, h->asp_tdnn))) return fail(rc);
    h->asp_conv_t.conv = h->asp_conv;
    if ((rc = plan_tdnn(h->wa, h->asp_conv_t))) return fail(rc);
    // SE convs are used as GEMM weights [N, K] (k = 1): keep unpacked copies
    const size_t el = (size_t)max_batch * cmaxc * max_T;
    const size_t elcat = (size_t)max_batch * CL * max_T;
    int cpmax = conv_tm_cp(CL > cfg->in_size ? CL : cfg->in_size);
    cpmax = conv_tm_cp(cmaxc) > cpmax ? conv_tm_cp(cmaxc) : cpmax;
    h->xp_elems = (size_t)max_batch * conv_tm_rows(max_T) * cpmax + 4096;
    if ((rc = h->ws.init((3 * el + 3 * elcat + 2 * h->xp_elems) * sizeof(float) +
                         3 * (size_t)max_batch * (3 * CL + 1024) * sizeof(float) + 16384)))
        return fail(rc);
    h->feat = h->ws.take<float>(el);
    h->t0 = h->ws.take<float>(el);
    h->t1 = h->ws.take<float>(el);
    h->cat = h->ws.take<float>(elcat);
    h->t2 = h->ws.take<float>(elcat);
    h->t3 = h->ws.take<float>(elcat);
    h->vecA = h->ws.take<float>((size_t)max_batch * (3 * CL + 1024));
    h->vecB = h->ws.take<float>((size_t)max_batch * (3 * CL + 1024));
    h->vecC = h->ws.take<float>((size_t)max_batch * (3 * CL + 1024));
    h->xp[0] = h->ws.take<float>(h->xp_elems);
    h->xp[1] = h->ws.take<float>(h->xp_elems);
    if (!h->t3 || !h->vecC || !h->xp[1]) return fail(AFTER_E_NOMEM);
    if (hipDeviceSynchronize() != hipSuccess) return fail(AFTER_E_HIP);
    *out = h;
    return AFTER_OK;
}

extern "C" void after_ecapa_destroy(after_ecapa* h) {
    if (!h) return;
    h->wa.release();
    h->ws.release();
    delete h;
}

extern "C" int after_ecapa_forward(after_ecapa* h, const float* z, float* out, int B, int T,
                                   void* stream) {
    AFTER_REQUIRE(h && z && out, AFTER_E_INVALID, "null argument");
    AFTER_REQUIRE(B > 0 && T > 1, AFTER_E_INVALID, "ecapa: need T >= 2 (reflect padding)");
    AFTER_REQUIRE(B <= h->max_batch && T <= h->max_T, AFTER_E_CAPACITY,
                  "B=%d T=%d exceed max_batch=%d max_T=%d", B, T, h->max_batch, h->max_T);
    hipStream_t s = (hipStream_t)stream;
    const after_ecapa_cfg& c = h->cfg;
    const int n = c.n_blocks, scale = c.res2net_scale, CL = c.channels[n - 1];
    const int A = c.attention_channels, SE = c.se_channels;
    auto reflect_pad = [](const TdnnW& t) { return ((t.conv.k - 1) * t.dil) / 2; };
    auto halo_copy = [&](const float* x, int x_cm, int ldx, int C, float* dst, int reflect,
                         const float* x2 = nullptr, int ldx2 = 0, int act = ACT_NONE) -> int {
        ActPadTm p;
        memset(&p, 0, sizeof(p));
        p.x = x;
        p.x2 = x2;
        p.ldx2 = ldx2;
        p.y = dst;
        p.act = act;
        p.B = B;
        p.C = C;
        p.T = T;
        p.G = 1;
        p.x_cm = x_cm;
        p.ldx = ldx;
        p.pad_reflect = reflect;
        AFTER_REQUIRE(reflect < T, AFTER_E_INVALID, "ecapa: T=%d too short for the reflect padding", T);
        return launch_act_pad_tm(p, s);
    };
    // All activations are time-major [B][T][C] from here on (channel slices = column ranges).
    // blocks.0: TDNN(in -> C0) on the caller's [B][C][T] tensor
    AFTER_TRY(halo_copy(z, 1, 0, c.in_size, h->xp[0], 1));
    {
        TdnnIo io;
        io.src = h->xp[0];
        io.y = h->feat;
        io.y_ld = c.channels[0];
        AFTER_TRY(run_tdnn(h->first, s, io, B, T));
    }
    const float* xin = h->feat;  // view [B][T][xin_ld] starting at the block's first channel
    int xin_ld = c.channels[0];
    int cat_off = 0;
    for (int i = 1; i < n - 1; ++i) {
        const SeResW& b = h->blocks[i - 1];
        const int co = c.channels[i], cs = co / scale;
        AFTER_REQUIRE((c.channels[i - 1] & 31) == 0 && (co & 31) == 0, AFTER_E_INVALID,
                      "ecapa: channel widths must be multiples of 32");
        // a conv epilogue can write the next conv's haloed input itself when the slice fills whole
        // 32-deep K slabs (every shipped width); narrower test configurations take one act_pad more
        const bool fuse = (cs & 31) == 0;
        // tdnn1 (k = 1) on the raw view -> t0 [B][T][co]; its epilogue also lays out x_1 (channels
        // [cs, 2 cs)) with mirrored halo rows for the first Res2Net conv
        {
            TdnnIo io;
            io.src = xin;
            io.raw_ld = xin_ld;
            io.y = h->t0;
            io.y_ld = co;
            if (fuse) {
                io.y2 = h->xp[0];
                io.y2_clo = cs;
                io.y2_chi = 2 * cs;
                io.y2_reflect = reflect_pad(b.res[0]);
            }
            AFTER_TRY(run_tdnn(b.tdnn1, s, io, B, T));
            if (!fuse) AFTER_TRY(halo_copy(h->t0 + cs, 0, co, cs, h->xp[0], reflect_pad(b.res[0])));
        }
        // Res2Net chain -> t1 [B][T][co]: y_0 = x_0 ; y_1 = f_0(x_1) ; y_j+1 = f_j(x_j+1 + y_j)
        {
            const size_t tot4 = (size_t)B * T * cs / 4;
            hipLaunchKernelGGL(copy_slice_tm_kernel, dim3((unsigned)cdivll(tot4, 256)), dim3(256), 0, s, h->t0,
                               co, h->t1, co, cs, tot4);
            AFTER_HIP_CHECK(hipGetLastError());
        }
        for (int j = 0; j < scale - 1; ++j) {
            TdnnIo io;
            io.src = h->xp[j & 1];
            io.y = h->t1;
            io.y_ld = co;
            io.y_coff = (j + 1) * cs;
            const bool more = j + 1 < scale - 1;
            if (more && fuse) {  // the next conv's input: this output + the next slice of tdnn1's
                io.y2 = h->xp[(j + 1) & 1];
                io.y2_chi = cs;
                io.y2_reflect = reflect_pad(b.res[j + 1]);
                io.y2_add = h->t0;
                io.y2_add_ld = co;
                io.y2_add_coff = (j + 2) * cs;
            }
            AFTER_TRY(run_tdnn(b.res[j], s, io, B, T));
            if (more && !fuse)
                AFTER_TRY(halo_copy(h->t0 + (j + 2) * cs, 0, co, cs, h->xp[(j + 1) & 1], reflect_pad(b.res[j + 1]),
                                    h->t1 + (j + 1) * cs, co));
        }
        // tdnn2 (k = 1) on t1 -> t0
        {
            TdnnIo io;
            io.src = h->t1;
            io.raw_ld = co;
            io.y = h->t0;
            io.y_ld = co;
            AFTER_TRY(run_tdnn(b.tdnn2, s, io, B, T));
        }
        // SE: s = sigmoid(W2 relu(W1 mean_t(x) + b1) + b2)
        hipLaunchKernelGGL(time_stats_tm_kernel, dim3(cdiv(co, 64), B), dim3(256), 0, s, h->t0, co,
                           (const float*)nullptr, 0, h->vecA, (const float*)nullptr, (const float*)nullptr, co,
                           T, 1);
        AFTER_HIP_CHECK(hipGetLastError());
        AFTER_TRY(gemm_rows(s, h->vecA, co, b.se1.w, pad16(co), b.se1.bias, h->vecB, SE, B, SE, co,
                            EPI_RELU));
        AFTER_TRY(gemm_rows(s, h->vecB, SE, b.se2.w, pad16(SE), b.se2.bias, h->vecC, co, B, co, SE,
                            EPI_SIGMOID));
        // residual (identity or 1x1 shortcut), then out = s * x + residual -> cat slice
        const float* res = xin;
        int res_ld = xin_ld;
        if (b.has_shortcut) {
            TdnnIo io;
            io.src = xin;
            io.raw_ld = xin_ld;
            io.y = h->t1;
            io.y_ld = co;
            io.plain = true;
            AFTER_TRY(run_tdnn(b.shortcut_t, s, io, B, T));
            res = h->t1;
            res_ld = co;
        }
        {
            const size_t tot4 = (size_t)B * T * co / 4;
            hipLaunchKernelGGL(se_scale_add_tm_kernel, dim3((unsigned)cdivll(tot4, 256)), dim3(256), 0, s, h->t0,
                               co, h->vecC, res, res_ld, h->cat + cat_off, CL, co, T, tot4);
            AFTER_HIP_CHECK(hipGetLastError());
        }
        xin = h->cat + cat_off;
        xin_ld = CL;
        cat_off += co;
    }
    // mfa over the concatenation -> t2 [B][T][CL]
    AFTER_TRY(halo_copy(h->cat, 0, CL, CL, h->xp[0], reflect_pad(h->mfa)));
    {
        TdnnIo io;
        io.src = h->xp[0];
        io.y = h->t2;
        io.y_ld = CL;
        AFTER_TRY(run_tdnn(h->mfa, s, io, B, T));
    }
    // attentive statistics pooling with global context
    hipLaunchKernelGGL(time_stats_tm_kernel, dim3(cdiv(CL, 64), B), dim3(256), 0, s, h->t2, CL,
                       (const float*)nullptr, 0, h->vecA, (const float*)nullptr, (const float*)nullptr, CL, T, 0);
    AFTER_HIP_CHECK(hipGetLastError());
    // per-clip bias = W[:, CL:3CL] [mean | std] + b
    AFTER_TRY(gemm_rows(s, h->vecA, 2 * CL, h->asp_w23, 2 * CL, h->asp_tdnn.conv.bias, h->vecB, A, B, A,
                        2 * CL, EPI_NONE));
    // asp.tdnn (k = 1, per-clip bias) -> tanh -> the haloed input of asp.conv, no raw copy needed
    {
        TdnnIo io;
        io.src = h->t2;
        io.raw_ld = CL;
        io.bias = h->vecB;
        io.bias_bs = A;
        if ((A & 31) == 0) {
            io.y2 = h->xp[1];
            io.y2_chi = A;
            io.y2_act = ACT_TANH;
        } else {
            io.y = h->t0;
            io.y_ld = A;
        }
        AFTER_TRY(run_tdnn(h->asp_tdnn, s, io, B, T));
        if (io.y) AFTER_TRY(halo_copy(h->t0, 0, A, A, h->xp[1], 0, nullptr, 0, ACT_TANH));
    }
    float* logits = h->t3;
    {
        TdnnIo io;  // conv 1x1: A -> CL
        io.src = h->xp[1];
        io.y = logits;
        io.y_ld = CL;
        io.plain = true;
        AFTER_TRY(run_tdnn(h->asp_conv_t, s, io, B, T));
    }
    hipLaunchKernelGGL(time_stats_tm_kernel, dim3(cdiv(CL, 64), B), dim3(256), 0, s, h->t2, CL, logits, CL,
                       h->vecA, h->asp_bn.scale, h->asp_bn.shift, CL, T, 0);
    AFTER_HIP_CHECK(hipGetLastError());
    AFTER_TRY(gemm_rows(s, h->vecA, 2 * CL, h->fc.w, 2 * CL, h->fc.bias, out, c.out_dim, B, c.out_dim,
                        2 * CL, EPI_NONE));
    if (c.use_tanh) {
        const int tot = B * c.out_dim;
        hipLaunchKernelGGL(tanh_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, s, out, tot);
        AFTER_HIP_CHECK(hipGetLastError());
    }
    return AFTER_OK;
}
