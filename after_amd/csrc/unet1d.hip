// UNET1D: the Conv1d / GroupNorm / SiLU / FiLM denoiser (reference
// after/diffusion/networks/unet1d.py:254-429, blocks :29-252) on gfx950.
//
// Not selected by any shipped gin config (all four use DenoiserV2); built from SURVEY 8(f)-4 on
// the time-major conv path of conv_tm.hip: activations [B][T][C], channel concatenation = column
// ranges of one row-pitched buffer, GroupNorm statistics by a pass over the concatenated tensor
// (or by the producing conv's epilogue), GroupNorm-apply + SiLU + zero halo in act_pad_tm, the
// conv as the balanced LDS-DMA GEMM, FiLM (time and cond modulation, unet1d.py:100-110) as a
// per-(clip, channel) affine in the conv epilogue, strided pools via the row stride of the A
// operand, nearest upsampling as a row copy.
// Supported: time_cond_channels > 0, cond_channels > 0; n_attn_layers > 0 (round 4): SelfAttention1d (blocks.py:201-243) as
// GroupNorm(1, C) statistics pass -> 1 x 1 qkv conv (the same conv path) -> unet_attn_kernel (full softmax attention per
// head, one lane per query, keys staged through LDS) -> 1 x 1 out_proj conv with the block output as residual.
#include <new>
#include <vector>

#include "conv.h"

namespace after {
namespace {

// SPE (unet1d.py:7-24): cat[sin(w x), cos(w x)], x = 32 t, w_i = (1/10000)^(2 i / dim)
__global__ void spe_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim,
                           float max_positions, float scale) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int half = dim / 2;
    if (idx >= B * half) return;
    const int b = idx / half, i = idx - b * half;
    const float x = t[b] * scale;
    const float w = powf(1.0f / max_positions, (2.0f * (float)i) / (float)dim);
    out[(size_t)b * dim + i] = sinf(w * x);
    out[(size_t)b * dim + half + i] = cosf(w * x);
}

// FiLM parameters of one ConvBlock1D (unet1d.py:100-110): for clip b
//   [tm | ta] = W2t silu(W1t emb_t + b1t) + b2t,  [cm | ca] = W2c silu(W1c cond + b1c) + b2c
//   (conv + bias) * tm + ta, then * cm + ca   ==   (conv + bias) * ps + pt
//   ps = tm cm,  pt = ta cm + ca.                 One workgroup per clip.
struct FilmArgs {
    const float *emb, *cond;             // [B, TC], [B, CC]
    const float *w1t, *b1t, *w2t, *b2t;  // [128, TC], [128], [2C, 128], [2C]
    const float *w1c, *b1c, *w2c, *b2c;  // [128, CC], ...
    float *ps, *pt;                      // [B, C]
    int TC, CC, C, H;                    // H = 128
};
__global__ __launch_bounds__(256) void film_kernel(FilmArgs a) {
    __shared__ float ht[256], hc[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < a.H; j += 256) {
        float st = a.b1t[j], sc = a.b1c[j];
        for (int k = 0; k < a.TC; ++k) st += a.w1t[(size_t)j * a.TC + k] * a.emb[(size_t)b * a.TC + k];
        for (int k = 0; k < a.CC; ++k) sc += a.w1c[(size_t)j * a.CC + k] * a.cond[(size_t)b * a.CC + k];
        ht[j] = st / (1.0f + expf(-st));
        hc[j] = sc / (1.0f + expf(-sc));
    }
    __syncthreads();
    for (int c = tid; c < a.C; c += 256) {
        float tm = a.b2t[c], ta = a.b2t[a.C + c], cm = a.b2c[c], ca = a.b2c[a.C + c];
        for (int k = 0; k < a.H; ++k) {
            tm += a.w2t[(size_t)c * a.H + k] * ht[k];
            ta += a.w2t[(size_t)(a.C + c) * a.H + k] * ht[k];
            cm += a.w2c[(size_t)c * a.H + k] * hc[k];
            ca += a.w2c[(size_t)(a.C + c) * a.H + k] * hc[k];
        }
        a.ps[(size_t)b * a.C + c] = tm * cm;
        a.pt[(size_t)b * a.C + c] = ta * cm + ca;
    }
}

// ---- SelfAttention1d's core (blocks.py:231-241): qkv [B][T][3 C] time-major, head h = channels [h D, (h + 1) D) of each third;
// att = softmax((q s) (k s)^T), s = D^-1/4 on both operands as in the reference; y [B][T][C].  One lane per query row (online
// softmax over key tiles of 64 staged in LDS, q and the output row in registers); grid (T / 64, heads, B).  Not a hot kernel:
// no shipped config selects UNET1D, and its attention runs at 1/4 .. 1/32 of the clip's frames.
template <int D>
__global__ __launch_bounds__(64) void unet_attn_kernel(const float* __restrict__ qkv, float* __restrict__ y, int T, int C) {
    __shared__ float ks[64][D + 1], vs[64][D + 1];
    const int t = blockIdx.x * 64 + threadIdx.x, hd = blockIdx.y, b = blockIdx.z;
    const float scale = powf((float)D, -0.25f);
    const float* base = qkv + (size_t)b * T * 3 * C;
    float q[D], o[D];
    const int tq = t < T ? t : T - 1;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        q[j] = base[(size_t)tq * 3 * C + hd * D + j] * scale;
        o[j] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < T; k0 += 64) {
        const int kr = k0 + threadIdx.x;
        __syncthreads();
        if (kr < T) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                ks[threadIdx.x][j] = base[(size_t)kr * 3 * C + C + hd * D + j] * scale;
                vs[threadIdx.x][j] = base[(size_t)kr * 3 * C + 2 * C + hd * D + j];
            }
        }
        __syncthreads();
        const int nk = min(64, T - k0);
        for (int kk = 0; kk < nk; ++kk) {
            float sc = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) sc += q[j] * ks[kk][j];
            const float mn = fmaxf(m, sc);
            const float corr = expf(m - mn), pw = expf(sc - mn);
            l = l * corr + pw;
#pragma unroll
            for (int j = 0; j < D; ++j) o[j] = o[j] * corr + pw * vs[kk][j];
            m = mn;
        }
    }
    if (t < T) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int j = 0; j < D; ++j) y[((size_t)b * T + t) * C + hd * D + j] = o[j] * inv;
    }
}

// ---- classifier-free guidance around the network (model.py:721-785): everything on the device.
// 3x batch of one evaluation: rows r = part * B + b;  x3 = x repeated; t3 = the step's time;
// cond3 / tc3 = the conditioning or `drop` per the CFG arrangement (mode 2 = export_midi.py:329-358).
__global__ __launch_bounds__(256) void cfg3_cond_kernel(const float* __restrict__ cond,
                                                        const float* __restrict__ tc,
                                                        float* __restrict__ cond3, float* __restrict__ tc3,
                                                        int B, int CC, size_t tc_per, float drop, int midi) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n_c = (size_t)3 * B * CC, n_t = (size_t)3 * B * tc_per;
    if (idx < n_c) {
        const int r = idx / CC, b = r % B, part = r / B;
        const bool keep = midi ? part <= 1 : part == 0;
        cond3[idx] = keep ? cond[(size_t)b * CC + idx % CC] : drop;
    }
    if (idx < n_t) {
        const size_t r = idx / tc_per;
        const int b = r % B, part = r / B;
        const bool keep = midi ? part == 0 : part <= 1;
        tc3[idx] = keep ? tc[(size_t)b * tc_per + idx % tc_per] : drop;
    }
}

// x3[r] = x[r % B];  t3[r] = time[r % B] (forward) or torch.linspace(0, 1, N + 1)[step] (sampler,
// at::linspace's fp32 formula as in denoiser.hip / tests/test_boundary_cpu.py)
__global__ __launch_bounds__(256) void cfg3_x_kernel(const float* __restrict__ x, float* __restrict__ x3,
                                                     const float* __restrict__ time, float* __restrict__ t3,
                                                     int B, size_t per, int step, int nb_steps) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < (size_t)3 * B * per) x3[idx] = x[idx % ((size_t)B * per)];
    if (idx < (size_t)3 * B) {
        float t;
        if (time) {
            t = time[idx % B];
        } else {
            const int pts = nb_steps + 1;
            const float st = __fdiv_rn(1.0f, (float)(pts - 1));
            t = (step < pts / 2) ? __fmul_rn(st, (float)step) : __fmaf_rn(-st, (float)(pts - step - 1), 1.0f);
        }
        t3[idx] = t;
    }
}

// out = xin * keep + dt * (d_none + total (d_mid + factor (d_full - d_mid) - d_none))
// (keep = 1, dt = 1/N: Euler step; keep = 0, dt = 1: model_forward)
__global__ __launch_bounds__(256) void cfg3_combine_kernel(const float* __restrict__ d3,
                                                           const float* __restrict__ xin,
                                                           float* __restrict__ out, size_t n, float total,
                                                           float factor, float dt, float keep) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const float df = d3[idx], dm = d3[n + idx], dn = d3[2 * n + idx];
    const float v = dn + total * (dm + factor * (df - dm) - dn);
    out[idx] = (xin ? xin[idx] * keep : 0.f) + v * dt;
}

struct WCur {
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
    const float* opt() {  // may legitimately be null
        if (i >= n) {
            ok = false;
            return nullptr;
        }
        return w[i++];
    }
};

struct PackedConv {
    float *w = nullptr, *bias = nullptr;
    int cin = 0, cout = 0, k = 1;
    ConvDmaPlanIn in;   // geometry + GEMM operand for conv_tm.hip
    ConvTmPlan plan;
    float* wd = nullptr;
};
struct AttnW {  // SelfAttention1d (blocks.py:201-243)
    PackedConv qkv, out;
    float *nw = nullptr, *nb = nullptr;  // GroupNorm(1, C) affine
    int C = 0, nh = 0;
    bool on = false;
};
struct BlockW {  // ConvBlock1D
    PackedConv c1, c2, to_out;
    float *gn1_w = nullptr, *gn1_b = nullptr, *gn2_w = nullptr, *gn2_b = nullptr;
    float *w1t, *b1t, *w2t, *b2t, *w1c, *b1c, *w2c, *b2c;
    int in_c = 0, skip_c = 0, tc_c = 0, out_c = 0;
    bool res = true, has_to_out = false;
};

}  // namespace
}  // namespace after

using namespace after;

struct after_unet1d {
    after_unet1d_cfg cfg;
    int max_batch, max_T, n;
    Arena wa, ws;
    std::vector<PackedConv> cond_emb;  // n + 1
    std::vector<BlockW> down, up;      // n each
    std::vector<PackedConv> pool, upconv;
    std::vector<char> up_has_conv;
    BlockW mid;
    std::vector<AttnW> down_attn, up_attn;  // n each (`on` where unet1d.py:339 / :350 switch them on)
    AttnW mid_attn;
    float *attx = nullptr, *qkvb = nullptr, *attb = nullptr;  // a block's output in front of its attention, qkv [B][T][3C], att @ v
    // workspaces
    float *emb = nullptr, *ps = nullptr, *pt = nullptr;
    double* stats = nullptr;  // [kSlots][conv_tm_stat_sub()][max_batch][16][2][kStatBins] GroupNorm accumulators (conv.h: stat_bins)
    int stat_slot = 0;
    float *cat = nullptr, *tmp = nullptr, *resb = nullptr, *xa = nullptr, *xb = nullptr, *ups = nullptr;
    float *xtm = nullptr;     // the network input in time-major form
    float* xp = nullptr;      // activated + haloed conv input
    size_t xp_elems = 0;
    std::vector<float*> skips, tconds;  // n each (+1 tcond for the middle block)
    int cmax = 0, ccat_max = 0;
    // CFG workspaces (3x batch): allocated with the handle when max_batch is a multiple of 3
    float *x3 = nullptr, *t3 = nullptr, *cond3 = nullptr, *tc3 = nullptr, *d3 = nullptr, *xs = nullptr;
};

namespace {

int dev_copy(Arena& a, float** dst, const float* src, size_t n) {
    *dst = a.take<float>(n);
    AFTER_REQUIRE(*dst, AFTER_E_NOMEM, "unet1d: weight arena exhausted");
    AFTER_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice));
    return AFTER_OK;
}

// geometry of a "same"-padded conv (odd k: pad k // 2 both sides) with an optional stride
int plan_pconv(Arena& a, PackedConv& p, int stride) {
    memset(&p.in, 0, sizeof(p.in));
    p.in.Cin = p.cin;
    p.in.Cout = p.cout;
    p.in.taps = p.k;
    p.in.phases = 1;
    p.in.istride = stride;
    p.in.ostride = 1;
    for (int t = 0; t < p.k; ++t) p.in.toff[0][t] = t - p.k / 2;
    conv_tm_plan(p.in, &p.plan);
    AFTER_REQUIRE(p.plan.ok, AFTER_E_INVALID, "unet1d: tap pattern outside the conv path");
    p.wd = a.take<float>(p.plan.w_floats);
    AFTER_REQUIRE(p.wd, AFTER_E_NOMEM, "unet1d: weight arena exhausted");
    return conv_tm_repack(p.w, p.wd, p.in, p.plan, 0);
}

int load_pconv(Arena& a, WCur& c, PackedConv& p, int cin, int cout, int k, int stride = 1) {
    p.cin = cin;
    p.cout = cout;
    p.k = k;
    const float* w = c.next();
    const float* b = c.next();
    AFTER_REQUIRE(c.ok, AFTER_E_INVALID, "unet1d: missing conv tensors");
    p.w = a.take<float>((size_t)cout * k * pad16(cin));
    AFTER_REQUIRE(p.w, AFTER_E_NOMEM, "unet1d: weight arena exhausted");
    AFTER_TRY(pack_conv_weight(w, nullptr, p.w, cout, cin, k, pad16(cin), 0));
    AFTER_TRY(dev_copy(a, &p.bias, b, cout));
    return plan_pconv(a, p, stride);
}

int gn_groups(int C) { return C / 4 < 16 ? C / 4 : 16; }  // unet1d.py:52-54,65

int load_block(Arena& a, WCur& c, BlockW& b, int in_c, int out_c, int skip_c, int tc_c, int k, int TC,
               int CC, bool res) {
    b.in_c = in_c;
    b.out_c = out_c;
    b.skip_c = skip_c;
    b.tc_c = tc_c;
    b.res = res;
    const int ccat = in_c + skip_c + tc_c;
    AFTER_REQUIRE(gn_groups(ccat) > 0 && ccat % gn_groups(ccat) == 0 && gn_groups(out_c) > 0 &&
                      out_c % gn_groups(out_c) == 0,
                  AFTER_E_INVALID,
                  "unet1d: GroupNorm(min(16, C/4), C) undefined for C=%d / %d (the reference "
                  "silently replaces it by Identity; not built)", ccat, out_c);
    AFTER_TRY(load_pconv(a, c, b.c1, ccat, out_c, k));
    const float *g1w = c.next(), *g1b = c.next();
    AFTER_REQUIRE(c.ok, AFTER_E_INVALID, "unet1d: missing gn1");
    AFTER_TRY(dev_copy(a, &b.gn1_w, g1w, ccat));
    AFTER_TRY(dev_copy(a, &b.gn1_b, g1b, ccat));
    AFTER_TRY(load_pconv(a, c, b.c2, out_c, out_c, k));
    const float *g2w = c.next(), *g2b = c.next();
    AFTER_REQUIRE(c.ok, AFTER_E_INVALID, "unet1d: missing gn2");
    AFTER_TRY(dev_copy(a, &b.gn2_w, g2w, out_c));
    AFTER_TRY(dev_copy(a, &b.gn2_b, g2b, out_c));
    const float* m[8];
    for (int i = 0; i < 8; ++i) m[i] = c.next();
    AFTER_REQUIRE(c.ok, AFTER_E_INVALID, "unet1d: missing FiLM MLP tensors");
    AFTER_TRY(dev_copy(a, &b.w1t, m[0], (size_t)128 * TC));
    AFTER_TRY(dev_copy(a, &b.b1t, m[1], 128));
    AFTER_TRY(dev_copy(a, &b.w2t, m[2], (size_t)2 * out_c * 128));
    AFTER_TRY(dev_copy(a, &b.b2t, m[3], 2 * out_c));
    AFTER_TRY(dev_copy(a, &b.w1c, m[4], (size_t)128 * CC));
    AFTER_TRY(dev_copy(a, &b.b1c, m[5], 128));
    AFTER_TRY(dev_copy(a, &b.w2c, m[6], (size_t)2 * out_c * 128));
    AFTER_TRY(dev_copy(a, &b.b2c, m[7], 2 * out_c));
    b.has_to_out = skip_c > 0;  // unet1d.py:78-81
    if (b.has_to_out) AFTER_TRY(load_pconv(a, c, b.to_out, in_c, out_c, 1));
    AFTER_REQUIRE(b.has_to_out || in_c == out_c || !res, AFTER_E_INVALID,
                  "unet1d: identity shortcut with in_c != out_c");
    return AFTER_OK;
}

int load_attn(Arena& a, WCur& c, AttnW& w, int C, int nh) {
    AFTER_REQUIRE(nh >= 1 && C % nh == 0, AFTER_E_INVALID, "unet1d: self-attention over %d channels with %d heads", C, nh);
    const int D = C / nh;
    AFTER_REQUIRE(D == 4 || D == 8 || D == 16 || D == 32 || D == 64, AFTER_E_INVALID,
                  "unet1d: self-attention head size %d (built: 4, 8, 16, 32, 64)", D);
    w.C = C;
    w.nh = nh;
    w.on = true;
    const float *nw = c.next(), *nb = c.next();
    AFTER_REQUIRE(c.ok, AFTER_E_INVALID, "unet1d: missing self_attn.norm");
    AFTER_TRY(dev_copy(a, &w.nw, nw, C));
    AFTER_TRY(dev_copy(a, &w.nb, nb, C));
    AFTER_TRY(load_pconv(a, c, w.qkv, C, 3 * C, 1));
    return load_pconv(a, c, w.out, C, C, 1);
}

size_t conv_fl(int cin, int cout, int k) {  // staging copy + GEMM operand + bias
    return (size_t)cout * k * (pad16(cin) + conv_tm_cp(cin)) + cout + 256;
}
size_t block_fl(int in_c, int out_c, int skip_c, int tc_c, int k, int TC, int CC) {
    const int ccat = in_c + skip_c + tc_c;
    return conv_fl(ccat, out_c, k) + conv_fl(out_c, out_c, k) + conv_fl(in_c, out_c, 1) + 2 * (size_t)ccat +
           2 * (size_t)out_c + 128 * (size_t)(TC + CC) + 256 + 4 * (size_t)out_c * 128 + 4 * (size_t)out_c + 2048;
}

constexpr int kSlots = 64;  // GroupNorm statistics slots per forward: 2 per ConvBlock1D (2 n + 1 blocks), 1 per self-attention

double* next_slot(after_unet1d* h) {
    double* p = h->stats + (size_t)(h->stat_slot % kSlots) * conv_tm_stat_sub() * h->max_batch * 16 * kStatWords;
    ++h->stat_slot;
    return p;
}

// One conv on time-major tensors: act(GroupNorm-affine(x)) into the haloed scratch, then the GEMM.
struct ConvIo {
    const float* x = nullptr;  // [B][T][x_ld] (x_cm: the caller's [B][C][T])
    int x_ld = 0, x_cm = 0;
    const double* stats = nullptr;  // GroupNorm accumulators of x (with gamma / beta), or nullptr
    const float *gamma = nullptr, *beta = nullptr;
    int G = 1, act = ACT_NONE, out_act = ACT_NONE;
    float* y = nullptr;  // [B][Tout][cout] (y_cm: [B][cout][Tout])
    int y_cm = 0;
    const float* res = nullptr;  // [B][Tout][res_ld]
    int res_ld = 0;
    const float *ps = nullptr, *pt = nullptr;  // FiLM: per-(clip, channel) scale / shift
    double* stats_out = nullptr;               // accumulate the statistics of y (G_out groups)
    int G_out = 1;
};

int conv_tm_run(after_unet1d* h, hipStream_t s, const PackedConv& p, const ConvIo& io, int B, int Tin) {
    const int stride = p.in.istride, Tout = Tin / stride;
    AFTER_REQUIRE((size_t)B * conv_tm_cp(p.cin) * conv_tm_rows(Tin) <= h->xp_elems, AFTER_E_CAPACITY,
                  "unet1d: activation scratch too small");
    ActPadTm a;
    memset(&a, 0, sizeof(a));
    a.x = io.x;
    a.ldx = io.x_ld;
    a.x_cm = io.x_cm;
    a.y = h->xp;
    a.stats = io.stats;
    a.gamma = io.gamma;
    a.beta = io.beta;
    a.act = io.act;
    a.B = B;
    a.C = p.cin;
    a.T = Tin;
    a.G = io.G;
    a.sub_stride = h->max_batch * 16 * kStatWords;
    AFTER_TRY(launch_act_pad_tm(a, s));
    ConvTmRun r;
    memset(&r, 0, sizeof(r));
    r.xp = h->xp;
    r.w = p.wd;
    r.bias = p.bias;
    r.res = io.res;
    r.res_ld = io.res_ld;
    r.res_bs = io.res ? (long long)Tout * io.res_ld : 0;
    r.y = io.y;
    r.y_cm = io.y_cm;
    r.out_act = io.out_act;
    r.post_scale = io.ps;
    r.post_shift = io.pt;
    r.post_bstride = io.ps ? p.cout : 0;
    r.stats = io.stats_out;
    r.G = io.G_out;
    r.sub_stride = h->max_batch * 16 * kStatWords;
    r.B = B;
    r.Tp = conv_tm_rows(Tin);
    r.Tout = Tout;
    r.Nn = Tout;
    return launch_conv_tm(r, p.in, p.plan, s);
}

// ConvBlock1D.forward (unet1d.py:83-118).  x: [B][T][x_ld] view of in_c channels.
int run_block(after_unet1d* h, hipStream_t s, const BlockW& b, const float* x, int x_ld, const float* skip,
              const float* tcond, const float* cond, float* y, int y_cm, int B, int T) {
    const int ccat = b.in_c + b.skip_c + b.tc_c;
    const size_t rows = (size_t)B * T;
    const float* in = x;
    int in_ld = x_ld;
    if (b.skip_c || b.tc_c) {  // torch.cat((x, skip, time_cond), 1) = column ranges of one buffer
        AFTER_TRY(launch_copy_cols_tm(x, x_ld, h->cat, ccat, b.in_c, rows, s));
        if (b.skip_c) AFTER_TRY(launch_copy_cols_tm(skip, b.skip_c, h->cat + b.in_c, ccat, b.skip_c, rows, s));
        if (b.tc_c) AFTER_TRY(launch_copy_cols_tm(tcond, b.tc_c, h->cat + b.in_c + b.skip_c, ccat, b.tc_c, rows, s));
        in = h->cat;
        in_ld = ccat;
    }
    FilmArgs f{h->emb, cond, b.w1t, b.b1t, b.w2t, b.b2t, b.w1c, b.b1c, b.w2c, b.b2c, h->ps, h->pt,
               h->cfg.time_channels, h->cfg.cond_channels, b.out_c, 128};
    hipLaunchKernelGGL(film_kernel, dim3(B), dim3(256), 0, s, f);
    AFTER_HIP_CHECK(hipGetLastError());
    // GroupNorm 1 over the concatenation: its parts have three different producers -> one pass
    double* st1 = next_slot(h);
    AFTER_TRY(launch_stats_accum_tm(in, st1, B, ccat, T, gn_groups(ccat), s, in_ld, h->max_batch * 16 * kStatWords));
    double* st2 = next_slot(h);
    {
        ConvIo io;
        io.x = in;
        io.x_ld = in_ld;
        io.stats = st1;
        io.gamma = b.gn1_w;
        io.beta = b.gn1_b;
        io.G = gn_groups(ccat);
        io.act = ACT_SILU;
        io.y = h->tmp;
        io.ps = h->ps;
        io.pt = h->pt;
        io.stats_out = st2;  // GroupNorm 2 sees exactly this conv's (FiLM'd) output
        io.G_out = gn_groups(b.out_c);
        AFTER_TRY(conv_tm_run(h, s, b.c1, io, B, T));
    }
    const float* res = nullptr;
    int res_ld = 0;
    if (b.res) {
        res = x;
        res_ld = x_ld;
        if (b.has_to_out) {
            ConvIo io;
            io.x = x;
            io.x_ld = x_ld;
            io.y = h->resb;
            AFTER_TRY(conv_tm_run(h, s, b.to_out, io, B, T));
            res = h->resb;
            res_ld = b.out_c;
        }
    }
    ConvIo io;
    io.x = h->tmp;
    io.x_ld = b.out_c;
    io.stats = st2;
    io.gamma = b.gn2_w;
    io.beta = b.gn2_b;
    io.G = gn_groups(b.out_c);
    io.act = ACT_SILU;
    io.y = y;
    io.y_cm = y_cm;
    io.res = res;
    io.res_ld = res_ld;
    return conv_tm_run(h, s, b.c2, io, B, T);
}

// SelfAttention1d.forward (blocks.py:231-243) on the block output x [B][T][C]: y = x + out_proj(attention(qkv_proj(norm(x))))
int run_attn(after_unet1d* h, hipStream_t s, const AttnW& w, const float* x, float* y, int B, int T) {
    const int C = w.C, D = C / w.nh;
    double* st = next_slot(h);
    AFTER_TRY(launch_stats_accum_tm(x, st, B, C, T, 1, s, C, h->max_batch * 16 * kStatWords));
    {
        ConvIo io;
        io.x = x;
        io.x_ld = C;
        io.stats = st;
        io.gamma = w.nw;
        io.beta = w.nb;
        io.G = 1;
        io.y = h->qkvb;
        AFTER_TRY(conv_tm_run(h, s, w.qkv, io, B, T));
    }
    const dim3 grid(cdiv(T, 64), w.nh, B);
    switch (D) {
        case 4: hipLaunchKernelGGL(unet_attn_kernel<4>, grid, dim3(64), 0, s, h->qkvb, h->attb, T, C); break;
        case 8: hipLaunchKernelGGL(unet_attn_kernel<8>, grid, dim3(64), 0, s, h->qkvb, h->attb, T, C); break;
        case 16: hipLaunchKernelGGL(unet_attn_kernel<16>, grid, dim3(64), 0, s, h->qkvb, h->attb, T, C); break;
        case 32: hipLaunchKernelGGL(unet_attn_kernel<32>, grid, dim3(64), 0, s, h->qkvb, h->attb, T, C); break;
        default: hipLaunchKernelGGL(unet_attn_kernel<64>, grid, dim3(64), 0, s, h->qkvb, h->attb, T, C); break;
    }
    AFTER_HIP_CHECK(hipGetLastError());
    ConvIo io;
    io.x = h->attb;
    io.x_ld = C;
    io.y = y;
    io.res = x;
    io.res_ld = C;
    return conv_tm_run(h, s, w.out, io, B, T);
}

}  // namespace

extern "C" int after_unet1d_create(const after_unet1d_cfg* cfg, const float* const* weights,
                                   int n_weights, int max_batch, int max_T, after_unet1d** out) {
    AFTER_REQUIRE(cfg && weights && out, AFTER_E_INVALID, "null argument");
    *out = nullptr;
    const int n = cfg->n_blocks, k = cfg->kernel_size;
    AFTER_REQUIRE(n >= 1 && n <= 8 && k % 2 == 1 && k <= kMaxTaps, AFTER_E_INVALID,
                  "unet1d: 1..8 blocks and an odd kernel_size <= %d required", kMaxTaps);
    AFTER_REQUIRE(cfg->time_cond_channels > 0 && cfg->cond_channels > 0 && cfg->time_channels > 0 &&
                      cfg->time_channels % 2 == 0,
                  AFTER_E_INVALID,
                  "unet1d: built for time_cond_channels > 0, cond_channels > 0, even time_channels");
    AFTER_REQUIRE(cfg->ratios[0] == 1, AFTER_E_INVALID, "unet1d: ratios[0] is the prepended 1 (unet1d.py:283)");
    AFTER_REQUIRE(max_batch > 0 && max_T > 0, AFTER_E_INVALID, "bad capacities");
    after_unet1d* h = new (std::nothrow) after_unet1d();
    AFTER_REQUIRE(h, AFTER_E_NOMEM, "out of host memory");
    h->cfg = *cfg;
    h->n = n;
    h->max_batch = max_batch;
    int total_ratio = 1;
    for (int i = 0; i < n; ++i) total_ratio *= cfg->ratios[i];
    h->max_T = (max_T / total_ratio) * total_ratio;
    auto fail = [&](int rc) {
        after_unet1d_destroy(h);
        return rc;
    };
    const int TC = cfg->time_channels, CC = cfg->cond_channels, tcc = cfg->time_cond_channels;
    const int in0 = cfg->in_size, outsz = cfg->out_size > 0 ? cfg->out_size : cfg->in_size;
    const int* ch = cfg->channels;
    auto in_of = [&](int i) { return i == 0 ? in0 : ch[i - 1]; };
    size_t wf = conv_fl(cfg->time_cond_in_channels, tcc, k) + n * conv_fl(tcc, tcc, k);
    int cmax = in0 > outsz ? in0 : outsz, ccat_max = 0;
    for (int i = 0; i < n; ++i) {
        wf += block_fl(in_of(i), in_of(i), 0, tcc, k, TC, CC) + conv_fl(in_of(i), ch[i], k);
        cmax = ch[i] > cmax ? ch[i] : cmax;
        ccat_max = in_of(i) + tcc > ccat_max ? in_of(i) + tcc : ccat_max;
    }
    wf += block_fl(ch[n - 1], ch[n - 1], 0, tcc, k, TC, CC);
    ccat_max = ch[n - 1] + tcc > ccat_max ? ch[n - 1] + tcc : ccat_max;
    // self-attention layers (unet1d.py:339, 350, 372): down_layers i >= n - n_attn (i >= 1), up_layers i <= n_attn (loop index
    // 1 .. n - 1), the middle block when n_attn > 0
    const int na = cfg->n_attn_layers;
    AFTER_REQUIRE(na >= 0 && na <= n, AFTER_E_INVALID, "unet1d: n_attn_layers = %d", na);
    auto attn_fl = [&](int C) { return conv_fl(C, 3 * C, 1) + conv_fl(C, C, 1) + 2 * (size_t)C + 64; };
    for (int i = 1; i < n; ++i) {
        if (i >= n - na) wf += attn_fl(in_of(i));
        if (i <= na) wf += attn_fl(ch[n - i - 1]);
    }
    if (na > 0) wf += attn_fl(ch[n - 1]);
    for (int i = 1; i <= n; ++i) {
        const int ic = ch[n - i], oc = i < n ? ch[n - i - 1] : outsz;
        const int sk = i < n ? oc : in0;
        wf += conv_fl(ic, oc, 3) + block_fl(oc, oc, sk, tcc, k, TC, CC);
        ccat_max = oc + sk + tcc > ccat_max ? oc + sk + tcc : ccat_max;
        cmax = oc > cmax ? oc : cmax;
    }
    h->cmax = cmax;
    h->ccat_max = ccat_max;
    int rc = h->wa.init(wf * sizeof(float) + (1 << 20));
    if (rc) return fail(rc);
    WCur cur{weights, n_weights};
#define U_TRY(expr)                            \
    do {                                       \
        int rc2__ = (expr);                    \
        if (rc2__ != AFTER_OK) return fail(rc2__); \
    } while (0)
    // cond_emb_time (unet1d.py:296-313): entry 0: in -> tcc stride 1; entry i: stride ratios[i-1]
    h->cond_emb.resize(n + 1);
    U_TRY(load_pconv(h->wa, cur, h->cond_emb[0], cfg->time_cond_in_channels, tcc, k));
    for (int i = 1; i <= n; ++i) U_TRY(load_pconv(h->wa, cur, h->cond_emb[i], tcc, tcc, k, cfg->ratios[i - 1]));
    // down layers (:318-340): ConvBlock(in -> in) then pool(in -> channels[i], stride ratios[i])
    h->down.resize(n);
    h->pool.resize(n);
    h->down_attn.resize(n);
    h->up_attn.resize(n);
    for (int i = 0; i < n; ++i) {
        U_TRY(load_block(h->wa, cur, h->down[i], in_of(i), in_of(i), 0, tcc, k, TC, CC, true));
        if (i >= 1 && i >= n - na) U_TRY(load_attn(h->wa, cur, h->down_attn[i], in_of(i), 4));
        U_TRY(load_pconv(h->wa, cur, h->pool[i], in_of(i), ch[i], k, cfg->ratios[i]));
    }
    U_TRY(load_block(h->wa, cur, h->mid, ch[n - 1], ch[n - 1], 0, tcc, k, TC, CC, true));
    if (na > 0) {
        if (ch[n - 1] < 32) {
            set_error("unet1d: the middle block's self-attention has in_c // 32 heads: %d channels", ch[n - 1]);
            return fail(AFTER_E_INVALID);
        }
        U_TRY(load_attn(h->wa, cur, h->mid_attn, ch[n - 1], ch[n - 1] / 32));
    }
    // up layers (:341-372): i = 1..n-1: channels[n-i] -> channels[n-i-1], ratio ratios[n-i];
    // last: channels[0] -> out_size, ratio ratios[0], skip = in_size, res = use_res_last
    h->up.resize(n);
    h->upconv.resize(n);
    h->up_has_conv.assign(n, 0);
    for (int i = 1; i <= n; ++i) {
        const int ic = ch[n - i], oc = i < n ? ch[n - i - 1] : outsz;
        const int sk = i < n ? oc : in0;
        const int ratio = cfg->ratios[n - i];
        const bool has_conv = ratio != 1 || ic != oc;  // blocks DecoderBlock1D.__init__ :206-221
        h->up_has_conv[i - 1] = has_conv;
        if (has_conv) {
            U_TRY(load_pconv(h->wa, cur, h->upconv[i - 1], ic, oc, 3));
        } else {
            (void)cur.opt();
            (void)cur.opt();
        }
        U_TRY(load_block(h->wa, cur, h->up[i - 1], oc, oc, sk, tcc, k, TC, CC, i < n ? true : cfg->use_res_last != 0));
        if (i < n && i <= na) U_TRY(load_attn(h->wa, cur, h->up_attn[i - 1], oc, 4));
    }
#undef U_TRY
    if (!cur.ok || cur.i != n_weights) {
        set_error("unet1d: expected %d weight tensors, got %d", cur.i, n_weights);
        return fail(AFTER_E_INVALID);
    }
    // workspaces (all activations time-major [B][T][C])
    const size_t T = h->max_T, Bm = max_batch;
    const size_t act = Bm * cmax * T, catn = Bm * ccat_max * T, tcn = Bm * tcc * T;
    int cin_max = ccat_max > cmax ? ccat_max : cmax;
    cin_max = cfg->time_cond_in_channels > cin_max ? cfg->time_cond_in_channels : cin_max;
    h->xp_elems = Bm * (size_t)conv_tm_cp(cin_max) * conv_tm_rows((int)T);
    const size_t stat_d = (size_t)kSlots * conv_tm_stat_sub() * Bm * 16 * kStatWords;
    size_t bytes = (Bm * TC + 2 * Bm * cmax) * sizeof(float) + stat_d * sizeof(double) +
                   (catn + (na > 0 ? 10 : 5) * act + (size_t)n * act + (size_t)(n + 2) * tcn + h->xp_elems) * sizeof(float) +
                   (1 << 16) +
                   (Bm * T * (size_t)(3 * in0 + outsz + cfg->time_cond_in_channels) + Bm * (CC + 1)) * sizeof(float) +
                   8192;
    if ((rc = h->ws.init(bytes))) return fail(rc);
    h->emb = h->ws.take<float>(Bm * TC);
    h->ps = h->ws.take<float>(Bm * cmax);
    h->pt = h->ws.take<float>(Bm * cmax);
    h->stats = h->ws.take<double>(stat_d);
    h->cat = h->ws.take<float>(catn);
    h->tmp = h->ws.take<float>(act);
    h->resb = h->ws.take<float>(act);
    h->xa = h->ws.take<float>(act);
    h->xb = h->ws.take<float>(act);
    h->ups = h->ws.take<float>(act);
    if (na > 0) {
        h->attx = h->ws.take<float>(act);
        h->qkvb = h->ws.take<float>(3 * act);
        h->attb = h->ws.take<float>(act);
        if (!h->attb) return fail(AFTER_E_NOMEM);
    }
    h->xp = h->ws.take<float>(h->xp_elems);
    h->skips.resize(n);
    h->tconds.resize(n + 2);
    for (int i = 0; i < n; ++i) h->skips[i] = h->ws.take<float>(act);
    for (int i = 0; i < n + 2; ++i) h->tconds[i] = h->ws.take<float>(tcn);
    h->xtm = h->ws.take<float>(Bm * in0 * T);
    h->x3 = h->ws.take<float>(Bm * in0 * T);
    h->d3 = h->ws.take<float>(Bm * outsz * T);
    h->xs = h->ws.take<float>(Bm * in0 * T);
    h->tc3 = h->ws.take<float>(Bm * cfg->time_cond_in_channels * T);
    h->cond3 = h->ws.take<float>(Bm * CC);
    h->t3 = h->ws.take<float>(Bm);
    if (!h->tconds[n + 1] || !h->skips[n - 1] || !h->ups || !h->t3 || !h->xp || !h->xtm) return fail(AFTER_E_NOMEM);
    if (hipDeviceSynchronize() != hipSuccess) {  // weight repacks done before the staging copies are reused
        set_error("unet1d: device initialisation failed");
        return fail(AFTER_E_HIP);
    }
    *out = h;
    return AFTER_OK;
}

extern "C" void after_unet1d_destroy(after_unet1d* h) {
    if (!h) return;
    h->wa.release();
    h->ws.release();
    delete h;
}

// UNET1D.forward, time_cond_channels > 0 branch (unet1d.py:374-414)
extern "C" int after_unet1d_forward(after_unet1d* h, const float* x, const float* time, const float* cond,
                                    const float* time_cond, float* out, int B, int T, void* stream) {
    AFTER_REQUIRE(h && x && time && cond && time_cond && out, AFTER_E_INVALID, "null argument");
    AFTER_REQUIRE(B > 0 && T > 0, AFTER_E_INVALID, "empty batch");
    AFTER_REQUIRE(B <= h->max_batch && T <= h->max_T, AFTER_E_CAPACITY,
                  "B=%d T=%d exceed max_batch=%d max_T=%d", B, T, h->max_batch, h->max_T);
    const after_unet1d_cfg& c = h->cfg;
    const int n = h->n;
    int total_ratio = 1;
    for (int i = 0; i < n; ++i) total_ratio *= c.ratios[i];
    AFTER_REQUIRE(T % total_ratio == 0, AFTER_E_INVALID, "unet1d: T=%d not a multiple of %d", T, total_ratio);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(spe_kernel, dim3(cdiv(B * c.time_channels / 2, 256)), dim3(256), 0, s, time, h->emb, B,
                       c.time_channels, 10000.0f, 32.0f);
    AFTER_HIP_CHECK(hipGetLastError());
    // GroupNorm accumulators of this forward: one slot per normalisation, zeroed together
    h->stat_slot = 0;
    AFTER_HIP_CHECK(hipMemsetAsync(h->stats, 0, (size_t)kSlots * conv_tm_stat_sub() * h->max_batch * 16 * kStatWords * sizeof(double), s));
    AFTER_REQUIRE(2 * (2 * n + 1) + 2 * n <= kSlots, AFTER_E_INVALID, "unet1d: statistics slots");
    // the network input is read twice (concatenation, shortcut): one transpose
    AFTER_TRY(launch_cm_to_tm(x, h->xtm, B, c.in_size, T, c.in_size, s));
    // ---- encoder: time_cond is re-embedded (conv + SiLU) at every scale
    const float* cur = h->xtm;
    int cur_c = c.in_size;
    int Tc = T;  // length of x == length of the time_cond of this scale
    std::vector<int> Ts(n);
    for (int i = 0; i < n; ++i) {
        ConvIo io;
        io.x = i == 0 ? time_cond : h->tconds[i - 1];
        io.x_ld = c.time_cond_channels;
        io.x_cm = i == 0;
        io.y = h->tconds[i];
        io.out_act = ACT_SILU;
        AFTER_TRY(conv_tm_run(h, s, h->cond_emb[i], io, B, i == 0 ? T : Ts[i - 1]));
        Ts[i] = Tc;
        if (h->down_attn[i].on) {  // skip = self_attn(conv(x)), unet1d.py:162-163
            AFTER_TRY(run_block(h, s, h->down[i], cur, cur_c, nullptr, h->tconds[i], cond, h->attx, 0, B, Tc));
            AFTER_TRY(run_attn(h, s, h->down_attn[i], h->attx, h->skips[i], B, Tc));
        } else {
            AFTER_TRY(run_block(h, s, h->down[i], cur, cur_c, nullptr, h->tconds[i], cond, h->skips[i], 0, B, Tc));
        }
        float* nx = (cur == h->xa) ? h->xb : h->xa;
        ConvIo pl;
        pl.x = h->skips[i];
        pl.x_ld = h->down[i].out_c;
        pl.y = nx;
        AFTER_TRY(conv_tm_run(h, s, h->pool[i], pl, B, Tc));
        cur = nx;
        cur_c = h->pool[i].cout;
        Tc /= c.ratios[i];
    }
    {
        ConvIo io;
        io.x = h->tconds[n - 1];
        io.x_ld = c.time_cond_channels;
        io.y = h->tconds[n];
        io.out_act = ACT_SILU;
        AFTER_TRY(conv_tm_run(h, s, h->cond_emb[n], io, B, Ts[n - 1]));
        float* nx = (cur == h->xa) ? h->xb : h->xa;
        if (h->mid_attn.on) {
            AFTER_TRY(run_block(h, s, h->mid, cur, cur_c, nullptr, h->tconds[n], cond, h->attx, 0, B, Tc));
            AFTER_TRY(run_attn(h, s, h->mid_attn, h->attx, nx, B, Tc));
        } else {
            AFTER_TRY(run_block(h, s, h->mid, cur, cur_c, nullptr, h->tconds[n], cond, nx, 0, B, Tc));
        }
        cur = nx;
    }
    // ---- decoder
    for (int i = 1; i <= n; ++i) {
        const int ratio = c.ratios[n - i];
        const float* up_in = cur;
        if (ratio != 1) {  // nn.Upsample(mode="nearest"): every row `ratio` times
            AFTER_TRY(launch_upsample_rows_tm(cur, h->ups, B, cur_c, Tc, ratio, s));
            up_in = h->ups;
            Tc *= ratio;
        }
        float* ux = (cur == h->xa) ? h->xb : h->xa;
        const float* bx = up_in;
        if (h->up_has_conv[i - 1]) {
            ConvIo io;
            io.x = up_in;
            io.x_ld = cur_c;
            io.y = ux;
            AFTER_TRY(conv_tm_run(h, s, h->upconv[i - 1], io, B, Tc));
            bx = ux;
            cur_c = h->upconv[i - 1].cout;
        }
        float* y = i == n ? out : ((bx == h->xa) ? h->xb : h->xa);
        if (i < n && y == bx) y = h->ups;  // (identity `up`: keep input and output apart)
        if (h->up_attn[i - 1].on) {  // (never the last block: y is time-major here)
            AFTER_TRY(run_block(h, s, h->up[i - 1], bx, cur_c, h->skips[n - i], h->tconds[n - i], cond, h->attx, 0, B, Tc));
            AFTER_TRY(run_attn(h, s, h->up_attn[i - 1], h->attx, y, B, Tc));
        } else {
            AFTER_TRY(run_block(h, s, h->up[i - 1], bx, cur_c, h->skips[n - i], h->tconds[n - i], cond, y, i == n, B, Tc));
        }
        cur = y;
        cur_c = h->up[i - 1].out_c;
    }
    return AFTER_OK;
}

// ---- RectifiedFlow.model_forward / .sample around UNET1D (model.py:721-785), all on the device:
// the 3x CFG batch is assembled by kernels, the network runs on 3B rows, the combination and the
// Euler update are one kernel.  The handle must have been created with max_batch >= 3 B.
namespace {

int cfg_factors(int cfg_mode, float gt, float gs, float* total, float* factor) {
    *total = 0.5f * (gs + gt);
    if (cfg_mode == 0) *factor = gt / (gs > 0.01f ? gs : 0.01f);       // model.py:749-759
    else if (cfg_mode == 1) *factor = gt / (gs > 0.1f ? gs : 0.1f);    // export.py:364-394
    else if (cfg_mode == 2) *factor = gs / (gt > 0.1f ? gt : 0.1f);    // export_midi.py:329-358
    else {
        set_error("unet1d: unknown cfg_mode %d", cfg_mode);
        return AFTER_E_INVALID;
    }
    return AFTER_OK;
}

int cfg_eval(after_unet1d* h, hipStream_t s, const float* x, const float* time, int step, int nb_steps, int B,
             int T) {
    const size_t per = (size_t)h->cfg.in_size * T;
    const size_t n3 = 3 * (size_t)B * per;
    hipLaunchKernelGGL(cfg3_x_kernel, dim3((unsigned)cdivll((long long)n3, 256)), dim3(256), 0, s, x, h->x3, time,
                       h->t3, B, per, step, nb_steps);
    AFTER_HIP_CHECK(hipGetLastError());
    return after_unet1d_forward(h, h->x3, h->t3, h->cond3, h->tc3, h->d3, 3 * B, T, s);
}

int cfg_prepare(after_unet1d* h, hipStream_t s, const float* cond, const float* tc, int B, int T, float drop,
                int cfg_mode) {
    AFTER_REQUIRE(3 * B <= h->max_batch, AFTER_E_CAPACITY, "unet1d: CFG needs max_batch >= 3 B (B=%d, max_batch=%d)",
                  B, h->max_batch);
    AFTER_REQUIRE((h->cfg.out_size > 0 ? h->cfg.out_size : h->cfg.in_size) == h->cfg.in_size, AFTER_E_INVALID,
                  "unet1d: the sampler needs out_size == in_size");
    // before anything is launched: cfg3_cond_kernel / cfg3_x_kernel write workspaces sized for max_T
    AFTER_REQUIRE(T > 0 && T <= h->max_T, AFTER_E_CAPACITY, "unet1d: T=%d exceeds max_T=%d", T, h->max_T);
    {
        int total_ratio = 1;
        for (int i = 0; i < h->n; ++i) total_ratio *= h->cfg.ratios[i];
        AFTER_REQUIRE(T % total_ratio == 0, AFTER_E_INVALID, "unet1d: T=%d not a multiple of %d", T, total_ratio);
    }
    const size_t tc_per = (size_t)h->cfg.time_cond_in_channels * T;
    const size_t n = 3 * (size_t)B * (tc_per > (size_t)h->cfg.cond_channels ? tc_per : h->cfg.cond_channels);
    hipLaunchKernelGGL(cfg3_cond_kernel, dim3((unsigned)cdivll((long long)n, 256)), dim3(256), 0, s, cond, tc,
                       h->cond3, h->tc3, B, h->cfg.cond_channels, tc_per, drop, cfg_mode == 2 ? 1 : 0);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

}  // namespace

extern "C" int after_unet1d_model_forward(after_unet1d* h, const float* x, const float* time, const float* cond,
                                          const float* time_cond, float* out, int B, int T, float g_timbre,
                                          float g_structure, float drop_value, int cfg_mode, void* stream) {
    AFTER_REQUIRE(h && x && time && cond && time_cond && out && B > 0 && T > 0, AFTER_E_INVALID, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    float total, factor;
    AFTER_TRY(cfg_factors(cfg_mode, g_timbre, g_structure, &total, &factor));
    AFTER_TRY(cfg_prepare(h, s, cond, time_cond, B, T, drop_value, cfg_mode));
    AFTER_TRY(cfg_eval(h, s, x, time, 0, 1, B, T));
    const size_t n = (size_t)B * h->cfg.in_size * T;
    hipLaunchKernelGGL(cfg3_combine_kernel, dim3((unsigned)cdivll((long long)n, 256)), dim3(256), 0, s, h->d3,
                       (const float*)nullptr, out, n, total, factor, 1.0f, 0.0f);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

extern "C" int after_unet1d_sample(after_unet1d* h, const float* x0, const float* cond, const float* time_cond,
                                   float* out, int B, int T, int nb_steps, float g_timbre, float g_structure,
                                   float drop_value, int cfg_mode, void* stream) {
    AFTER_REQUIRE(h && x0 && cond && time_cond && out && B > 0 && T > 0 && nb_steps > 0, AFTER_E_INVALID,
                  "bad argument");
    hipStream_t s = (hipStream_t)stream;
    float total, factor;
    AFTER_TRY(cfg_factors(cfg_mode, g_timbre, g_structure, &total, &factor));
    AFTER_TRY(cfg_prepare(h, s, cond, time_cond, B, T, drop_value, cfg_mode));
    const size_t n = (size_t)B * h->cfg.in_size * T;
    const float dt = 1.0f / (float)nb_steps;
    const float* xin = x0;
    for (int i = 0; i < nb_steps; ++i) {
        AFTER_TRY(cfg_eval(h, s, xin, nullptr, i, nb_steps, B, T));
        // ping-pong between out and the handle's scratch so that the last step lands in `out`
        float* dst = ((nb_steps - 1 - i) & 1) ? h->xs : out;
        hipLaunchKernelGGL(cfg3_combine_kernel, dim3((unsigned)cdivll((long long)n, 256)), dim3(256), 0, s, h->d3,
                           xin, dst, n, total, factor, dt, 1.0f);
        AFTER_HIP_CHECK(hipGetLastError());
        xin = dst;
    }
    return AFTER_OK;
}
