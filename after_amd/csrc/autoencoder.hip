// AutoEncoder.encode / .decode on gfx950 (reference
// after/autoencoder/networks/SimpleNetsStream.py:831-954, pqmf.py:252-301).
//
// Activations are time-major [B][T][C] between the API edges.  Every conv layer is
// GroupNorm-apply + SnakeBeta + halo (act_pad_tm) followed by the balanced LDS-DMA GEMM of
// conv_tm.hip with bias / residual in its epilogue; the full-sequence GroupNorm statistics
// (offline semantics: one (mean, var) per (clip, group) over the WHOLE time axis,
// SimpleNetsStream.py:146-147) are accumulated by the producing conv's epilogue, and a
// conv that feeds a Snake-only consumer writes that consumer's activated input itself.
// The two PQMF filter banks are HBM-bound polyphase FIRs on the vector ALUs with their
// taps in scalar registers.
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

#include <cmath>
#include <cstring>

#include "conv.h"
#include "gemm_h3_pipe.h"

namespace after {
namespace {

// ------------------------------------------------------------------ PQMF analysis
// mb[b, c, n] = sgn(c, n) * sum_k w[c][k] x[b, 16 n + k - pl]      (pqmf.py:286-290, 16-20)
// polyphase form: k = M q + r  ->  sum_r sum_q w[c][M q + r] xp[r][n + q],
// xp[r][m] = x[M m + r - pl] staged de-interleaved in LDS, so lanes (= consecutive n) read
// consecutive addresses.  One lane = one frame, all M = 16 bands in registers; the taps are
// re-laid-out at create time to wp[r][q][c] so that the 16 band weights of one (r, q) are one
// wave-uniform s_load_dwordx16 (SGPR operands of the 16 FMAs).
constexpr int PQ_BT = 256;  // frames per block

template <int M>
__global__ __launch_bounds__(256) void pqmf_forward_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ wp,
                                                           float* __restrict__ mb, int L, int Q,
                                                           int pl,
                                                           const float* __restrict__ state, int cs, int ts) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ldp = PQ_BT + Q + 1;
    float* ws = sm;                // [M][Q][M] taps, read as wave-uniform (broadcast) float4s
    float* xp = sm + M * Q * M;    // [M][PQ_BT + Q + 1]
    const int b = blockIdx.y, n0 = blockIdx.x * PQ_BT;
    const int Tm = L / M;
    const float* xb = x + (size_t)b * L;
    for (int idx = threadIdx.x; idx < M * Q * M / 4; idx += 256)
        reinterpret_cast<float4*>(ws)[idx] = reinterpret_cast<const float4*>(wp)[idx];
    for (int idx = threadIdx.x; idx < M * (PQ_BT + Q); idx += 256) {
        const int m = idx / M, r = idx - m * M;  // consecutive threads read consecutive samples
        const long long s = (long long)(n0 + m) * M + r - pl;
        float v = 0.f;
        if (s >= 0 && s < L) v = xb[s];
        else if (s < 0 && state) v = state[(size_t)b * pl + (pl + s)];  // streaming: pl = K - 1
        xp[r * ldp + m] = v;
    }
    __syncthreads();
    float acc[M];
#pragma unroll
    for (int c = 0; c < M; ++c) acc[c] = 0.f;
    for (int r = 0; r < M; ++r) {
        const float* xr = xp + r * ldp + threadIdx.x;
        const float4* wr = reinterpret_cast<const float4*>(ws + (size_t)r * Q * M);
#pragma unroll 3
        for (int q = 0; q < Q; ++q) {
            const float xv = xr[q];
#pragma unroll
            for (int c4 = 0; c4 < M / 4; ++c4) {
                const float4 w4 = wr[q * (M / 4) + c4];
                acc[4 * c4 + 0] += w4.x * xv;
                acc[4 * c4 + 1] += w4.y * xv;
                acc[4 * c4 + 2] += w4.z * xv;
                acc[4 * c4 + 3] += w4.w * xv;
            }
        }
    }
    const int n = n0 + threadIdx.x;
    if (n < Tm) {
#pragma unroll
        for (int c = 0; c < M; ++c) {
            const bool neg = (c & 1) && !(n & 1);  // reverse_half: odd bands, even frames
            mb[(size_t)b * M * Tm + (size_t)c * cs + (size_t)n * ts] = neg ? -acc[c] : acc[c];
        }
    }
}

// generic fallback (any M): one thread per (band, frame)
__global__ __launch_bounds__(256) void pqmf_forward_generic_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ w,
                                                                   float* __restrict__ mb, int L, int M,
                                                                   int K, int pl,
                                                                   const float* __restrict__ state, int cs,
                                                                   int ts) {
    const int Tm = L / M;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (idx >= (size_t)Tm * M) return;
    const int c = idx / Tm, n = idx - (size_t)c * Tm;
    const float* xb = x + (size_t)b * L;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const long long s = (long long)n * M + k - pl;
        if (s >= 0 && s < L) acc += w[(size_t)c * K + k] * xb[s];
        else if (s < 0 && state) acc += w[(size_t)c * K + k] * state[(size_t)b * pl + (pl + s)];
    }
    mb[(size_t)b * M * Tm + (size_t)c * cs + (size_t)n * ts] = ((c & 1) && !(n & 1)) ? -acc : acc;
}

// ------------------------------------------------------------------ PQMF synthesis
// audio[b, M t + m] = M * sum_c sum_k w[M-1-m][c][k] z[c][t + k - pl],
// z[c][t] = sgn(c, t) * band(c, t),  band = y[c] * sigmoid(y[M + c]) with the loudness
// gate (SimpleNetsStream.py:644-646) or y[c] without.          (pqmf.py:292-301)
// One lane = one frame t, all M phases in registers; taps re-laid-out to wi[c][k][m] =
// w[M-1-m][c][k] so the 16 phase weights of one (c, k) are one s_load_dwordx16.
template <int M>
__global__ __launch_bounds__(256) void pqmf_inverse_kernel(const float* __restrict__ y,
                                                           const float* __restrict__ wi,
                                                           float* __restrict__ audio, int Tm,
                                                           int K, int pl, int gated, int ychan,
                                                           const float* __restrict__ zstate, int cs, int ts) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ldz = 256 + K;
    float* ws = sm;              // [M][K][M] taps (broadcast float4 reads)
    float* zs = sm + M * K * M;  // [M][256 + K]
    const int b = blockIdx.y, t0 = blockIdx.x * 256;
    const float* yb = y + (size_t)b * ychan * Tm;
    for (int idx = threadIdx.x; idx < M * K * M / 4; idx += 256)
        reinterpret_cast<float4*>(ws)[idx] = reinterpret_cast<const float4*>(wi)[idx];
    for (int idx = threadIdx.x; idx < M * ldz; idx += 256) {
        const int c = idx / ldz, col = idx - c * ldz;
        const int t = t0 + col - pl;
        float v = 0.f;
        if (t >= 0 && t < Tm) {
            v = yb[(size_t)c * cs + (size_t)t * ts];
            if (gated) v *= 1.0f / (1.0f + expf(-yb[(size_t)(M + c) * cs + (size_t)t * ts]));
            if ((c & 1) && !(t & 1)) v = -v;
        } else if (t < 0 && zstate) {  // streaming: gated, sign-flipped frames of the last chunk
            v = zstate[((size_t)b * M + c) * pl + (pl + t)];
        }
        zs[idx] = v;
    }
    __syncthreads();
    float acc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = 0.f;
    for (int c = 0; c < M; ++c) {
        const float* zc = zs + c * ldz + threadIdx.x;
        const float4* wc = reinterpret_cast<const float4*>(ws + (size_t)c * K * M);
#pragma unroll 3
        for (int k = 0; k < K; ++k) {
            const float zv = zc[k];
#pragma unroll
            for (int m4 = 0; m4 < M / 4; ++m4) {
                const float4 w4 = wc[k * (M / 4) + m4];
                acc[4 * m4 + 0] += w4.x * zv;
                acc[4 * m4 + 1] += w4.y * zv;
                acc[4 * m4 + 2] += w4.z * zv;
                acc[4 * m4 + 3] += w4.w * zv;
            }
        }
    }
    const int t = t0 + threadIdx.x;
    if (t < Tm) {
        float* o = audio + (size_t)b * Tm * M + (size_t)t * M;
#pragma unroll
        for (int m = 0; m < M; m += 4)
            *reinterpret_cast<float4*>(o + m) = make_float4(acc[m] * M, acc[m + 1] * M,
                                                            acc[m + 2] * M, acc[m + 3] * M);
    }
}

// wp[r][q][c] = w[c][M q + r] (0 past K);  wi[c][k][m] = w[M-1-m][c][k]
__global__ void pqmf_relayout_kernel(const float* __restrict__ fw, const float* __restrict__ iw,
                                     float* __restrict__ wp, float* __restrict__ wi, int M, int Kf,
                                     int Q, int Ki) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < M * Q * M) {
        const int c = idx % M, q = (idx / M) % Q, r = idx / (M * Q);
        const int k = M * q + r;
        wp[idx] = k < Kf ? fw[(size_t)c * Kf + k] : 0.f;
    }
    if (idx < M * Ki * M) {
        const int m = idx % M, k = (idx / M) % Ki, c = idx / (M * Ki);
        wi[idx] = iw[((size_t)(M - 1 - m) * M + c) * Ki + k];
    }
}

// generic (any M) fallback of the synthesis bank: one thread per output sample
__global__ __launch_bounds__(256) void pqmf_inverse_generic_kernel(const float* __restrict__ y,
                                                                   const float* __restrict__ w,
                                                                   float* __restrict__ audio,
                                                                   int Tm, int M, int K, int pl,
                                                                   int gated, int ychan,
                                                                   const float* __restrict__ zstate, int cs,
                                                                   int ts) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (idx >= (size_t)Tm * M) return;
    const int t = idx / M, m = idx - (size_t)t * M;
    const float* yb = y + (size_t)b * ychan * Tm;
    float acc = 0.f;
    for (int c = 0; c < M; ++c)
        for (int k = 0; k < K; ++k) {
            const int tt = t + k - pl;
            if (tt >= Tm || (tt < 0 && !zstate)) continue;
            float v;
            if (tt < 0) {
                v = zstate[((size_t)b * M + c) * pl + (pl + tt)];
            } else {
                v = yb[(size_t)c * cs + (size_t)tt * ts];
                if (gated) v *= 1.0f / (1.0f + expf(-yb[(size_t)(M + c) * cs + (size_t)tt * ts]));
                if ((c & 1) && !(tt & 1)) v = -v;
            }
            acc += w[((size_t)(M - 1 - m) * M + c) * K + k] * v;
        }
    audio[(size_t)b * Tm * M + idx] = acc * M;
}

// streaming synthesis state: the last S = K - 1 gated, sign-flipped band frames of the chunk
__global__ void pqmf_istate_kernel(const float* __restrict__ y, float* __restrict__ zstate, int Tm,
                                   int M, int S, int gated, int ychan, int total, int cs, int ts) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = idx % S, c = (idx / S) % M, b = idx / (S * M);
    const int t = Tm - S + j;
    const float* yb = y + (size_t)b * ychan * Tm;
    float v = yb[(size_t)c * cs + (size_t)t * ts];
    if (gated) v *= 1.0f / (1.0f + expf(-yb[(size_t)(M + c) * cs + (size_t)t * ts]));
    if ((c & 1) && !(t & 1)) v = -v;
    zstate[idx] = v;
}

// CachedGroupNorm(stream=True) (SimpleNetsStream.py:95-147): GroupNorm statistics over the previous
// P frames + this chunk.  ring[b][g][i] = (sum, sum of squares) over the group's channels of one of
// the last P frames (circular, `head` = oldest).  One block per (group, clip): adds the ring (before
// overwriting it), the chunk's frames, stores the totals where act_pad_tm reads its sub-slot 0 and
// moves the chunk's last min(T, P) frames into the ring.  Fixed summation order.
__global__ __launch_bounds__(256) void gn_window_kernel(const float* __restrict__ x, float* __restrict__ ring,
                                                        double* __restrict__ stats, int T, int C, int G, int P,
                                                        int head) {
    __shared__ double red[2][256];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int Cg = C / G;
    float* rg = ring + ((size_t)b * G + g) * P * 2;
    double s = 0, q = 0;
    for (int i = tid; i < P; i += 256) {
        s += rg[2 * i];
        q += rg[2 * i + 1];
    }
    __syncthreads();  // every ring entry is read before any is replaced
    const int keep0 = T > P ? T - P : 0;  // first chunk frame that stays in the window
    for (int t = tid; t < T; t += 256) {
        const float* xr = x + ((size_t)b * T + t) * C + (size_t)g * Cg;
        float fs = 0.f, fq = 0.f;
        for (int c = 0; c < Cg; ++c) {
            const float v = xr[c];
            fs += v;
            fq += v * v;
        }
        s += fs;
        q += fq;
        if (t >= keep0) {
            const int slot = T >= P ? t - keep0 : (head + t) % P;
            rg[2 * slot] = fs;
            rg[2 * slot + 1] = fq;
        }
    }
    red[0][tid] = s;
    red[1][tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            red[0][tid] += red[0][tid + o];
            red[1][tid] += red[1][tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {  // (the act_pad consumer reads binned integer words: conv.h, stat_bins)
        long long* sp = reinterpret_cast<long long*>(stats) + ((size_t)b * G + g) * kStatWords;
        stat_bins_set(sp, red[0][0]);
        stat_bins_set(sp + kStatBins, red[1][0]);
    }
}

// cached_conv.CachedPadding1d(d, crop=True) on a time-major tensor: out[t] = t < d ? state[t] : x[t - d];
// the new state is the last d rows of (state, x), written to the other half of the ping-pong pair.
__global__ void delay_rows_kernel(const float* __restrict__ x, const float* __restrict__ st_in,
                                  float* __restrict__ st_out, float* __restrict__ out, int T, int C, int d,
                                  size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const size_t per = (size_t)(T + d) * C;
    const int b = (int)(idx / per);
    const size_t e = idx - (size_t)b * per;
    const int r = (int)(e / C), c = (int)(e - (size_t)r * C);
    const float* xb = x + (size_t)b * T * C;
    const float* sb = st_in + (size_t)b * d * C;
    if (r < T) {
        out[((size_t)b * T + r) * C + c] = r < d ? sb[(size_t)r * C + c] : xb[(size_t)(r - d) * C + c];
    } else {
        const int j = r - T;  // new state row j = row (j + T) of (state, x)
        st_out[((size_t)b * d + j) * C + c] = j + T < d ? sb[(size_t)(j + T) * C + c] : xb[(size_t)(j + T - d) * C + c];
    }
}

}  // namespace
}  // namespace after

using namespace after;

namespace {

// geometry + GEMM-operand weights of one conv (conv_tm.hip)
struct DmaConv {
    ConvDmaPlanIn in;
    ConvTmPlan tplan;
    float* w = nullptr;
    unsigned short* w3 = nullptr;     // the same weights as bf16 planes (conv_x6.hip), where the layer has a bf16-pipe form
    unsigned short* wh3 = nullptr;    // ... and as two fp16 pieces x ws (conv_h3_split: for inputs a GroupNorm bounds, gemm_h3_pipe.h)
    float ws = 0.f;                   // the weights' power-of-two scale: max|w| x ws <= 2^15
    int toff_offline[kMaxTaps] = {};  // phase-0 taps of the offline plan while a cached (streaming) form is active
};
struct ConvBlockW {
    float *gn_w = nullptr, *gn_b = nullptr, *alpha = nullptr, *invb = nullptr, *w = nullptr,
          *bias = nullptr;
    int cin = 0, cout = 0, k = 1, dil = 1;
    DmaConv d;
    // bounds of the block's parameters (create time): |GroupNorm(x)_i| <= sqrt(n) gn_wmax + gn_bmax over a group of n elements, and
    // |SnakeBeta(v)| <= |v| + invb_max -- what the power-of-two scale of the conv input's fp16 pieces is chosen from (run_dma)
    float gn_wmax = 0.f, gn_bmax = 0.f, invb_max = 0.f;
    // CachedGroupNorm(stream=True): per-frame group sums of the previous gn_P frames (circular)
    float* gn_ring = nullptr;  // [max_batch][G][gn_P][2]
    int gn_P = 0;
    mutable int gn_head = 0;
};
struct ResBlockW {
    ConvBlockW cb0, cb1;
    float *to_w = nullptr, *to_b = nullptr;  // 1x1 shortcut when cin != cout
    DmaConv to_d;
    // cached_conv.AlignBranches: the shortcut's input runs `delay` frames late (ping-pong delay line)
    float* dline[2] = {nullptr, nullptr};  // [max_batch][delay][cin]
    int delay = 0;
    mutable int dflip = 0;
};
struct ResampleW {
    float *alpha = nullptr, *invb = nullptr, *w = nullptr, *bias = nullptr;
    int cin = 0, cout = 0, f = 1;
    DmaConv d;
    float* w_stream = nullptr;  // ConvTranspose1d packed with padding 0 (overlap-add form)
    DmaConv d_stream;
};
struct PlainConvW {
    float *w = nullptr, *bias = nullptr;
    int cin = 0, cout = 0, k = 3;
    DmaConv d;
};

}  // namespace

struct after_ae {
    after_ae_cfg cfg;
    int M, ratio, max_batch, max_samples;
    bool causal, norm;
    Arena wa, ws;
    float *pq_fw = nullptr, *pq_iw = nullptr;    // reference layouts
    float *pq_fwp = nullptr, *pq_iwp = nullptr;  // polyphase / phase-major re-layouts
    int pq_fk = 0, pq_ik = 0;
    // encoder
    ResBlockW enc_stem;
    std::vector<std::vector<ResBlockW>> enc_res;
    std::vector<ResampleW> enc_down;
    float *enc_tail_alpha = nullptr, *enc_tail_invb = nullptr;
    PlainConvW enc_tail;
    // decoder
    PlainConvW dec_head;
    std::vector<ResampleW> dec_up;
    std::vector<std::vector<ResBlockW>> dec_res;
    ConvBlockW synth0, synth1;
    // NoiseGenerator (SimpleNetsStream.py:499-550; use_noise): three plain k = 3, stride-2 convs with LeakyReLU(0.2)
    // between them, then filtered uniform noise added to the band signal
    struct NoiseW {
        float *w[3] = {nullptr, nullptr, nullptr}, *bias[3] = {nullptr, nullptr, nullptr};
        int cin[3] = {0, 0, 0}, cout[3] = {0, 0, 0};
        DmaConv d[3];
    } noise;
    float *nz[3] = {nullptr, nullptr, nullptr}, *nadd = nullptr;  // conv outputs [B][T / 2^i][C]; gated bands + noise [B][T][M]
    const float* noise_u = nullptr;  // after_ae_set_noise: uniform [0, 1) draws [B][T / 8][M][8] for the next decode
    // workspaces
    float *buf[3] = {nullptr, nullptr, nullptr};
    size_t buf_elems = 0;
    int cmax = 0;
    // conv path (conv_tm.hip): activations time-major [B][T][C] between the API edges
    float* xp = nullptr;          // activated + haloed scratch tensor
    unsigned short* xp3 = nullptr;  // the same, as bf16 planes: input of the convs that run on the bf16 pipe (conv_x6.hip)
    size_t xp3_elems = 0;
    float* xp2 = nullptr;         // second one: a conv epilogue prepares the NEXT conv's input there
    const float* prepared = nullptr;  // haloed input already laid out by the producer (time-major path)
    const float* next_alpha = nullptr;  // Snake of the following resampling conv: request to the next
    const float* next_invb = nullptr;   //   run_dma to emit that conv's activated input itself
    const ConvBlockW* bound_of = nullptr;  // the ConvBlock1d whose GroupNorm -> Snake feeds the next run_dma: its parameter bounds
    int h3_mode = 1;                    // AFTER_CONV_H3=0: GroupNorm-bounded convs of the bf16 pipe on three bf16 planes (A/B)
    size_t xp_elems = 0;
    double* stats_ring = nullptr; // [kStatSlots][stat_sub][max_batch][8][2][kStatBins] 64-bit words (conv.h: stat_bins)
    int stat_sub = 1;             // accumulator pairs per (clip, group): conv_tm_stat_sub() on the time-major path
    int stat_slot = 0;
    Arena wd;                     // repacked weights
    // streaming (cached-conv semantics): left-context state per conv, in traversal order
    bool streaming = false;
    Arena sa;
    float *enc_state = nullptr, *dec_state = nullptr;  // [slot][max_batch][cmax][HALO]
    float *pq_fstate = nullptr, *pq_istate = nullptr;  // [B][Kf-1] audio, [B][M][Ki-1] bands
    int state_slot = 0;
    // every conv's context exists twice: a pass reads half `flip` and writes half `flip ^ 1` (act_pad_tm writes the new context
    // itself: no second launch per conv), and a pass that completed flips.  pass_*: the area of the pass being issued.
    int enc_slots = 0, dec_slots = 0, nc_slots = 0;
    int enc_flip = 0, dec_flip = 0, nc_flip = 0;
    // rows of the streams that a flip belongs to, fixed by the first pass after a reset (0 = none yet): a pass updates the
    // contexts of ITS rows only, so a later pass with other rows would read two-chunk-old contexts for the rest -- refused
    int enc_rows = 0, enc_rows1 = 0, dec_rows = 0, nc_rows = 0;
    // after_ae_set_stream_lanes: the streaming encoder's batch rows are two independent groups of `lane_rows` streams
    // each (Streamer: structure audio, timbre audio) with their own context parity, so that ONE pass can encode both
    // groups (rows [0, 2 lane_rows)) or either alone (rows [row0, row0 + B)); 0 = one group
    int lane_rows = 0, enc_flip1 = 0;
    int pass_row0 = 0;  // first state row of the pass in flight
    int pass_slots = 0;
    int* pass_flip = nullptr;
    size_t slot_elems = 0;
    // streaming NON-causal encoder (export_autoencoder.py:305-312): cached centred-padding convs with
    // delay compensation + CachedGroupNorm(stream=True); the PQMF and the decoder stay offline
    bool enc_cached = false;   // mode enabled
    bool pass_stream = false;  // the pass being issued keeps conv state
    // A pass that keeps host-side stream counters (windowed-GroupNorm ring heads, delay-line flips: advanced at
    // ENQUEUE time) and fails half way leaves earlier layers advanced and later ones not: `in_pass` stays set and
    // the next stateful call is refused until after_ae_reset_state.  (For the same reason such passes must not be
    // captured into a hipGraph: a replay would reuse the baked-in heads.)
    bool in_pass = false;
    bool pass_cached = false;  // ... and is the cached non-causal encoder
    bool pass_gnwin = false;   // GroupNorm statistics over a sliding window (CachedGroupNorm.stream)
    int dec_gn_frames = 0;     // decoder: CachedGroupNorm window in latent frames (0: plain GroupNorm)
    int dec_gn_alloc = 0;
    Arena sg;                  // the decoder's GroupNorm rings
    int gn_window = 0;         // CachedGroupNorm window, in audio samples
    int enc_delay = 0;         // latent frames the cached encoder lags the offline one
    Arena sn;                  // its state: conv contexts, delay lines, GroupNorm rings
    float* nc_state = nullptr;
};
constexpr int kStatSlots = 96;

namespace {

struct WeightCursor {
    const float* const* w;
    int n, i = 0;
    bool ok = true;
    const float* next(bool optional = false) {
        if (i >= n) {
            ok = false;
            return nullptr;
        }
        const float* p = w[i++];
        if (!p && !optional) ok = false;
        return p;
    }
};

#define AE_TAKE(ptr, n)                                              \
    do {                                                             \
        (ptr) = h->wa.take<float>(n);                                \
        if (!(ptr)) {                                                \
            set_error("autoencoder: weight arena exhausted");        \
            return AFTER_E_NOMEM;                                    \
        }                                                            \
    } while (0)

// max |p[i]| of a device vector (create time: synchronous); NaN if anything is not finite
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ p, size_t n, unsigned* __restrict__ out) {
    float m = 0.f;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = fabsf(p[i]);
        bad = bad || !(v <= 3.0e38f);
        m = fmaxf(m, v);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (__builtin_amdgcn_ballot_w64(bad)) m = __builtin_inff();
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));  // (non-negative floats order like their bit patterns)
}
float dev_absmax(const float* p, size_t n) {
    unsigned* d = nullptr;
    unsigned hv = 0x7fc00000u;  // NaN unless everything below succeeds
    if (p && n && hipMalloc(&d, sizeof(unsigned)) == hipSuccess) {
        if (hipMemset(d, 0, sizeof(unsigned)) == hipSuccess) {
            hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, nullptr, p, n, d);
            if (hipGetLastError() != hipSuccess || hipMemcpy(&hv, d, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) hv = 0x7fc00000u;
        }
        (void)hipFree(d);
    }
    float v;
    memcpy(&v, &hv, sizeof(float));
    return v;
}

int copy_vec(after_ae* h, float** dst, const float* src, int n) {
    AE_TAKE(*dst, n);
    AFTER_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice));
    return AFTER_OK;
}

int load_snake(after_ae* h, WeightCursor& c, float** alpha, float** invb, int C) {
    const float* a = c.next();
    const float* b = c.next();
    AFTER_REQUIRE(c.ok, AFTER_E_INVALID, "autoencoder: missing snake parameters");
    AFTER_TRY(copy_vec(h, alpha, a, C));
    AE_TAKE(*invb, C);
    return snake_inv_beta(b, *invb, C, 0);
}

int load_wnconv(after_ae* h, WeightCursor& c, float** w, float** bias, int cout, int cin, int k) {
    const float* g = c.next();
    const float* v = c.next();
    const float* b = c.next();
    AFTER_REQUIRE(c.ok, AFTER_E_INVALID, "autoencoder: missing conv weights");
    AE_TAKE(*w, (size_t)cout * k * pad16(cin));
    AFTER_TRY(pack_conv_weight(v, g, *w, cout, cin, k, pad16(cin), 0));
    return copy_vec(h, bias, b, cout);
}

int load_convblock(after_ae* h, WeightCursor& c, ConvBlockW& cb, int cin, int cout, int k, int dil) {
    cb.cin = cin;
    cb.cout = cout;
    cb.k = k;
    cb.dil = dil;
    const float* gw = c.next(true);
    const float* gb = c.next(true);
    if (h->norm) {
        AFTER_REQUIRE(gw && gb, AFTER_E_INVALID, "autoencoder: GroupNorm parameters missing");
        AFTER_TRY(copy_vec(h, &cb.gn_w, gw, cin));
        AFTER_TRY(copy_vec(h, &cb.gn_b, gb, cin));
    }
    AFTER_TRY(load_snake(h, c, &cb.alpha, &cb.invb, cin));
    if (h->norm) {
        cb.gn_wmax = dev_absmax(cb.gn_w, (size_t)cin);
        cb.gn_bmax = dev_absmax(cb.gn_b, (size_t)cin);
        cb.invb_max = dev_absmax(cb.invb, (size_t)cin);
    }
    return load_wnconv(h, c, &cb.w, &cb.bias, cout, cin, k);
}

int load_resblock(after_ae* h, WeightCursor& c, ResBlockW& rb, int cin, int cout, int k, int dil) {
    AFTER_TRY(load_convblock(h, c, rb.cb0, cin, cout, k, dil));
    AFTER_TRY(load_convblock(h, c, rb.cb1, cout, cout, 1, 1));
    if (cin != cout) AFTER_TRY(load_wnconv(h, c, &rb.to_w, &rb.to_b, cout, cin, 1));
    return AFTER_OK;
}

// ------------------------------------------------------------------ conv plans (time-major path)
int make_dma(after_ae* h, DmaConv& d, const float* packed, int cin, int cout, int taps, int phases,
             int istride, int ostride, const int (*toff)[kMaxTaps], const int* ooff, int Nn_hint) {
    memset(&d.in, 0, sizeof(d.in));
    d.in.Cin = cin;
    d.in.Cout = cout;
    d.in.taps = taps;
    d.in.phases = phases;
    d.in.istride = istride;
    d.in.ostride = ostride;
    for (int p = 0; p < phases; ++p) {
        for (int t = 0; t < taps; ++t) d.in.toff[p][t] = toff[p][t];
        d.in.ooff[p] = ooff ? ooff[p] : 0;
    }
    d.in.Nn_hint = Nn_hint;
    d.in.B_hint = h->max_batch;  // the handle is (re)built for the batch it serves
    conv_tm_plan(d.in, &d.tplan);
    AFTER_REQUIRE(d.tplan.ok, AFTER_E_INVALID, "autoencoder: tap pattern outside the time-major conv path");
    d.w = h->wd.take<float>(d.tplan.w_floats);
    AFTER_REQUIRE(d.w, AFTER_E_NOMEM, "autoencoder: conv weight arena exhausted");
    AFTER_TRY(conv_tm_repack(packed, d.w, d.in, d.tplan, 0));
    d.w3 = nullptr;
    if (conv_x6_mode() && conv_x6_eligible(d.in, d.tplan) && cout >= 64 && d.tplan.K >= 192) {
        d.w3 = h->wd.take<unsigned short>(conv_x6_weight_elems(d.in, d.tplan));
        AFTER_REQUIRE(d.w3, AFTER_E_NOMEM, "autoencoder: conv weight arena exhausted");
        AFTER_TRY(conv_x6_split(d.w, d.w3, d.in, d.tplan, 0));
        // the two-piece fp16 form of the same weights (used where a GroupNorm bounds the conv's input: run_dma)
        const float wmax = dev_absmax(d.w, d.tplan.w_floats);
        const float ws = h3_scale_for(wmax);
        if (std::isfinite(wmax) && wmax * ws <= 32768.0f) {
            d.wh3 = h->wd.take<unsigned short>(conv_h3_weight_elems(d.in, d.tplan));
            AFTER_REQUIRE(d.wh3, AFTER_E_NOMEM, "autoencoder: conv weight arena exhausted");
            AFTER_TRY(conv_h3_split(d.w, d.wh3, d.in, d.tplan, ws, 0));
            d.ws = ws;
        }
    }
    return AFTER_OK;
}

double* next_stats(after_ae* h, int B) {
    double* p = h->stats_ring + (size_t)(h->stat_slot % kStatSlots) * h->stat_sub * h->max_batch * 8 * kStatWords;
    ++h->stat_slot;
    (void)B;
    return p;
}

// act(GroupNorm(x)) -> haloed scratch, then the DMA conv.  stats_in: accumulators of x (or
// nullptr: no norm); returns in *stats_out the accumulators of y when want_stats.
int run_dma(after_ae* h, hipStream_t s, const DmaConv& d, const float* x, const double* stats_in,
            const float* gamma, const float* beta, const float* alpha, const float* invb, int act,
            const float* bias, const float* res, float* y, int B, int Tin, int Tout, int Nn,
            bool want_stats, double** stats_out, float* state_base = nullptr, int x_cm = 0, int y_cm = 0,
            int stat_T = 0) {
    const int cin = d.in.Cin, cout = d.in.Cout;
    const ConvBlockW* bw = h->bound_of;  // (consumed by this call, whatever path it takes)
    h->bound_of = nullptr;
    float* state = nullptr;
    float* state_out = nullptr;
    if (h->pass_stream && state_base) {
        const int f = *h->pass_flip, slot = h->state_slot++;
        const size_t roff = (size_t)h->pass_row0 * conv_tm_halo() * conv_tm_cp(cin);  // a slot is [row][HALO][Cp] of this conv
        state = state_base + ((size_t)f * h->pass_slots + slot) * h->slot_elems + roff;
        state_out = state_base + ((size_t)(f ^ 1) * h->pass_slots + slot) * h->slot_elems + roff;
    }
    // x, res, y time-major [B][T][C] (x_cm / y_cm: the reference's [B][C][T] at the API edges)
    AFTER_REQUIRE((size_t)B * conv_tm_cp(cin) * conv_tm_rows(Tin) <= h->xp_elems, AFTER_E_CAPACITY,
                  "autoencoder: activation scratch too small");
    // Snake-only inputs (the resampling convs, the encoder tail) carry no full-tensor statistics:
    // offline their producer's epilogue has already written the activated, haloed tensor
    const float* xin = h->prepared;
    h->prepared = nullptr;
    ConvTmRun r;
    memset(&r, 0, sizeof(r));
    r.w = d.w;
    r.bias = bias;
    r.res = res;
    r.y = y;
    r.G = cout < 8 ? cout : 8;
    if (want_stats && h->norm && !h->pass_gnwin) {  // (streaming GroupNorm: window statistics, taken by the consumer)
        r.stats = next_stats(h, B);
        if (stats_out) *stats_out = r.stats;
    }
    r.B = B;
    r.Tp = conv_tm_rows(Tin);
    r.Tout = Tout;
    r.Nn = Nn;
    r.y_cm = y_cm;
    r.sub_stride = h->max_batch * 8 * kStatWords;
    static int fuse_snake = -1;  // AFTER_AE_FUSE_SNAKE=0: A/B switch (separate act_pad launches)
    if (fuse_snake < 0) {
        const char* e = getenv("AFTER_AE_FUSE_SNAKE");
        fuse_snake = e ? atoi(e) : 1;
    }
    if (fuse_snake && h->next_alpha && !h->pass_stream && (cout & 31) == 0 && d.in.ostride == 1 &&
        (size_t)B * cout * conv_tm_rows(Tout) <= h->xp_elems) {
        r.y2 = xin == h->xp2 ? h->xp : h->xp2;
        r.y2_act = ACT_SNAKE;
        r.y2_pa = h->next_alpha;
        r.y2_pb = h->next_invb;
        r.y = nullptr;  // the raw tensor has no other reader
        h->prepared = r.y2;
    }
    h->next_alpha = h->next_invb = nullptr;
    // GroupNorm -> Snake -> Conv1d(k = 1) + residual (the second conv of a ResnetBlock1d) as ONE launch where the tensor fills
    // the chip with row tiles of all channels: no activated tensor, no act_pad launch (conv_tm.hip: conv1_act_kernel)
    if (!xin && !r.y2 && d.in.taps == 1 && d.in.phases == 1 && d.in.istride == 1 && d.in.ostride == 1 && d.in.toff[0][0] == 0 &&
        cin == cout && !x_cm && !y_cm && !state && !h->pass_stream && !h->pass_gnwin && act == ACT_SNAKE && Tin == Tout && Nn == Tout &&
        stat_T == 0 && conv1_act_eligible(B, Tin, cin, cin < 8 ? cin : 8, stats_in || r.stats)) {
        Conv1ActRun c;
        memset(&c, 0, sizeof(c));
        c.x = x, c.stats_in = stats_in, c.gamma = gamma, c.beta = beta, c.act_a = alpha, c.act_b = invb;
        c.w = d.w, c.ldw = d.tplan.K, c.bias = bias, c.res = res, c.y = y, c.stats_out = r.stats;
        c.B = B, c.T = Tin, c.C = cin, c.G = cin < 8 ? cin : 8, c.act = act, c.sub_stride = r.sub_stride;
        static const bool trace_k1 = getenv("AFTER_AE_TRACE") != nullptr;
        if (trace_k1) fprintf(stderr, "ae conv: B %d T %d C %d k 1 res %d stats %d -> conv1_act (fused GroupNorm + Snake + conv)\n", B, Tin, cin, res != nullptr, r.stats != nullptr);
        return launch_conv1_act(c, s);
    }
    // MFMA-bound whole-clip launches run on the bf16 pipe (conv_x6.hip): their input is written as bf16 planes
    // ... on TWO fp16 pieces per operand where a whole-clip GroupNorm bounds the activated input (gemm_h3_pipe.h): with n elements per
    // (clip, group), |snake(GroupNorm(x))| <= sqrt(n) max|gamma| + max|beta| + max(1 / snake beta); the scale is the largest power of
    // two that keeps that bound inside fp16's range -- exact, and the fp32 accumulators are scaled back in the conv's epilogue.
    // (Snake-only inputs -- the resampling convs -- and the GroupNorm-free causal codec have no bound: they stay on three bf16 planes.)
    float hs = 0.f;
    if (!xin && d.wh3 && h->h3_mode && bw && stats_in && gamma && act == ACT_SNAKE && stat_T == 0 && !h->pass_gnwin && !h->pass_stream && !x_cm) {
        const int G = cin < 8 ? cin : 8;
        const float bound = sqrtf((float)(cin / G) * (float)Tin) * bw->gn_wmax + bw->gn_bmax + bw->invb_max;
        const float sc = h3_scale_for(bound);
        if (std::isfinite(bound) && bound * sc <= 32768.0f) hs = sc;
    }
    const bool x6 = !xin && d.w3 && !h->pass_stream && !x_cm && conv_x6_wins(r, d.in, d.tplan, hs != 0.f) &&
                    conv_x6_plane_elems(B, Tin, cin) <= h->xp3_elems;
    if (!x6) hs = 0.f;
    if (!xin) {
        ActPadTm p;
        memset(&p, 0, sizeof(p));
        p.x = x;
        p.y = h->xp;
        p.y3 = x6 ? h->xp3 : nullptr;
        p.hscale = hs;
        p.stats = stats_in;
        p.gamma = gamma;
        p.beta = beta;
        p.act_a = alpha;
        p.act_b = invb;
        p.state = state;
        p.state_out = state_out;
        p.act = act;
        p.B = B;
        p.C = cin;
        p.T = Tin;
        p.G = cin < 8 ? cin : 8;
        p.x_cm = x_cm;
        p.stat_T = stat_T;
        p.sub_stride = h->max_batch * 8 * kStatWords;
        AFTER_TRY(launch_act_pad_tm(p, s));
        xin = h->xp;
    }
    r.xp = xin;
    static const bool trace_layers = getenv("AFTER_AE_TRACE") != nullptr;  // one line per conv launch: shape and path
    if (trace_layers)
        fprintf(stderr, "ae conv: B %d Tin %d Tout %d Nn %d Cin %d Cout %d taps %d phases %d K %d res %d stats %d y2 %d -> %s\n", B, Tin, Tout, Nn, cin,
                cout, d.in.taps, d.in.phases, d.tplan.K, res != nullptr, r.stats != nullptr, r.y2 != nullptr, x6 ? "conv_x6" : "conv_tm");
    if (x6) {
        r.xp3 = h->xp3;
        r.w3 = hs != 0.f ? d.wh3 : d.w3;
        r.hscale = hs;
        r.oscale = hs != 0.f ? 1.0f / (hs * d.ws) : 0.f;
        return launch_conv_x6(r, d.in, d.tplan, s);
    }
    return launch_conv_tm(r, d.in, d.tplan, s);
}

// Would run_dma send this conv down the bf16 pipe (conv_x6.hip) if its input came through act_pad?  Decides whether a producer
// hands it the activated tensor as its own second output (fp32: the consumer then stays on the fp32 conv) or leaves the
// activation to an act_pad launch that writes bf16 planes.  Measured on the decoder's ConvTranspose phases at eight clips:
// 244 us per launch on the fp32 conv against ~130 + an act_pad of 42 -- decode 6.86 -> 6.51 ms (one clip: 1.45 -> 1.43).
bool dma_takes_x6(const after_ae* h, const DmaConv& d, int B, int Tin, int Nn, int Tout) {
    if (!d.w3 || h->pass_stream) return false;
    static const int always = [] { const char* e = getenv("AFTER_AE_FUSE_SNAKE"); return e && atoi(e) == 2; }();  // A/B: round 4's rule
    if (always) return false;
    ConvTmRun r;
    memset(&r, 0, sizeof(r));
    r.B = B, r.Tp = conv_tm_rows(Tin), r.Tout = Tout, r.Nn = Nn;
    r.G = d.in.Cout < 8 ? d.in.Cout : 8;
    if (h->norm && !h->pass_gnwin) r.stats = reinterpret_cast<double*>(h->stats_ring);  // (a non-null marker: the resampling convs accumulate statistics)
    return conv_x6_wins(r, d.in, d.tplan) && conv_x6_plane_elems(B, Tin, d.in.Cin) <= h->xp3_elems;
}

// ConvBlock1d on the DMA path
int run_convblock2(after_ae* h, hipStream_t s, const ConvBlockW& cb, const float* x,
                   const double* stats_x, float* y, const float* res, int B, int T, bool want_stats,
                   double** stats_y, float* sb = nullptr) {
    if (h->pass_gnwin && h->norm) {
        // CachedGroupNorm(stream=True): statistics of (the previous gn_P frames, this chunk), taken here
        // from the raw input; the ring then holds the newest gn_P frames
        AFTER_REQUIRE(cb.gn_ring && cb.gn_P > 0, AFTER_E_INVALID, "autoencoder: streaming GroupNorm state missing");
        const int G = cb.cin < 8 ? cb.cin : 8;
        double* st = next_stats(h, B);
        hipLaunchKernelGGL(gn_window_kernel, dim3(G, B), dim3(256), 0, s, x, cb.gn_ring, st, T, cb.cin, G, cb.gn_P,
                           cb.gn_head);
        AFTER_HIP_CHECK(hipGetLastError());
        if (T < cb.gn_P) cb.gn_head = (cb.gn_head + T) % cb.gn_P;
        else cb.gn_head = 0;
        return run_dma(h, s, cb.d, x, st, cb.gn_w, cb.gn_b, cb.alpha, cb.invb, ACT_SNAKE, cb.bias, res, y, B, T,
                       T, T, false, nullptr, sb, 0, 0, cb.gn_P + T);
    }
    h->bound_of = h->norm ? &cb : nullptr;
    return run_dma(h, s, cb.d, x, h->norm ? stats_x : nullptr, cb.gn_w, cb.gn_b, cb.alpha, cb.invb,
                   ACT_SNAKE, cb.bias, res, y, B, T, T, T, want_stats, stats_y, sb);
}

// ResnetBlock1d on the DMA path; *stats carries the accumulators of the block input in and of
// the block output out
// (na, nb): Snake parameters of a resampling conv that consumes this block's output -- its
// activated input is then written by the block's last conv (see run_dma)
int run_resblock2(after_ae* h, hipStream_t s, const ResBlockW& rb, const float* bx, float* bt,
                  float* by, int B, int T, double** stats, float* sb = nullptr, const float* na = nullptr,
                  const float* nb = nullptr) {
    const float* res = bx;
    const float* sx = bx;  // the shortcut branch's input
    if (h->pass_cached && rb.delay > 0) {
        // cached_conv.AlignBranches(net, to_out, delays = [block1's delay, 0]): the shortcut sees its input
        // rb.delay frames late (the second haloed scratch is free: no fused second outputs while streaming)
        const int C = rb.cb0.cin;
        AFTER_REQUIRE((size_t)B * T * C <= h->xp_elems, AFTER_E_CAPACITY, "autoencoder: delay scratch too small");
        const size_t total = (size_t)B * (T + rb.delay) * C;
        hipLaunchKernelGGL(delay_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, bx,
                           rb.dline[rb.dflip], rb.dline[rb.dflip ^ 1], h->xp2, T, C, rb.delay, total);
        AFTER_HIP_CHECK(hipGetLastError());
        rb.dflip ^= 1;
        sx = res = h->xp2;
    }
    if (rb.to_w) {  // 1x1 shortcut: no temporal context, no state
        AFTER_TRY(run_dma(h, s, rb.to_d, sx, nullptr, nullptr, nullptr, nullptr, nullptr, ACT_NONE,
                          rb.to_b, nullptr, by, B, T, T, T, false, nullptr));
        res = by;
    }
    double* st1 = nullptr;
    AFTER_TRY(run_convblock2(h, s, rb.cb0, bx, *stats, bt, nullptr, B, T, true, &st1, sb));
    double* st2 = nullptr;
    h->next_alpha = na;
    h->next_invb = nb;
    AFTER_TRY(run_convblock2(h, s, rb.cb1, bt, st1, by, res, B, T, true, &st2));
    *stats = st2;
    return AFTER_OK;
}

int begin_pass(after_ae* h, hipStream_t s) {
    const bool stateful = h->streaming || h->enc_cached || h->dec_gn_frames > 0;
    AFTER_REQUIRE(!(h->in_pass && stateful), AFTER_E_INVALID,
                  "autoencoder: an earlier streaming pass failed half way, the stream state is desynchronised: call "
                  "after_ae_reset_state");
    h->in_pass = stateful;  // (a stateless pass that fails half way leaves nothing behind)
    h->stat_slot = 0;
    h->state_slot = 0;
    h->pass_row0 = 0;
    h->prepared = nullptr;
    h->next_alpha = h->next_invb = nullptr;
    if (h->norm)
        AFTER_HIP_CHECK(hipMemsetAsync(h->stats_ring, 0,
                                       (size_t)kStatSlots * h->stat_sub * h->max_batch * 8 * kStatWords * sizeof(double), s));
    return AFTER_OK;
}

int check_ae(after_ae* h, int B, long long samples) {
    AFTER_REQUIRE(h, AFTER_E_INVALID, "null handle");
    AFTER_REQUIRE(B > 0 && samples > 0, AFTER_E_INVALID, "empty batch");
    AFTER_REQUIRE(B <= h->max_batch && samples <= h->max_samples, AFTER_E_CAPACITY,
                  "B=%d samples=%lld exceed max_batch=%d max_samples=%d", B, samples, h->max_batch,
                  h->max_samples);
    AFTER_REQUIRE(samples % h->ratio == 0, AFTER_E_INVALID,
                  "length %lld is not a multiple of the codec ratio %d", samples, h->ratio);
    return AFTER_OK;
}

// tm: multiband in the time-major layout of the conv path ([B][L/M][M]) instead of [B][M][L/M]
int pqmf_forward(after_ae* h, hipStream_t s, const float* x, float* mb, int B, int L,
                 float* state = nullptr, bool tm = false) {
    const int M = h->M, K = h->pq_fk;
    const int cs = tm ? 1 : L / M, ts = tm ? M : 1;
    const int Q = (K + M - 1) / M;
    const int pl = h->causal ? K - 1 : (K - 1) / 2;
    if (M == 16) {
        const size_t lds = ((size_t)M * Q * M + (size_t)M * (PQ_BT + Q + 1)) * sizeof(float);
        static LdsAttr attr;
        AFTER_TRY(ensure_lds_attr(attr, reinterpret_cast<const void*>(pqmf_forward_kernel<16>), lds));
        hipLaunchKernelGGL(pqmf_forward_kernel<16>, dim3(cdiv(L / M, PQ_BT), B), dim3(256), lds, s, x,
                           h->pq_fwp, mb, L, Q, pl, state, cs, ts);
    } else {
        hipLaunchKernelGGL(pqmf_forward_generic_kernel, dim3((unsigned)cdivll((long long)L, 256), B),
                           dim3(256), 0, s, x, h->pq_fw, mb, L, M, K, pl, state, cs, ts);
    }
    AFTER_HIP_CHECK(hipGetLastError());
    if (state && K > 1) {  // keep the last K - 1 input samples of every clip
        const int S = K - 1;
        AFTER_REQUIRE(L >= S && (L / M) % 2 == 0, AFTER_E_INVALID,
                      "pqmf streaming: chunk of %d samples too short / odd frame count", L);
        AFTER_HIP_CHECK(hipMemcpy2DAsync(state, (size_t)S * sizeof(float), x + (L - S),
                                         (size_t)L * sizeof(float), (size_t)S * sizeof(float), B,
                                         hipMemcpyDeviceToDevice, s));
    }
    return AFTER_OK;
}

int pqmf_inverse(after_ae* h, hipStream_t s, const float* y, float* audio, int B, int Tm, int gated,
                 int ychan, float* zstate = nullptr, bool tm = false) {
    const int M = h->M, K = h->pq_ik;
    const int cs = tm ? 1 : Tm, ts = tm ? ychan : 1;
    const int pl = h->causal ? K - 1 : (K - 1) / 2;
    if (M == 16) {
        const size_t lds = ((size_t)M * K * M + (size_t)M * (256 + K)) * sizeof(float);
        static LdsAttr attr;
        AFTER_TRY(ensure_lds_attr(attr, reinterpret_cast<const void*>(pqmf_inverse_kernel<16>), lds));
        hipLaunchKernelGGL(pqmf_inverse_kernel<16>, dim3(cdiv(Tm, 256), B), dim3(256), lds, s, y, h->pq_iwp,
                           audio, Tm, K, pl, gated, ychan, zstate, cs, ts);
    } else {
        hipLaunchKernelGGL(pqmf_inverse_generic_kernel, dim3((unsigned)cdivll((long long)Tm * M, 256), B),
                           dim3(256), 0, s, y, h->pq_iw, audio, Tm, M, K, pl, gated, ychan, zstate, cs, ts);
    }
    AFTER_HIP_CHECK(hipGetLastError());
    if (zstate && K > 1) {
        const int S = K - 1;
        AFTER_REQUIRE(Tm >= S && Tm % 2 == 0, AFTER_E_INVALID,
                      "pqmf streaming: chunk of %d frames too short / odd", Tm);
        const int total = B * M * S;
        hipLaunchKernelGGL(pqmf_istate_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, y, zstate, Tm, M, S,
                           gated, ychan, total, cs, ts);
        AFTER_HIP_CHECK(hipGetLastError());
    }
    return AFTER_OK;
}

}  // namespace

extern "C" int after_ae_create(const after_ae_cfg* cfg, const float* const* weights, int n_weights,
                               int max_batch, int max_samples, after_ae** out) {
    AFTER_REQUIRE(cfg && weights && out, AFTER_E_INVALID, "null argument");
    *out = nullptr;
    AFTER_REQUIRE(cfg->n_stages >= 1 && cfg->n_stages <= AFTER_AE_MAX_STAGES &&
                      cfg->n_dilations >= 1 && cfg->n_dilations <= AFTER_AE_MAX_STAGES,
                  AFTER_E_INVALID, "autoencoder: bad stage / dilation count");
    // (pqmf_bands = 1: no filter bank -- the reference's DummyIdentity, SimpleNetsStream.py:854-858: the codec runs on the audio
    //  samples themselves; here the one-band bank with the single tap 1.0 on the generic PQMF kernels)
    AFTER_REQUIRE(cfg->pqmf_bands >= 1 && cfg->kernel_size % 2 == 1 && cfg->kernel_size <= 7,
                  AFTER_E_INVALID, "autoencoder: pqmf_bands >= 1 and odd kernel_size <= 7 required");
    AFTER_REQUIRE(max_batch > 0 && max_samples > 0, AFTER_E_INVALID, "bad capacities");
    // one statistics slot per normalised conv of a pass (the larger of encode / decode):
    // 2 per ResnetBlock + 1 per resampling conv + stem / synth
    AFTER_REQUIRE(2 * (cfg->n_stages * cfg->n_dilations + 1) + cfg->n_stages + 4 <= kStatSlots,
                  AFTER_E_CAPACITY, "autoencoder: %d stages x %d dilations need more than %d statistics slots",
                  cfg->n_stages, cfg->n_dilations, kStatSlots);
    after_ae* h = new (std::nothrow) after_ae();
    AFTER_REQUIRE(h, AFTER_E_NOMEM, "out of host memory");
    h->cfg = *cfg;
    h->M = cfg->pqmf_bands;
    h->causal = cfg->causal != 0;
    h->norm = cfg->use_norm != 0;
    {
        const char* e = getenv("AFTER_CONV_H3");  // 0: the GroupNorm-bounded convs of the bf16 pipe stay on three bf16 planes (A/B switch)
        h->h3_mode = e ? atoi(e) != 0 : 1;
    }
    h->max_batch = max_batch;
    int ratio = h->M;
    for (int i = 0; i < cfg->n_stages; ++i) ratio *= cfg->factors[i];
    h->ratio = ratio;
    h->max_samples = (max_samples / ratio) * ratio;
    const int n = cfg->n_stages, nd = cfg->n_dilations, k = cfg->kernel_size, C0 = cfg->channels;
    const int out_ch = cfg->use_loudness ? 2 * h->M : h->M;

    auto fail = [&](int rc) {
        after_ae_destroy(h);
        return rc;
    };
    // weight arena size: walk the architecture
    size_t wf = 0;
    auto cbsz = [&](int cin, int cout, int kk) {
        return (size_t)cout * kk * pad16(cin) + cout + 4 * (size_t)cin + 6 * 64;
    };
    {
        int c = C0 * cfg->multipliers[0];
        wf += cbsz(h->M, c, k) + cbsz(c, c, 1) + cbsz(h->M, c, 1);
        for (int i = 0; i < n; ++i) {
            c = C0 * cfg->multipliers[i];
            const int cn = C0 * cfg->multipliers[i + 1];
            wf += nd * (cbsz(c, c, k) + cbsz(c, c, 1)) + cbsz(c, cn, 2 * cfg->factors[i]);
        }
        wf += cbsz(C0 * cfg->multipliers[n], cfg->encoder_out_channels > 0 ? cfg->encoder_out_channels : cfg->z_channels, 3);
        c = C0 * cfg->dec_multipliers[0];
        wf += cbsz(cfg->z_channels, c, k);
        for (int i = 0; i < n; ++i) {
            c = C0 * cfg->dec_multipliers[i];
            const int cn = C0 * cfg->dec_multipliers[i + 1];
            wf += 2 * cbsz(c, cn, 2 * cfg->factors[n - 1 - i]) + nd * (cbsz(cn, cn, k) + cbsz(cn, cn, 1));
        }
        c = C0 * cfg->dec_multipliers[n];
        wf += cbsz(c, out_ch, k) + cbsz(out_ch, out_ch, 1);
        if (cfg->use_noise) wf += cbsz(c, 128, 3) + cbsz(128, 128, 3) + cbsz(128, h->M * 5, 3);
        wf += (size_t)h->M * 1024 + (size_t)h->M * h->M * 64;
    }
    int rc = h->wa.init(wf * sizeof(float) + (1 << 20));
    if (rc != AFTER_OK) return fail(rc);

    WeightCursor cur{weights, n_weights};
#define AE_TRY(expr)                        \
    do {                                    \
        int rc2__ = (expr);                 \
        if (rc2__ != AFTER_OK) return fail(rc2__); \
    } while (0)
    // PQMF banks (pqmf.py:258-280): forward [M,1,Kf], inverse [M,M,Ki]; the kernel sizes
    // follow from the arena request below (Kf = 32 M + 1, Ki = 2 M + 1 for hk of 32 M taps)
    h->pq_fk = h->M > 1 ? 32 * h->M + 1 : 1;
    h->pq_ik = h->M > 1 ? 2 * h->M + 1 : 1;
    {
        const float* fw = cur.next();
        const float* iw = cur.next();
        if (!cur.ok) {
            set_error("autoencoder: missing pqmf filters");
            return fail(AFTER_E_INVALID);
        }
        AE_TRY(copy_vec(h, &h->pq_fw, fw, h->M * h->pq_fk));
        AE_TRY(copy_vec(h, &h->pq_iw, iw, h->M * h->M * h->pq_ik));
        const int Q = (h->pq_fk + h->M - 1) / h->M;
        h->pq_fwp = h->wa.take<float>((size_t)h->M * Q * h->M);
        h->pq_iwp = h->wa.take<float>((size_t)h->M * h->pq_ik * h->M);
        if (!h->pq_fwp || !h->pq_iwp) return fail(AFTER_E_NOMEM);
        const int nmax = h->M * h->M * (Q > h->pq_ik ? Q : h->pq_ik);
        hipLaunchKernelGGL(pqmf_relayout_kernel, dim3(cdiv(nmax, 256)), dim3(256), 0, 0, h->pq_fw,
                           h->pq_iw, h->pq_fwp, h->pq_iwp, h->M, h->pq_fk, Q, h->pq_ik);
        if (hipGetLastError() != hipSuccess) return fail(AFTER_E_HIP);
    }
    // encoder (SimpleNetsStream.py:400-459)
    AE_TRY(load_resblock(h, cur, h->enc_stem, h->M, C0 * cfg->multipliers[0], k, 1));
    h->enc_res.resize(n);
    h->enc_down.resize(n);
    for (int i = 0; i < n; ++i) {
        const int c = C0 * cfg->multipliers[i], cn = C0 * cfg->multipliers[i + 1];
        h->enc_res[i].resize(nd);
        for (int j = 0; j < nd; ++j)
            AE_TRY(load_resblock(h, cur, h->enc_res[i][j], c, c, k, cfg->dilations[j]));
        ResampleW& d = h->enc_down[i];
        d.cin = c;
        d.cout = cn;
        d.f = cfg->factors[i];
        AE_TRY(load_snake(h, cur, &d.alpha, &d.invb, c));
        AE_TRY(load_wnconv(h, cur, &d.w, &d.bias, cn, c, 2 * d.f));
    }
    {
        const int c = C0 * cfg->multipliers[n];
        AE_TRY(load_snake(h, cur, &h->enc_tail_alpha, &h->enc_tail_invb, c));
        h->enc_tail.cin = c;
        const int zo = cfg->encoder_out_channels > 0 ? cfg->encoder_out_channels : cfg->z_channels;  // VAE: 2 Z
        h->enc_tail.cout = zo;
        h->enc_tail.k = 3;
        AE_TRY(load_wnconv(h, cur, &h->enc_tail.w, &h->enc_tail.bias, zo, c, 3));
    }
    // decoder (SimpleNetsStream.py:552-651)
    h->dec_head.cin = cfg->z_channels;
    h->dec_head.cout = C0 * cfg->dec_multipliers[0];
    h->dec_head.k = k;
    AE_TRY(load_wnconv(h, cur, &h->dec_head.w, &h->dec_head.bias, h->dec_head.cout, cfg->z_channels, k));
    h->dec_up.resize(n);
    h->dec_res.resize(n);
    for (int i = 0; i < n; ++i) {
        const int c = C0 * cfg->dec_multipliers[i], cn = C0 * cfg->dec_multipliers[i + 1];
        ResampleW& u = h->dec_up[i];
        u.cin = c;
        u.cout = cn;
        u.f = cfg->factors[n - 1 - i];
        if (!(u.f >= 2 && u.f <= kMaxPhases && u.f % 2 == 0)) {
            set_error("autoencoder: upsampling factor %d unsupported (even, <= %d)", u.f, kMaxPhases);
            return fail(AFTER_E_INVALID);
        }
        AE_TRY(load_snake(h, cur, &u.alpha, &u.invb, c));
        {
            const float* g = cur.next();
            const float* v = cur.next();
            const float* b = cur.next();
            if (!cur.ok) {
                set_error("autoencoder: missing transposed conv weights");
                return fail(AFTER_E_INVALID);
            }
            u.w = h->wa.take<float>((size_t)u.f * cn * 2 * pad16(c));
            if (!u.w) return fail(AFTER_E_NOMEM);
            AE_TRY(pack_convT_weight(v, g, u.w, c, cn, u.f, pad16(c), 0));
            if (h->causal && !h->norm) {  // streaming twin: padding 0 (cached_conv overlap-add)
                u.w_stream = h->wa.take<float>((size_t)u.f * cn * 2 * pad16(c));
                if (!u.w_stream) return fail(AFTER_E_NOMEM);
                AE_TRY(pack_convT_weight(v, g, u.w_stream, c, cn, u.f, pad16(c), 0, 0));
            }
            AE_TRY(copy_vec(h, &u.bias, b, cn));
        }
        h->dec_res[i].resize(nd);
        for (int j = 0; j < nd; ++j)
            AE_TRY(load_resblock(h, cur, h->dec_res[i][j], cn, cn, k, cfg->dilations[j]));
    }
    {
        const int c = C0 * cfg->dec_multipliers[n];
        AE_TRY(load_convblock(h, cur, h->synth0, c, out_ch, k, 1));
        AE_TRY(load_convblock(h, cur, h->synth1, out_ch, out_ch, 1, 1));
        if (cfg->use_noise) {  // NoiseGenerator(in_size = c, data_size = M, ratios = [2, 2, 2], noise_bands = 5, hidden 128)
            const int ch[4] = {c, 128, 128, h->M * 5};
            for (int i = 0; i < 3; ++i) {
                const float* w = cur.next();
                const float* b = cur.next();
                if (!cur.ok) break;
                h->noise.cin[i] = ch[i];
                h->noise.cout[i] = ch[i + 1];
                h->noise.w[i] = h->wa.take<float>((size_t)ch[i + 1] * 3 * pad16(ch[i]));
                if (!h->noise.w[i]) {
                    set_error("autoencoder: weight arena exhausted");
                    return fail(AFTER_E_NOMEM);
                }
                AE_TRY(pack_conv_weight(w, nullptr, h->noise.w[i], ch[i + 1], ch[i], 3, pad16(ch[i]), 0));
                AE_TRY(copy_vec(h, &h->noise.bias[i], b, ch[i + 1]));
            }
        }
    }
    if (!cur.ok || cur.i != n_weights) {
        set_error("autoencoder: expected %d weight tensors, got %d", cur.i, n_weights);
        return fail(AFTER_E_INVALID);
    }
    // ---- per-conv geometry + GEMM-operand weights
    h->stat_sub = conv_tm_stat_sub();
    {
        AE_TRY(h->wd.init((size_t)(wf * 2.0) * sizeof(float) + (size_t)(wf * 2.0) * (3 + 2) * sizeof(unsigned short) + (16 << 20)));
        const size_t Tm = h->max_samples / h->M;
        auto plan_conv = [&](DmaConv& d, const float* packed, int cin, int cout, int kk, int dil,
                             size_t T) -> int {
            int toff[kMaxPhases][kMaxTaps] = {};
            const int pl = conv_left_pad(kk, dil, h->causal);
            for (int t = 0; t < kk; ++t) toff[0][t] = t * dil - pl;
            return make_dma(h, d, packed, cin, cout, kk, 1, 1, 1, toff, nullptr, (int)T);
        };
        auto plan_res = [&](ResBlockW& rb, size_t T) -> int {
            AFTER_TRY(plan_conv(rb.cb0.d, rb.cb0.w, rb.cb0.cin, rb.cb0.cout, rb.cb0.k, rb.cb0.dil, T));
            AFTER_TRY(plan_conv(rb.cb1.d, rb.cb1.w, rb.cb1.cin, rb.cb1.cout, 1, 1, T));
            if (rb.to_w) AFTER_TRY(plan_conv(rb.to_d, rb.to_w, rb.cb0.cin, rb.cb0.cout, 1, 1, T));
            return AFTER_OK;
        };
        size_t T = Tm;
        AE_TRY(plan_res(h->enc_stem, T));
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < nd; ++j) AE_TRY(plan_res(h->enc_res[i][j], T));
            ResampleW& d = h->enc_down[i];
            int toff[kMaxPhases][kMaxTaps] = {};
            const int pl = conv_left_pad(2 * d.f, 1, h->causal);
            for (int t = 0; t < 2 * d.f; ++t) toff[0][t] = t - pl;
            AE_TRY(make_dma(h, d.d, d.w, d.cin, d.cout, 2 * d.f, 1, d.f, 1, toff, nullptr, (int)(T / d.f)));
            T /= d.f;
        }
        AE_TRY(plan_conv(h->enc_tail.d, h->enc_tail.w, h->enc_tail.cin, h->enc_tail.cout, 3, 1, T));
        T = h->max_samples / h->ratio;
        AE_TRY(plan_conv(h->dec_head.d, h->dec_head.w, h->dec_head.cin, h->dec_head.cout, h->dec_head.k, 1, T));
        for (int i = 0; i < n; ++i) {
            ResampleW& u = h->dec_up[i];
            int toff[kMaxPhases][kMaxTaps] = {};
            int ooff[kMaxPhases] = {};
            for (int r = 0; r < u.f; ++r) {
                const int cc = (r + u.f / 2) / u.f;
                toff[r][0] = cc - 1;
                toff[r][1] = cc;
                ooff[r] = r;
            }
            AE_TRY(make_dma(h, u.d, u.w, u.cin, u.cout, 2, u.f, 1, u.f, toff, ooff, (int)T));
            if (u.w_stream) {
                for (int r = 0; r < u.f; ++r) {
                    toff[r][0] = -1;
                    toff[r][1] = 0;
                }
                AE_TRY(make_dma(h, u.d_stream, u.w_stream, u.cin, u.cout, 2, u.f, 1, u.f, toff, ooff, (int)T));
            }
            T *= u.f;
            for (int j = 0; j < nd; ++j) AE_TRY(plan_res(h->dec_res[i][j], T));
        }
        AE_TRY(plan_conv(h->synth0.d, h->synth0.w, h->synth0.cin, h->synth0.cout, h->synth0.k, 1, T));
        AE_TRY(plan_conv(h->synth1.d, h->synth1.w, h->synth1.cin, h->synth1.cout, 1, 1, T));
        if (cfg->use_noise) {  // cc.Conv1d(k = 3, stride = 2, padding = get_padding(3, 2) = (1, 1)): y[n] = sum_t w[t] x[2 n + t - 1]
            AFTER_REQUIRE(!h->causal, AFTER_E_INVALID, "autoencoder: use_noise with causal padding is not built");
            size_t Tn = T;
            for (int i = 0; i < 3; ++i) {
                int toff[kMaxPhases][kMaxTaps] = {};
                for (int t = 0; t < 3; ++t) toff[0][t] = t - 1;
                AE_TRY(make_dma(h, h->noise.d[i], h->noise.w[i], h->noise.cin[i], h->noise.cout[i], 3, 1, 2, 1, toff, nullptr,
                                (int)(Tn / 2)));
                Tn /= 2;
            }
        }
    }
#undef AE_TRY

    // workspaces: three rotating activation buffers of the largest C x T footprint
    size_t elems = 0;
    int cmax = h->M;
    {
        const size_t Tm = h->max_samples / h->M;
        size_t T = Tm;
        auto upd = [&](int c, size_t t) {
            elems = (size_t)c * t > elems ? (size_t)c * t : elems;
            cmax = c > cmax ? c : cmax;
        };
        upd(out_ch, Tm);
        for (int i = 0; i <= n; ++i) {
            upd(C0 * cfg->multipliers[i], T);
            if (i < n) T /= cfg->factors[i];
        }
        T = h->max_samples / h->ratio;
        for (int i = 0; i <= n; ++i) {
            upd(C0 * cfg->dec_multipliers[i], T);
            if (i < n) {
                T *= cfg->factors[n - 1 - i];
                upd(C0 * cfg->dec_multipliers[i + 1], T);
            }
        }
    }
    h->buf_elems = elems * max_batch;
    h->cmax = cmax;
    // activated + haloed scratch: largest C x padded-row footprint
    size_t xpe = 0;
    {
        const size_t Tm = h->max_samples / h->M;
        size_t T = Tm;
        auto upd = [&](int c, size_t t) {
            const size_t e2 = (size_t)conv_tm_cp(c) * conv_tm_rows((int)t);
            xpe = e2 > xpe ? e2 : xpe;
        };
        upd(h->M, Tm);
        upd(out_ch, Tm);
        if (cfg->use_noise) {
            upd(128, Tm / 2);
            upd(128, Tm / 4);
        }
        for (int i = 0; i <= n; ++i) {
            upd(C0 * cfg->multipliers[i], T);
            if (i < n) T /= cfg->factors[i];
        }
        T = h->max_samples / h->ratio;
        upd(cfg->encoder_out_channels > cfg->z_channels ? cfg->encoder_out_channels : cfg->z_channels, T);
        for (int i = 0; i <= n; ++i) {
            upd(C0 * cfg->dec_multipliers[i], T);
            if (i < n) {
                T *= cfg->factors[n - 1 - i];
                upd(C0 * cfg->dec_multipliers[i + 1], T);
            }
        }
    }
    h->xp_elems = xpe * max_batch + 4096;
    h->xp3_elems = 3 * (h->xp_elems + (size_t)max_batch * 16 * conv_tm_cp(h->cmax));
    const size_t TmN = h->max_samples / h->M;
    const size_t nz_elems[4] = {cfg->use_noise ? (size_t)max_batch * (TmN / 2) * 128 : 0, cfg->use_noise ? (size_t)max_batch * (TmN / 4) * 128 : 0,
                                cfg->use_noise ? (size_t)max_batch * (TmN / 8) * h->M * 5 : 0, cfg->use_noise ? (size_t)max_batch * TmN * h->M : 0};
    rc = h->ws.init(3 * h->buf_elems * sizeof(float) + 16384 + 2 * h->xp_elems * sizeof(float) + h->xp3_elems * sizeof(unsigned short) + (nz_elems[0] + nz_elems[1] + nz_elems[2] + nz_elems[3] + 1024) * sizeof(float) + (size_t)kStatSlots * h->stat_sub * max_batch * 8 * kStatWords * sizeof(double));
    if (rc != AFTER_OK) return fail(rc);
    for (int i = 0; i < 3; ++i) h->buf[i] = h->ws.take<float>(h->buf_elems);
    h->xp = h->ws.take<float>(h->xp_elems);
    h->xp2 = h->ws.take<float>(h->xp_elems);
    h->xp3 = h->ws.take<unsigned short>(h->xp3_elems);
    if (cfg->use_noise) {
        for (int i = 0; i < 3; ++i) h->nz[i] = h->ws.take<float>(nz_elems[i]);
        h->nadd = h->ws.take<float>(nz_elems[3]);
        if (!h->nadd) return fail(AFTER_E_NOMEM);
    }
    h->stats_ring = h->ws.take<double>((size_t)kStatSlots * h->stat_sub * max_batch * 8 * kStatWords);
    if (!h->buf[2] || !h->xp || !h->xp2 || !h->xp3 || !h->stats_ring) return fail(AFTER_E_NOMEM);
    if (hipDeviceSynchronize() != hipSuccess) {
        set_error("autoencoder: device initialisation failed");
        return fail(AFTER_E_HIP);
    }
    *out = h;
    return AFTER_OK;
}

extern "C" void after_ae_destroy(after_ae* h) {
    if (!h) return;
    h->sa.release();
    h->wd.release();
    h->sn.release();
    h->sg.release();
    h->wa.release();
    h->ws.release();
    delete h;
}

extern "C" int after_ae_ratio(const after_ae* h) { return h ? h->ratio : 0; }

// ---- streaming: cached_conv semantics for the causal, norm-free codec
// (export_autoencoder.py:293-303: `cc.use_cached_conv(True)` twin of the causal model).
namespace {

template <class F>
void for_each_enc_resblock(after_ae* h, F fn) {
    fn(h->enc_stem, 0);
    for (size_t i = 0; i < h->enc_res.size(); ++i)
        for (ResBlockW& rb : h->enc_res[i]) fn(rb, (int)i);
}

// decoder ConvBlock1d's in traversal order; stage = UpsampleBlock1d index (n_stages: the synth blocks)
template <class F>
void for_each_dec_convblock(after_ae* h, F fn) {
    for (size_t i = 0; i < h->dec_res.size(); ++i)
        for (ResBlockW& rb : h->dec_res[i]) {
            fn(rb.cb0, (int)i);
            fn(rb.cb1, (int)i);
        }
    fn(h->synth0, (int)h->dec_res.size());
    fn(h->synth1, (int)h->dec_res.size());
}

// phase-0 taps of a conv <- `toff` (weights do not depend on them); keeps the offline ones for restore
int set_taps(DmaConv& d, const int* toff, bool save) {
    for (int t = 0; t < d.in.taps; ++t) {
        if (save) d.toff_offline[t] = d.in.toff[0][t];
        d.in.toff[0][t] = toff[t];
    }
    conv_tm_plan(d.in, &d.tplan);
    AFTER_REQUIRE(d.tplan.ok, AFTER_E_INVALID, "autoencoder: cached conv context exceeds the %d-frame halo",
                  conv_tm_halo());
    return AFTER_OK;
}

int right_pad(int k, int dil) { return k == 1 ? 0 : ((k - 1) * dil + 1) / 2; }  // cached_conv.get_padding, centred

}  // namespace

extern "C" int after_ae_reset_state(after_ae* h, void* stream) {
    AFTER_REQUIRE(h, AFTER_E_INVALID, "null handle");
    AFTER_REQUIRE(h->sa.base || h->sn.base || h->sg.base, AFTER_E_INVALID, "autoencoder: streaming was never enabled");
    h->in_pass = false;
    h->enc_flip = h->dec_flip = h->nc_flip = 0;
    h->enc_flip1 = 0;
    h->enc_rows = h->enc_rows1 = h->dec_rows = h->nc_rows = 0;
    if (h->sa.base) AFTER_HIP_CHECK(hipMemsetAsync(h->sa.base, 0, h->sa.off, (hipStream_t)stream));
    if (h->sn.base) {
        AFTER_HIP_CHECK(hipMemsetAsync(h->sn.base, 0, h->sn.off, (hipStream_t)stream));
        for_each_enc_resblock(h, [](ResBlockW& rb, int) {
            rb.dflip = 0;
            rb.cb0.gn_head = rb.cb1.gn_head = 0;
        });
    }
    if (h->sg.base) {
        AFTER_HIP_CHECK(hipMemsetAsync(h->sg.base, 0, h->sg.off, (hipStream_t)stream));
        for_each_dec_convblock(h, [](ConvBlockW& cb, int) { cb.gn_head = 0; });
    }
    return AFTER_OK;
}

// The streaming twin of a NON-causal codec's encoder (export_autoencoder.py:305-312: an Encoder1d built
// under cc.use_cached_conv(True) with CachedGroupNorm.stream = True; PQMF, bottleneck and decoder stay
// the offline modules).  cached_conv.CachedConv1d turns the centred padding (l, r) of every conv into
// l + r frames of left context -- here: the same weights with all taps shifted to the past and the
// activated context kept in HBM like the causal codec's -- and accounts for the r frames of lag:
//   ResnetBlock1d   AlignBranches delays the shortcut's input by block1's r (SimpleNetsStream.py:236-249)
//   Downsample1d    stride_delay = (f - (r + cd) % f) % f extra frames of input delay so that the
//                   running delay stays a whole number of output frames; cd <- (r + stride_delay + cd) / f
// gn_window_samples: CachedGroupNorm's window ("automatic" in the reference = the length of the first
// call after construction, 131072 samples in the export script), in samples of the audio stream.
extern "C" int after_ae_enable_encoder_streaming(after_ae* h, int enable, int gn_window_samples) {
    AFTER_REQUIRE(h, AFTER_E_INVALID, "null handle");
    const after_ae_cfg& c = h->cfg;
    const int n = c.n_stages, k = c.kernel_size;
    if (!enable) {
        if (h->enc_cached) {
            for_each_enc_resblock(h, [&](ResBlockW& rb, int) { (void)set_taps(rb.cb0.d, rb.cb0.d.toff_offline, false); });
            for (ResampleW& d : h->enc_down) (void)set_taps(d.d, d.d.toff_offline, false);
            (void)set_taps(h->enc_tail.d, h->enc_tail.d.toff_offline, false);
        }
        h->enc_cached = false;
        h->nc_rows = 0;  // (the stream set ends with the cached encoder: a later enable starts from zeroed contexts)
        return AFTER_OK;
    }
    AFTER_REQUIRE(!h->causal, AFTER_E_INVALID,
                  "autoencoder: the causal codec streams through after_ae_enable_streaming");
    AFTER_REQUIRE(!h->streaming, AFTER_E_INVALID, "autoencoder: already streaming");
    AFTER_REQUIRE(!h->norm || (gn_window_samples > 0 && gn_window_samples % h->ratio == 0), AFTER_E_INVALID,
                  "autoencoder: CachedGroupNorm needs a window that is a multiple of %d samples (got %d)", h->ratio,
                  gn_window_samples);
    if (h->enc_cached) {
        AFTER_REQUIRE(!h->norm || gn_window_samples == h->gn_window, AFTER_E_INVALID,
                      "autoencoder: GroupNorm window is fixed at %d samples", h->gn_window);
        return AFTER_OK;
    }
    // ---- delays and cached tap patterns, in the order the reference builds the modules
    int rc = AFTER_OK, cd = 0;
    int toff[kMaxTaps];
    for_each_enc_resblock(h, [&](ResBlockW& rb, int) {
        rb.delay = right_pad(rb.cb0.k, rb.cb0.dil);
        for (int t = 0; t < rb.cb0.k; ++t) toff[t] = (t - (rb.cb0.k - 1)) * rb.cb0.dil;
        const int r2 = set_taps(rb.cb0.d, toff, true);
        rc = rc ? rc : r2;
    });
    if (rc) return rc;
    cd = h->enc_stem.delay;
    size_t fl = 0;  // floats of state
    for (int i = 0; i < n; ++i) {
        for (const ResBlockW& rb : h->enc_res[i]) cd += rb.delay;
        ResampleW& d = h->enc_down[i];
        const int f = d.f, r = f;                      // get_padding(2f) = (f - 1, f)
        const int sd = (f - (r + cd) % f) % f;
        cd = (r + sd + cd) / f;
        for (int t = 0; t < 2 * f; ++t) toff[t] = t - (2 * f - 1) - sd;
        AFTER_TRY(set_taps(d.d, toff, true));
    }
    for (int t = 0; t < 3; ++t) toff[t] = t - 2;
    AFTER_TRY(set_taps(h->enc_tail.d, toff, true));
    h->enc_delay = cd + right_pad(3, 1);
    (void)k;
    // ---- state: activated conv contexts, delay lines, GroupNorm rings
    const int nd = c.n_dilations;
    const int enc_slots = 1 + n * nd + n + 1;
    h->nc_slots = enc_slots;
    h->slot_elems = (size_t)h->max_batch * conv_tm_cp(h->cmax) * conv_tm_halo();
    auto pad64 = [](size_t v) { return (v + 63) & ~(size_t)63; };
    fl = pad64(2 * enc_slots * h->slot_elems);
    for_each_enc_resblock(h, [&](ResBlockW& rb, int stage) {
        fl += 2 * pad64((size_t)h->max_batch * rb.delay * rb.cb0.cin);
        if (h->norm) {
            int rate = h->M;  // samples per frame at this block
            for (int q = 0; q < stage; ++q) rate *= c.factors[q];
            // (stage = index of the DownsampleBlock1d holding the block; the stem and stage 0 share the rate)
            rb.cb0.gn_P = rb.cb1.gn_P = gn_window_samples / rate;
            fl += 2 * pad64((size_t)h->max_batch * 8 * rb.cb0.gn_P * 2);
        }
    });
    AFTER_TRY(h->sn.init(fl * sizeof(float) + (64 << 10)));
    h->nc_state = h->sn.take<float>(2 * enc_slots * h->slot_elems);
    bool ok = h->nc_state != nullptr;
    for_each_enc_resblock(h, [&](ResBlockW& rb, int) {
        for (int q = 0; q < 2; ++q) {
            rb.dline[q] = h->sn.take<float>((size_t)h->max_batch * rb.delay * rb.cb0.cin + 4);
            ok = ok && rb.dline[q];
        }
        if (h->norm) {
            rb.cb0.gn_ring = h->sn.take<float>((size_t)h->max_batch * 8 * rb.cb0.gn_P * 2);
            rb.cb1.gn_ring = h->sn.take<float>((size_t)h->max_batch * 8 * rb.cb1.gn_P * 2);
            ok = ok && rb.cb0.gn_ring && rb.cb1.gn_ring && rb.cb0.gn_P > 0;
        }
        rb.dflip = 0;
        rb.cb0.gn_head = rb.cb1.gn_head = 0;
    });
    AFTER_REQUIRE(ok, AFTER_E_NOMEM, "autoencoder: cached-encoder state allocation failed");
    AFTER_HIP_CHECK(hipMemset(h->sn.base, 0, h->sn.off));
    h->gn_window = gn_window_samples;
    h->enc_cached = true;
    h->nc_rows = 0;  // fresh zeroed contexts: the first encode fixes the streams of this set again
    return AFTER_OK;
}

// CachedGroupNorm(stream=True) on the (offline) decoder: what the reference's export_stream.ts of a
// non-causal codec runs (export_autoencoder.py:305-312 binds CachedGroupNorm.stream = True before BOTH
// twins are built) -- every GroupNorm of the decoder normalises over the previous window + the call's
// frames.  window_latent_frames: the window in latent frames (the reference: the length of the first
// decode after construction, 64 frames in the export script); 0 returns to plain GroupNorm.
extern "C" int after_ae_set_decoder_gn_window(after_ae* h, int window_latent_frames) {
    AFTER_REQUIRE(h && window_latent_frames >= 0, AFTER_E_INVALID, "bad argument");
    if (!h->norm || window_latent_frames == 0) {
        h->dec_gn_frames = 0;
        return AFTER_OK;
    }
    AFTER_REQUIRE(!h->streaming, AFTER_E_INVALID, "autoencoder: the causal streaming codec has no GroupNorm");
    if (h->sg.base) {
        AFTER_REQUIRE(window_latent_frames == h->dec_gn_alloc, AFTER_E_INVALID,
                      "autoencoder: decoder GroupNorm window is fixed at %d frames", h->dec_gn_alloc);
        h->dec_gn_frames = window_latent_frames;
        return AFTER_OK;
    }
    const after_ae_cfg& c = h->cfg;
    const int n = c.n_stages;
    size_t fl = 0;
    for_each_dec_convblock(h, [&](ConvBlockW& cb, int stage) {
        int up = 1;  // frames per latent frame at this block: the stage's own upsampling included
        for (int q = 0; q <= stage && q < n; ++q) up *= c.factors[n - 1 - q];
        cb.gn_P = window_latent_frames * up;
        fl += ((size_t)h->max_batch * 8 * cb.gn_P * 2 + 63) & ~(size_t)63;
    });
    AFTER_TRY(h->sg.init(fl * sizeof(float) + (64 << 10)));
    bool ok = true;
    for_each_dec_convblock(h, [&](ConvBlockW& cb, int) {
        cb.gn_ring = h->sg.take<float>((size_t)h->max_batch * 8 * cb.gn_P * 2);
        cb.gn_head = 0;
        ok = ok && cb.gn_ring;
    });
    AFTER_REQUIRE(ok, AFTER_E_NOMEM, "autoencoder: decoder GroupNorm state allocation failed");
    AFTER_HIP_CHECK(hipMemset(h->sg.base, 0, h->sg.off));
    h->dec_gn_alloc = window_latent_frames;
    h->dec_gn_frames = window_latent_frames;
    return AFTER_OK;
}

// latent frames by which the cached encoder's output lags the offline encoder's (0 when not enabled)
extern "C" int after_ae_encoder_delay(const after_ae* h) { return h && h->enc_cached ? h->enc_delay : 0; }

extern "C" int after_ae_enable_streaming(after_ae* h, int enable) {
    AFTER_REQUIRE(h, AFTER_E_INVALID, "null handle");
    if (!enable) {
        h->streaming = false;
        return AFTER_OK;
    }
    AFTER_REQUIRE(h->causal && !h->norm, AFTER_E_INVALID,
                  "autoencoder: streaming needs the causal, GroupNorm-free codec "
                  "(baseAE.gin:32-33,49); a non-causal codec's encoder streams through "
                  "after_ae_enable_encoder_streaming");
    AFTER_REQUIRE(!h->enc_cached, AFTER_E_INVALID, "autoencoder: the cached encoder is active");
    const after_ae_cfg& c = h->cfg;
    AFTER_REQUIRE((c.kernel_size - 1) * c.dilations[c.n_dilations - 1] <= conv_tm_halo(),
                  AFTER_E_INVALID, "autoencoder: receptive field exceeds the streaming halo");
    for (int j = 0; j < c.n_dilations; ++j)
        AFTER_REQUIRE((c.kernel_size - 1) * c.dilations[j] <= conv_tm_halo(), AFTER_E_INVALID,
                      "autoencoder: receptive field exceeds the streaming halo");
    if (!h->sa.base) {
        const int n = c.n_stages, nd = c.n_dilations;
        const int enc_slots = 1 + n * nd + n + 1, dec_slots = 1 + n + n * nd + 1;
        h->slot_elems = (size_t)h->max_batch * conv_tm_cp(h->cmax) * conv_tm_halo();
        const size_t fs = (size_t)h->max_batch * (h->pq_fk > 1 ? h->pq_fk - 1 : 1);  // (one-band identity bank: no history, one unused word)
        const size_t is = (size_t)h->max_batch * h->M * (h->pq_ik > 1 ? h->pq_ik - 1 : 1);
        h->enc_slots = enc_slots;
        h->dec_slots = dec_slots;
        AFTER_TRY(h->sa.init((2 * (size_t)(enc_slots + dec_slots) * h->slot_elems + fs + is) * sizeof(float) + 8192));
        h->enc_state = h->sa.take<float>(2 * (size_t)enc_slots * h->slot_elems);
        h->dec_state = h->sa.take<float>(2 * (size_t)dec_slots * h->slot_elems);
        h->pq_fstate = h->sa.take<float>(fs);
        h->pq_istate = h->sa.take<float>(is);
        AFTER_REQUIRE(h->pq_istate, AFTER_E_NOMEM, "autoencoder: streaming state allocation failed");
        AFTER_HIP_CHECK(hipMemset(h->sa.base, 0, h->sa.off));
    }
    h->streaming = true;
    h->enc_rows = h->enc_rows1 = h->dec_rows = 0;
    return AFTER_OK;
}

extern "C" int after_ae_pqmf_forward(after_ae* h, const float* x, float* mb, int B, int L,
                                     void* stream) {
    AFTER_REQUIRE(h && x && mb, AFTER_E_INVALID, "null argument");
    AFTER_REQUIRE(B > 0 && L > 0 && L % h->M == 0, AFTER_E_INVALID, "pqmf: L must be a multiple of %d", h->M);
    return pqmf_forward(h, (hipStream_t)stream, x, mb, B, L);
}

extern "C" int after_ae_pqmf_inverse(after_ae* h, const float* mb, float* x, int B, int Tm,
                                     void* stream) {
    AFTER_REQUIRE(h && x && mb && B > 0 && Tm > 0, AFTER_E_INVALID, "bad argument");
    return pqmf_inverse(h, (hipStream_t)stream, mb, x, B, Tm, 0, h->M);
}

static int encode_impl(after_ae* h, const float* x, float* z, int B, int L, int row0, void* stream);

extern "C" int after_ae_encode(after_ae* h, const float* x, float* z, int B, int L, void* stream) {
    return encode_impl(h, x, z, B, L, 0, stream);
}

extern "C" int after_ae_set_stream_lanes(after_ae* h, int lane_rows) {
    AFTER_REQUIRE(h && h->streaming, AFTER_E_INVALID, "autoencoder: lanes are a property of the streaming encoder (enable streaming first)");
    AFTER_REQUIRE(lane_rows >= 0 && 2 * lane_rows <= h->max_batch, AFTER_E_CAPACITY,
                  "autoencoder: two lanes of %d streams exceed max_batch=%d", lane_rows, h->max_batch);
    AFTER_REQUIRE(h->enc_flip == h->enc_flip1, AFTER_E_INVALID, "autoencoder: lanes can be set on a reset stream only");
    h->lane_rows = lane_rows;
    return AFTER_OK;
}

extern "C" int after_ae_encode_rows(after_ae* h, const float* x, float* z, int B, int L, int row0, void* stream) {
    AFTER_REQUIRE(h && h->streaming && h->lane_rows > 0, AFTER_E_INVALID, "autoencoder: encode_rows needs after_ae_set_stream_lanes");
    AFTER_REQUIRE((row0 == 0 && (B <= h->lane_rows || B == 2 * h->lane_rows)) || (row0 == h->lane_rows && B <= h->lane_rows),
                  AFTER_E_INVALID, "autoencoder: a pass covers streams of one lane (rows [0, n) or [lane, lane + n)) or both lanes in full");
    AFTER_REQUIRE(B <= h->lane_rows || h->enc_flip == h->enc_flip1, AFTER_E_INVALID,
                  "autoencoder: the two lanes are a chunk apart: encode them separately");
    return encode_impl(h, x, z, B, L, row0, stream);
}

static int encode_impl(after_ae* h, const float* x, float* z, int B, int L, int row0, void* stream) {
    AFTER_TRY(check_ae(h, B, L));
    AFTER_REQUIRE(x && z, AFTER_E_INVALID, "null tensor");
    AFTER_REQUIRE(row0 == 0 || (h->streaming && h->lane_rows > 0), AFTER_E_INVALID, "autoencoder: row offset without lanes");
    AFTER_REQUIRE(!(h->streaming && h->lane_rows > 0) || row0 > 0 || B <= h->lane_rows || B == 2 * h->lane_rows, AFTER_E_INVALID,
                  "autoencoder: with lanes set a pass covers one lane or both in full");
    hipStream_t s = (hipStream_t)stream;
    const bool lane1 = h->streaming && h->lane_rows > 0 && row0 > 0;
    const bool both = h->streaming && h->lane_rows > 0 && row0 == 0 && B == 2 * h->lane_rows;
    AFTER_REQUIRE(!both || h->enc_flip == h->enc_flip1, AFTER_E_INVALID, "autoencoder: the two lanes are a chunk apart: encode them separately");
    const after_ae_cfg& c = h->cfg;
    const int n = c.n_stages, nd = c.n_dilations;
    int T = L / h->M;
    float *b0 = h->buf[0], *b1 = h->buf[1], *b2 = h->buf[2];
    // (cached non-causal encoder: the PQMF analysis stays the offline, zero-padded one per chunk, as in the
    // reference's export_stream.ts, where only `model.encoder` is the cached twin)
    AFTER_TRY(begin_pass(h, s));  // (in front of the streaming PQMF: a desynchronised stream must not advance its filter state)
    h->pass_row0 = row0;
    AFTER_TRY(pqmf_forward(h, s, x, b0, B, L, h->streaming ? h->pq_fstate + (size_t)row0 * (h->pq_fk > 1 ? h->pq_fk - 1 : 1) : nullptr, true));
    h->pass_cached = h->pass_gnwin = h->enc_cached;
    h->pass_stream = h->streaming || h->enc_cached;
    float* sb = h->streaming ? h->enc_state : (h->enc_cached ? h->nc_state : nullptr);
    h->pass_slots = h->streaming ? h->enc_slots : h->nc_slots;
    h->pass_flip = h->streaming ? (lane1 ? &h->enc_flip1 : &h->enc_flip) : &h->nc_flip;
    if (sb) {  // the streams of a context set are the rows of its first pass (until after_ae_reset_state)
        const int rows = both ? h->lane_rows : B;
        int* const seen[2] = {h->streaming ? (lane1 ? &h->enc_rows1 : &h->enc_rows) : &h->nc_rows, both ? &h->enc_rows1 : nullptr};
        for (int* p : seen) {
            if (!p) continue;
            AFTER_REQUIRE(*p == 0 || *p == rows, AFTER_E_INVALID,
                          "autoencoder: this stream set was started with %d rows, the pass has %d: the conv contexts ping-pong per pass, "
                          "rows outside a pass would keep a two-chunk-old context -- after_ae_reset_state first", *p, rows);
            *p = rows;
        }
    }
    double* st = nullptr;
    if (h->norm && !h->pass_gnwin) {  // the first GroupNorm sees the PQMF output: its producer is not a conv
        st = next_stats(h, B);
        AFTER_TRY(launch_stats_accum_tm(b0, st, B, h->M, T, h->M < 8 ? h->M : 8, s, 0, h->max_batch * 8 * kStatWords));
    }
    AFTER_TRY(run_resblock2(h, s, h->enc_stem, b0, b1, b2, B, T, &st, sb));
    float *cur = b2, *t1 = b0, *t2 = b1;
    for (int i = 0; i < n; ++i) {
        const ResampleW& d = h->enc_down[i];
        for (int j = 0; j < nd; ++j) {
            const bool lastj = j == nd - 1;
            AFTER_TRY(run_resblock2(h, s, h->enc_res[i][j], cur, t1, t2, B, T, &st, sb,
                                    lastj ? d.alpha : nullptr, lastj ? d.invb : nullptr));
            float* o = cur;
            cur = t2;
            t2 = o;
        }
        if (i == n - 1) {  // the tail conv's Snake'd input comes out of the last strided conv
            h->next_alpha = h->enc_tail_alpha;
            h->next_invb = h->enc_tail_invb;
        }
        AFTER_TRY(run_dma(h, s, d.d, cur, nullptr, nullptr, nullptr, d.alpha, d.invb, ACT_SNAKE, d.bias,
                          nullptr, t1, B, T, T / d.f, T / d.f, true, &st, sb));
        float* o = cur;
        cur = t1;
        t1 = o;
        T /= d.f;
    }
    AFTER_TRY(run_dma(h, s, h->enc_tail.d, cur, nullptr, nullptr, nullptr, h->enc_tail_alpha,
                      h->enc_tail_invb, ACT_SNAKE, h->enc_tail.bias, nullptr, z, B, T, T, T, false,
                      nullptr, sb, 0, 1));  // z leaves in the reference's [B][Z][T] layout
    if (sb) *h->pass_flip ^= 1;  // the pass is enqueued in full: the next one reads what this one wrote
    if (sb && both) h->enc_flip1 ^= 1;
    h->pass_row0 = 0;
    h->in_pass = false;
    return AFTER_OK;
}

// x_multiband of Decoder1d.forward (SimpleNetsStream.py:643-646): y[:, :M] * sigmoid(y[:, M:])
__global__ __launch_bounds__(256) void loudness_gate_kernel(const float* __restrict__ y,
                                                            float* __restrict__ mb, int M, int Tm,
                                                            int gated, int ychan, int cs, int ts) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (idx >= (size_t)M * Tm) return;
    const int c = idx / Tm, t = idx - (size_t)c * Tm;  // output [B][M][Tm] (the reference's layout)
    const float* yb = y + (size_t)b * ychan * Tm;
    float v = yb[(size_t)c * cs + (size_t)t * ts];
    if (gated) v *= 1.0f / (1.0f + expf(-yb[(size_t)(M + c) * cs + (size_t)t * ts]));
    mb[(size_t)b * M * Tm + idx] = v;
}

static int write_multiband(after_ae* h, hipStream_t s, const float* y, float* mb, int B, int Tm, int gated,
                           int ychan) {
    if (!mb) return AFTER_OK;
    dim3 grid((unsigned)(((size_t)h->M * Tm + 255) / 256), B);
    const int cs = 1, ts = ychan;  // y is time-major
    hipLaunchKernelGGL(loudness_gate_kernel, grid, dim3(256), 0, s, y, mb, h->M, Tm, gated, ychan, cs, ts);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

// NoiseGenerator.forward after its conv stack (SimpleNetsStream.py:537-550, :462-495) fused with the decoder's tail
// (:643-650): per (clip, frame f of 8 band samples, band d)
//   amp[k] = mod_sigmoid(a[f][5 d + k] - 5) = 2 sigmoid(.)^2.3 + 1e-7                 (core.py:7-8), k < 5
//   h = irfft(amp) (8 samples), rolled by 4, times the periodic Hann window, rolled back = h[n] w[(n + 4) % 8]
//   noise[n] = sum_{i <= n} (2 u[n - i] - 1) ir[i]   (fft_convolve of 8 + 8 zero-padded samples, second half kept)
//   out[f * 8 + n][d] = y[.][d] * sigmoid(y[.][M + d]) (use_loudness; else y[.][d]) + noise[n]
// a: [B][F][5 M] time-major conv output, u: [B][F][M][8] uniform [0, 1) draws, y: [B][8 F][ychan], out: [B][8 F][M].
__global__ __launch_bounds__(256) void noise_synth_kernel(const float* __restrict__ a, const float* __restrict__ u,
                                                          const float* __restrict__ y, float* __restrict__ out, int F, int M,
                                                          int gated, int ychan) {
    const int b = blockIdx.y;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)F * M) return;
    const int f = (int)(idx / M), d = (int)(idx - (size_t)f * M);
    const float* ap = a + ((size_t)b * F + f) * (5 * M) + 5 * d;
    float amp[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float sg = 1.0f / (1.0f + expf(-(ap[k] - 5.0f)));
        amp[k] = 2.0f * powf(sg, 2.3f) + 1e-7f;
    }
    // cos(2 pi k n / 8) for k n mod 8: 1, r, 0, -r, -1, -r, 0, r with r = sqrt(1/2)
    const float r = 0.70710678118654752f;
    const float ct[8] = {1.f, r, 0.f, -r, -1.f, -r, 0.f, r};
    float ir[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        float hv = amp[0] + ((n & 1) ? -amp[4] : amp[4]);
#pragma unroll
        for (int k = 1; k < 4; ++k) hv += 2.0f * amp[k] * ct[(k * n) & 7];
        hv *= 0.125f;
        const float w = 0.5f - 0.5f * ct[(n + 4) & 7];  // hann_window(8, periodic)[(n + 4) % 8]
        ir[n] = hv * w;
    }
    const float4* up = reinterpret_cast<const float4*>(u + (((size_t)b * F + f) * M + d) * 8);
    const float4 u0 = up[0], u1 = up[1];
    const float nz[8] = {2.f * u0.x - 1.f, 2.f * u0.y - 1.f, 2.f * u0.z - 1.f, 2.f * u0.w - 1.f,
                         2.f * u1.x - 1.f, 2.f * u1.y - 1.f, 2.f * u1.z - 1.f, 2.f * u1.w - 1.f};
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i <= n; ++i) acc += nz[n - i] * ir[i];
        const size_t row = (size_t)b * 8 * F + (size_t)f * 8 + n;
        float v = y[row * ychan + d];
        if (gated) v *= 1.0f / (1.0f + expf(-y[row * ychan + M + d]));
        out[row * M + d] = v + acc;
    }
}

static int decode_impl(after_ae* h, const float* z, float* x, float* mb, int B, int T, void* stream) {
    AFTER_TRY(check_ae(h, B, (long long)T * (h ? h->ratio : 1)));
    AFTER_REQUIRE(z && x, AFTER_E_INVALID, "null tensor");
    hipStream_t s = (hipStream_t)stream;
    const after_ae_cfg& c = h->cfg;
    const int n = c.n_stages, nd = c.n_dilations;
    float *cur = h->buf[0], *t1 = h->buf[1], *t2 = h->buf[2];
    AFTER_TRY(begin_pass(h, s));
    h->pass_cached = false;
    h->pass_gnwin = h->norm && h->dec_gn_frames > 0;
    h->pass_stream = h->streaming;
    float* sb = h->streaming ? h->dec_state : nullptr;
    h->pass_slots = h->dec_slots;
    h->pass_flip = &h->dec_flip;
    if (sb) {
        AFTER_REQUIRE(h->dec_rows == 0 || h->dec_rows == B, AFTER_E_INVALID,
                      "autoencoder: the decoder's streams were started with %d rows, the pass has %d: after_ae_reset_state first "
                      "(the conv contexts ping-pong per pass)", h->dec_rows, B);
        h->dec_rows = B;
    }
    double* st = nullptr;
    AFTER_TRY(run_dma(h, s, h->dec_head.d, z, nullptr, nullptr, nullptr, nullptr, nullptr, ACT_NONE,
                      h->dec_head.bias, nullptr, cur, B, T, T, T, false, nullptr, sb, 1, 0));  // z: [B][Z][T]
    for (int i = 0; i < n; ++i) {
        const ResampleW& u = h->dec_up[i];
        AFTER_TRY(run_dma(h, s, h->streaming ? u.d_stream : u.d, cur, nullptr, nullptr, nullptr, u.alpha,
                          u.invb, ACT_SNAKE, u.bias, nullptr, t1, B, T, T * u.f, T, true, &st, sb));
        float* o = cur;
        cur = t1;
        t1 = o;
        T *= u.f;
        for (int j = 0; j < nd; ++j) {
            // the next stage starts with Snake -> ConvTranspose: its activated input is this block's second output -- unless
            // that conv is worth the bf16 pipe, whose bf16-plane input an act_pad launch writes (dma_takes_x6)
            const bool feed = j == nd - 1 && i + 1 < n &&
                              !dma_takes_x6(h, h->streaming ? h->dec_up[i + 1].d_stream : h->dec_up[i + 1].d, B, T, T, T * h->dec_up[i + 1].f);
            AFTER_TRY(run_resblock2(h, s, h->dec_res[i][j], cur, t1, t2, B, T, &st, sb,
                                    feed ? h->dec_up[i + 1].alpha : nullptr,
                                    feed ? h->dec_up[i + 1].invb : nullptr));
            o = cur;
            cur = t2;
            t2 = o;
        }
    }
    if (c.use_noise) {  // the noise branch reads the same tensor as the synthesis block (cc.AlignBranches, offline: no delay)
        AFTER_REQUIRE(!h->streaming, AFTER_E_INVALID, "autoencoder: use_noise is built for whole-clip decoding");
        AFTER_REQUIRE(h->noise_u, AFTER_E_INVALID, "autoencoder: use_noise needs the uniform draws of this call (after_ae_set_noise)");
        AFTER_REQUIRE(T % 8 == 0, AFTER_E_INVALID, "autoencoder: use_noise needs a multiple of 8 band samples");
        const float* src = cur;
        int Tn = T;
        for (int i = 0; i < 3; ++i) {
            AFTER_TRY(run_dma(h, s, h->noise.d[i], src, nullptr, nullptr, nullptr, nullptr, nullptr, i ? ACT_LRELU : ACT_NONE,
                              h->noise.bias[i], nullptr, h->nz[i], B, Tn, Tn / 2, Tn / 2, false, nullptr));
            src = h->nz[i];
            Tn /= 2;
        }
    }
    double* st1 = nullptr;
    AFTER_TRY(run_convblock2(h, s, h->synth0, cur, st, t1, nullptr, B, T, true, &st1, sb));
    AFTER_TRY(run_convblock2(h, s, h->synth1, t1, st1, t2, nullptr, B, T, false, nullptr));
    if (c.use_noise) {
        const int F = T / 8, ych = c.use_loudness ? 2 * h->M : h->M;
        hipLaunchKernelGGL(noise_synth_kernel, dim3((unsigned)cdivll((long long)F * h->M, 256), B), dim3(256), 0, s, h->nz[2],
                           h->noise_u, t2, h->nadd, F, h->M, c.use_loudness, ych);
        AFTER_HIP_CHECK(hipGetLastError());
        h->noise_u = nullptr;  // one set of draws per decode
        AFTER_TRY(write_multiband(h, s, h->nadd, mb, B, T, 0, h->M));
        AFTER_TRY(pqmf_inverse(h, s, h->nadd, x, B, T, 0, h->M, nullptr, true));
        h->in_pass = false;
        return AFTER_OK;
    }
    const int och = c.use_loudness ? 2 * h->M : h->M;
    AFTER_TRY(write_multiband(h, s, t2, mb, B, T, c.use_loudness, och));
    AFTER_TRY(pqmf_inverse(h, s, t2, x, B, T, c.use_loudness, och, h->streaming ? h->pq_istate : nullptr, true));
    if (sb) h->dec_flip ^= 1;
    h->in_pass = false;
    return AFTER_OK;
}

extern "C" int after_ae_set_noise(after_ae* h, const float* u) {
    AFTER_REQUIRE(h, AFTER_E_INVALID, "null handle");
    AFTER_REQUIRE(h->cfg.use_noise, AFTER_E_INVALID, "autoencoder: this codec has no noise branch (use_noise = 0)");
    h->noise_u = u;
    return AFTER_OK;
}

extern "C" int after_ae_decode(after_ae* h, const float* z, float* x, int B, int T, void* stream) {
    return decode_impl(h, z, x, nullptr, B, T, stream);
}

extern "C" int after_ae_decode_multi(after_ae* h, const float* z, float* x, float* multiband, int B, int T,
                                     void* stream) {
    AFTER_REQUIRE(multiband, AFTER_E_INVALID, "null multiband output");
    return decode_impl(h, z, x, multiband, B, T, stream);
}

// SimpleLatentReg (core.py:189-198) as ReluBottleneck.forward returns it next to z
// (SimpleNetsStream.py:742-760): mean(ELU(|z| - scale)) + 1.  One workgroup, fixed summation
// order (bit-deterministic); fp64 accumulation of the per-thread fp32 partial sums.
__global__ __launch_bounds__(1024) void latent_reg_kernel(const float* __restrict__ z, long long n,
                                                          float scale, float* __restrict__ out) {
    __shared__ double red[1024];
    double acc = 0.0;
    auto term = [scale](float x) {
        const float v = fabsf(x) - scale;
        return (double)(v > 0.f ? v : expm1f(v));
    };
    // four float4 in flight per thread (the loads of one trip are independent), scalar tail
    const long long n4 = ((uintptr_t)z & 15) == 0 ? n / 4 : 0;
    const float4* z4 = reinterpret_cast<const float4*>(z);
    for (long long i = threadIdx.x; i < n4; i += 4096) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i + 1024 * u < n4 ? z4[i + 1024 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + 1024 * u < n4) acc += (term(v[u].x) + term(v[u].y)) + (term(v[u].z) + term(v[u].w));
    }
    for (long long i = 4 * n4 + threadIdx.x; i < n; i += 1024) acc += term(z[i]);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)n + 1.0);
}

extern "C" int after_latent_reg(const float* z, long long n, float scale, float* out, void* stream) {
    AFTER_REQUIRE(z && out && n > 0, AFTER_E_INVALID, "latent_reg: bad argument");
    hipLaunchKernelGGL(latent_reg_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, z, n, scale, out);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

// ---------------------------------------------------------------- the other bottlenecks of SimpleNetsStream.py
__global__ __launch_bounds__(256) void bottleneck_tanh_kernel(float* __restrict__ z, long long n, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) z[i] = scale * tanhf(z[i]);
}

extern "C" int after_bottleneck_tanh(float* z, long long n, float scale, void* stream) {
    AFTER_REQUIRE(z && n > 0, AFTER_E_INVALID, "bottleneck_tanh: bad argument");
    hipLaunchKernelGGL(bottleneck_tanh_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, z, n,
                       scale);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

// one workgroup: thread t walks the (b, time) columns t, t + 1024, ...; per column the KL channel sum in fp64
__global__ __launch_bounds__(1024) void bottleneck_vae_kernel(const float* __restrict__ zraw,
                                                              const float* __restrict__ noise, float* __restrict__ z,
                                                              float* __restrict__ mean_out, float* __restrict__ kl,
                                                              int B, int Z, int T) {
    __shared__ double red[1024];
    double acc = 0.0;
    const long long cols = (long long)B * T;
    for (long long col = threadIdx.x; col < cols; col += 1024) {
        const int b = (int)(col / T), t = (int)(col - (long long)b * T);
        double ks = 0.0;
        for (int c = 0; c < Z; ++c) {
            const float m = zraw[((size_t)b * 2 * Z + c) * T + t];
            const float sc = zraw[((size_t)b * 2 * Z + Z + c) * T + t];
            // torch softplus (beta 1, threshold 20): x for x > 20, else log1p(exp(x))
            const float std = (sc > 20.f ? sc : log1pf(expf(sc))) + 1e-2f;
            const float var = std * std;
            const size_t o = ((size_t)b * Z + c) * T + t;
            z[o] = noise ? noise[o] * std + m : m;
            if (mean_out) mean_out[o] = m;
            ks += (double)(m * m + var - logf(var) - 1.0f);
        }
        acc += ks;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) kl[0] = (float)(red[0] / (double)cols);
}

extern "C" int after_bottleneck_vae(const float* zraw, const float* noise, float* z, float* mean_out, float* kl, int B,
                                    int Z, int T, void* stream) {
    AFTER_REQUIRE(zraw && z && kl && B > 0 && Z > 0 && T > 0, AFTER_E_INVALID, "bottleneck_vae: bad argument");
    hipLaunchKernelGGL(bottleneck_vae_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, zraw, noise, z, mean_out, kl,
                       B, Z, T);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}
