// 1-D convolution family of the AFTER autoencoder / conditioning encoders on gfx950.
#pragma once
#include "common.h"

namespace after {

enum ConvAct { ACT_NONE = 0, ACT_SNAKE = 1, ACT_SILU = 2, ACT_RELU = 3, ACT_TANH = 4, ACT_LRELU = 5 /* LeakyReLU(0.2) */ };
enum ConvPad { PAD_ZERO = 0, PAD_REFLECT = 1 };

constexpr int kMaxTaps = 8;
constexpr int kMaxPhases = 4;

// ---- GroupNorm statistics, accumulated across workgroups EXACTLY (round 4; before: fp64 atomicAdd of fp32 partial sums --
// exact, hence order-independent, only while the partials of one (clip, group) span < 2^19 in magnitude).
// A quantity (sum, sum of squares) of one (clip, group) is kStatBins signed 64-bit words: word k accumulates the bits of
// weight 2^(40 k - 64) .. 2^(40 k - 25) of every addend, as an integer.  An fp32 partial sum has a 24-bit significand: it is
// split over at most two adjacent words and added with INTEGER atomics -- associative and commutative, so the result does
// not depend on the order in which the workgroups arrive, by construction, whatever the magnitudes (no rounding happens
// before the words are folded, top word first, into one fp64 by the consumer).  A word holds 2^40 x 2^23 addends before it
// could overflow; addends below 2^-41 in magnitude lose bits at the bottom (flushed toward minus infinity by < 2^-64),
// above 2^95 saturate -- neither occurs for sums of O(1 .. 1e4) activations.
// Layout of a statistics buffer: [sub-slot][clip][group][2 quantities][kStatBins] 64-bit words (`double*` in the host-side
// structs only names an 8-byte word).
constexpr int kStatBins = 4;
constexpr int kStatWords = 2 * kStatBins;  // words per (clip, group)

#if defined(__HIPCC__)
// NaN / Inf / |p| >= 2^80 saturate to +-2^80 (NaN: +): finite words that add without overflow (< 2^26 in the top word per addend)
// and that no sum of real activations reaches -- the consumer (gn_scale_shift: kStatBlown) turns a quantity of that size into NaN
// statistics, so a non-finite element poisons its whole (clip, group) like torch's group_norm instead of vanishing from the sums
constexpr float kStatSat = 1.2089258196146292e24f;  // 2^80
constexpr double kStatBlown = 6.0e23;               // |sum| or sum of squares of a (clip, group) at or above this: NaN
__device__ __forceinline__ void stat_bins_add(long long* bins, float p) {
    if (p == 0.f) return;
    if (!(fabsf(p) < kStatSat)) p = p < 0.f ? -kStatSat : kStatSat;
    int e;
    const float f = frexpf(p, &e);                       // p = f x 2^e, 0.5 <= |f| < 1
    const long long m = (long long)(f * 16777216.0f);    // 24-bit signed significand: p = m x 2^(e - 24), exactly
    int pos = e - 24 + 64;                                // weight of m's least significant bit: 2^(pos - 64)
    long long v = m;
    if (pos < 0) {
        v >>= min(-pos, 63);
        pos = 0;
    }
    const int k = pos / 40, sh = pos - 40 * k;
    v <<= sh;                                             // |v| < 2^63
    const long long lo = v & ((1ll << 40) - 1), hi = v >> 40;  // v = hi x 2^40 + lo, 0 <= lo < 2^40 (floor semantics)
    if (lo) atomicAdd(reinterpret_cast<unsigned long long*>(bins + k), (unsigned long long)lo);
    if (hi && k + 1 < kStatBins) atomicAdd(reinterpret_cast<unsigned long long*>(bins + k + 1), (unsigned long long)hi);
}
// the words of one quantity, already summed over the sub-slots -> fp64 (top word first: a fixed order)
__device__ __forceinline__ double stat_bins_total(const long long (&w)[kStatBins]) {
    double t = 0.0;
#pragma unroll
    for (int k = kStatBins - 1; k >= 0; --k) t += (double)w[k] * __builtin_ldexp(1.0, 40 * k - 64);
    return t;
}
// a value computed elsewhere (the windowed GroupNorm of the streaming twin) stored in the same format, exactly
__device__ __forceinline__ void stat_bins_set(long long* bins, double v) {
#pragma unroll
    for (int k = kStatBins - 1; k >= 0; --k) {
        const double scale = __builtin_ldexp(1.0, 40 * k - 64);
        const double q = trunc(v / scale);
        bins[k] = (long long)q;
        v -= q * scale;
    }
}
#endif

// BatchNorm1d (eval) -> per-channel affine, replicated over B (done once at create)
int launch_bn_affine(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                     float* shift, int C, int B, float eps, hipStream_t s);

// weight-norm fold + re-layout (done once at create):
//   conv     v[Cout,Cin,k], g[Cout]  -> out[Cout][k][Cin_pad]
//   convT    v[Cin,Cout,2f], g[Cin]  -> out[f phases][Cout][2 taps][Cin_pad]
// g == nullptr: plain weights (no weight norm).
int pack_conv_weight(const float* v, const float* g, float* out, int Cout, int Cin, int k,
                     int Cin_pad, hipStream_t s);
// pad: the ConvTranspose1d `padding` (f/2 offline; 0 for the streaming form, where phase r
// uses x[n-1], x[n] only: cached_conv's overlap-add transposed conv)
int pack_convT_weight(const float* v, const float* g, float* out, int Cin, int Cout, int f,
                      int Cin_pad, hipStream_t s, int pad = -1);
int snake_inv_beta(const float* beta, float* out, int C, hipStream_t s);

// geometry of one conv: out[n * ostride + ooff[ph]] = sum_tap W[ph][tap] x[n * istride + toff[ph][tap]]
// (strided convs: istride; ConvTranspose1d: `phases` two-tap convs interleaved by ostride)
struct ConvDmaPlanIn {
    int Cin, Cout, taps, phases, istride, ostride;
    int toff[kMaxPhases][kMaxTaps];
    int ooff[kMaxPhases];
    int Nn_hint, B_hint;  // problem size the handle is built for
};

// ---- time-major conv path (conv_tm.hip): activations [B][T][C], conv = balanced LDS-DMA GEMM.
// Weights are the plain [phase][Cout][taps * Cp] GEMM operand, independent of the tile
// configuration (chosen per launch).
struct ConvTmPlan {
    int Cp, K, dil;
    bool ok;  // uniform tap spacing and |toff| within the halo
    size_t w_floats;
};
struct ConvTmRun {
    const float* xp;   // [B][Tp][Cp] activated, haloed input (launch_act_pad_tm)
    const float* w;    // conv_tm_repack output
    const unsigned short* xp3;  // launch_conv_x6: the same input as bf16 planes (ActPadTm::y3)
    const unsigned short* w3;   //                 conv_x6_split output
    // the two-piece fp16 form of the same launch (gemm_h3_pipe.h; conv_x6.hip's SPLIT tiles): xp3 / w3 hold h3 blocks of the input x
    // hs and of the weights x ws (conv_h3_split), hscale = hs (0: three bf16 planes), oscale = 1 / (hs x ws)
    float hscale, oscale;
    const float* bias;
    const float* res;  // [B][Tout][Cout] time-major, or nullptr
    float* y;          // [B][Tout][Cout] time-major ([B][Cout][Tout] when y_cm)
    double* stats;     // [conv_tm_stat_sub()][sub_stride] words, [B][G][2][kStatBins] in each: accumulators of y (stat_bins_add), or nullptr
    int B, Tp, Tout, Nn, G, y_cm, sub_stride;
    const float* post_scale;  // y = out_act(acc + bias) * post_scale[b * post_bstride + co] + post_shift[...]
    const float* post_shift;
    int post_bstride, out_act;
    int bias_bstride;         // 0: bias[Cout] shared; else bias[b * bias_bstride + co]
    // optional strides (0 = dense defaults): channel-sliced views of wider time-major tensors.
    // x_ld: row pitch of xp (an un-haloed tensor can serve a k = 1 conv as xp = t - halo * x_ld, Tp = T)
    int x_ld, y_ld, y_coff, res_ld, res_coff, res_cm;
    long long x_bs, y_bs, res_bs;
    // optional second output: the next conv's activated + haloed input (see conv_tm.hip)
    float* y2;
    const float* y2_scale;
    const float* y2_shift;
    const float* y2_pa;
    const float* y2_pb;
    const float* y2_add;
    long long y2_bs, y2_add_bs;
    int y2_ld, y2_coff, y2_clo, y2_chi, y2_act, y2_add_ld, y2_add_coff, y2_reflect;
};
struct ActPadTm {
    const float* x;       // [B][T][ldx] time-major ([B][C][T] when x_cm)
    float* y;             // [B][conv_tm_rows(T)][conv_tm_cp(C)]
    unsigned short* y3;   // instead of y: three bf16 planes in x6 blocks of [B x conv_x6_rows(T)][conv_tm_cp(C)] (conv_x6.hip)
    float hscale;         // != 0 with y3: TWO fp16 pieces of y x hscale in h3 blocks instead (a power of two from a bound of y: gemm_h3_pipe.h)
    const double* stats;  // producer's accumulators ([sub][sub_stride], see ConvTmRun) -> GroupNorm, or nullptr
    const float* gamma;   // with stats: GroupNorm weight; without: per-channel scale (BatchNorm eval) or nullptr
    const float* beta;
    const float* act_a;
    const float* act_b;
    float* state;         // streaming: [B][halo][Cp] left context, used and then replaced; or nullptr
    float* state_out;     // with `state`: where the NEW context goes (a second buffer: the caller ping-pongs) -- the kernel
                          // writes it beside the haloed tensor; nullptr: `state` is replaced in place by a second launch
    const float* scale_b; // per-(clip, channel) affine [B][C] (takes the place of gamma / beta), or nullptr
    const float* shift_b;
    int act, B, C, T, G, x_cm, ldx, pad_reflect, sub_stride;
    const float* x2;      // optional second time-major input [B][T][ldx2], added to x before the affine
    int ldx2;
    int stat_T;           // frames the statistics cover (0: T; streaming GroupNorm: window + T)
};
int conv_tm_halo();
int conv_tm_stat_sub();   // accumulator pairs per (clip, group): see conv_tm.hip
int conv_tm_cp(int C);    // channels rounded up to the 32-deep K slab
int conv_tm_rows(int T);  // rows of the haloed buffer
void conv_tm_plan(const ConvDmaPlanIn& in, ConvTmPlan* p);
int conv_tm_repack(const float* packed, float* out, const ConvDmaPlanIn& in, const ConvTmPlan& p, hipStream_t s);
int launch_conv_tm(const ConvTmRun& r, const ConvDmaPlanIn& in, const ConvTmPlan& p, hipStream_t s);
int launch_act_pad_tm(const ActPadTm& p, hipStream_t s);
// ---- GroupNorm -> activation -> Conv1d(k = 1) (+ residual) as ONE launch (conv_tm.hip: conv1_act_kernel): no activated tensor
struct Conv1ActRun {
    const float* x;          // [B][T][ldx] time-major raw input
    const double* stats_in;  // the producer's accumulators of x (GroupNorm), or nullptr
    const float *gamma, *beta, *act_a, *act_b;
    const float* w;          // conv_tm_repack output of the k = 1 conv: [C][ldw]
    const float* bias;
    const float* res;        // [B][T][res_ld] or nullptr (may alias y)
    float* y;                // [B][T][y_ld]
    double* stats_out;       // accumulators of y (ConvTmRun::stats) or nullptr
    int B, T, C, G, act, ldx, ldw, res_ld, y_ld, sub_stride;
};
bool conv1_act_eligible(int B, int T, int C, int G, bool stats);
int launch_conv1_act(const Conv1ActRun& r, hipStream_t s);
long long conv1_act_launches();
// ---- the same convs through the bf16 matrix pipe (conv_x6.hip): stride-1 convs of <= 3 taps, operands as bf16 planes
constexpr int kConvTmHalo = 32;  // == conv_tm_halo()
int conv_x6_rows(int T);                          // plane rows per clip: conv_tm_rows(T) rounded up to 16
size_t conv_x6_plane_elems(int B, int T, int C);  // unsigned shorts of the plane tensor act_pad_tm writes
size_t conv_x6_weight_elems(const ConvDmaPlanIn& in, const ConvTmPlan& p);
bool conv_x6_eligible(const ConvDmaPlanIn& in, const ConvTmPlan& p);
int conv_x6_split(const float* w_tm, unsigned short* w3, const ConvDmaPlanIn& in, const ConvTmPlan& p, hipStream_t s);
int conv_x6_mode();  // AFTER_CONV_X6: 0 never, 1 by size (default), 2 wherever eligible
// (h3: the launch can take the two-piece fp16 form -- its planes are no larger than the fp32 tensor, so every FILLED launch wins,
//  whatever the width; the three-plane form only where it measured faster: the decoder's widths)
bool conv_x6_wins(const ConvTmRun& r, const ConvDmaPlanIn& in, const ConvTmPlan& p, bool h3 = false);
// the weights of a conv_x6-eligible layer as two fp16 pieces x `scale` (h3 blocks per phase: conv_h3_weight_elems unsigned shorts)
size_t conv_h3_weight_elems(const ConvDmaPlanIn& in, const ConvTmPlan& p);
int conv_h3_split(const float* w_tm, unsigned short* w2, const ConvDmaPlanIn& in, const ConvTmPlan& p, float scale, hipStream_t s);
int launch_conv_x6(const ConvTmRun& r, const ConvDmaPlanIn& in, const ConvTmPlan& p, hipStream_t s);
// sub_stride > 0: stats is a [conv_tm_stat_sub()][sub_stride] accumulator (ConvTmRun) and the blocks spread over its sub-slots
int launch_stats_accum_tm(const float* x, double* stats, int B, int C, int T, int G, hipStream_t s, int ld = 0, int sub_stride = 0);
// small layout helpers on time-major tensors
int launch_copy_cols_tm(const float* src, int ld_src, float* dst, int ld_dst, int C, size_t rows, hipStream_t s);
int launch_cm_to_tm(const float* x, float* y, int B, int C, int T, int ld, hipStream_t s);
int launch_upsample_rows_tm(const float* x, float* y, int B, int C, int T, int r, hipStream_t s);

inline int pad16(int c) { return (c + 15) & ~15; }

// cached_conv.get_padding left pad (stride ignored): p = (k-1) d + 1
inline int conv_left_pad(int k, int dil, bool causal) {
    if (k == 1) return 0;
    const int p = (k - 1) * dil + 1;
    return causal ? p / 2 + (p - 1) / 2 : (p - 1) / 2;
}

}  // namespace after
