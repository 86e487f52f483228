// 1-D convolution family of the AFTER autoencoder / conditioning encoders on gfx950.
#pragma once
#include "common.h"

namespace after {

enum ConvAct { ACT_NONE = 0, ACT_SNAKE = 1, ACT_SILU = 2, ACT_RELU = 3, ACT_TANH = 4 };
enum ConvPad { PAD_ZERO = 0, PAD_REFLECT = 1 };

constexpr int kMaxTaps = 8;
constexpr int kMaxPhases = 4;

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
    const float* bias;
    const float* res;  // [B][Tout][Cout] time-major, or nullptr
    float* y;          // [B][Tout][Cout] time-major ([B][Cout][Tout] when y_cm)
    double* stats;     // [conv_tm_stat_sub()][sub_stride] doubles, [B][G][2] in each: accumulators of y, or nullptr
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
    const double* stats;  // producer's accumulators ([sub][sub_stride], see ConvTmRun) -> GroupNorm, or nullptr
    const float* gamma;   // with stats: GroupNorm weight; without: per-channel scale (BatchNorm eval) or nullptr
    const float* beta;
    const float* act_a;
    const float* act_b;
    float* state;         // streaming: [B][halo][Cp] left context, used and then replaced; or nullptr
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
int launch_stats_accum_tm(const float* x, double* stats, int B, int C, int T, int G, hipStream_t s, int ld = 0);
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
