// Stride-1 convolutions of the codec through the bf16 matrix pipe: the conv GEMM of conv_tm.hip
//
//   y[b, n * ostride + ooff, co] = bias[co] + res[...] + sum_{tap, ci} xp[b, HALO + n + toff0 + tap * dil, ci] * w[co, tap * Cp + ci]
//
// with every fp32 product formed by six exact bf16 MFMAs on three-way bf16 splits of both operands (gemm_x6.hip: fp32
// accumulation, error vs fp64 <= the fp32 MFMA chain's; 2.67x its issue rate).  The dilated k = 3 convs, the k = 1 convs
// and the two-tap phases of the transposed convs (reference SimpleNetsStream.py:150-194 ConvBlock1d, :51-70 Upsample1d)
// are MFMA-bound at the sizes of the decoder's upper stages: those launches run here, the others stay on conv_tm.
//
// Operands: the activated, haloed input arrives as bf16 planes in x6 blocks (common.h) of a [B x rows16][Cp] matrix,
// written by act_pad_tm (ActPadTm::y3) instead of its fp32 tensor; the weights are split once at create
// (conv_x6_split).  A tile's A rows for tap t are the plane rows m0 + shift_t .. m0 + shift_t + BM - 1: the stage holds
// the MB + 1 ALIGNED 16-row blocks that contain them (whole 1-KB LDS-DMA pieces whatever the shift; the x6 chunk
// permutation depends on row % 16 only, and 16 consecutive rows from any start are bank-conflict free) and the
// fragment reads start shift_t % 16 rows into it (gemm_x6_pipe.h, C::CONV).  Everything else -- ring, counted vmcnt,
// one barrier per slab, side work behind individual MFMAs -- is the pipeline of gemm_x6.
//
// Epilogue = conv_tm's for the features these launches use: bias, residual, fp32 time-major output (strided rows for the
// transposed-conv phases), statistics of the next GroupNorm (per-wave LDS slots added in a fixed order, binned integer
// atomics across workgroups: conv.h).
#include <cstdint>
#include <cstdlib>

#include "conv.h"
#include "gemm_x6_pipe.h"
#include "gemm_h3_pipe.h"

namespace after {
namespace {

constexpr int kStatSubX6 = 8;  // == conv_tm_stat_sub() (checked by the launcher)

struct ConvX6Args {
    const unsigned short* A3;  // planes of [B x rows16][Cp]
    const unsigned short* W3;  // [phases] x planes of [Cout][K]
    const float* bias;
    const float* res;
    float* y;
    double* stats;
    int Cp, Cout, K, Tout, Nn, phases, ostride, G, sub_stride;
    int blocks_per_clip, total_blocks, cpb, magic;
    int y_ld, y_coff, res_ld, res_coff;
    long long y_bs, res_bs;
    size_t w3_phase;            // elements between the phases of W3
    float oscale;               // SPLIT tiles (two fp16 pieces per operand): 1 / (input scale x weight scale), an exact power of two
    int blk0[kMaxPhases];       // first 16-row block of tap 0, relative to the tile's own block
    int sh[kMaxPhases][3];      // row shift inside the first block, per tap
    int dblk[kMaxPhases][3];    // blocks between tap t's and tap 0's first block
    int ooff[kMaxPhases];
};

template <class C>
__global__ __launch_bounds__(64 * C::NW, C::WPS) void conv_x6_kernel(ConvX6Args g, int tiles_m, int tiles_n, int xcd_pm,
                                                                     int ny) {
    static_assert(C::CONV == 1 && C::OUT3 == 0 && C::ACC2 == 0, "conv tiles: fp32 output");
    constexpr int BM = C::BM, BN = C::BN, MT = C::MT, NT = C::NT, KS = C::KS, RS = C::RS, NW = C::NW, NS = C::NS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    // ---- workgroup -> (clip / phase, tile): conv_tm's XCD-aware map
    const int nwg = tiles_m * tiles_n;
    int tm, tn, yi;
    if (xcd_pm > 0) {
        const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
        const int per = nwg >> 3;
        yi = li / per;
        const int l2 = li - yi * per;
        const int pn = 8 / xcd_pm;
        const int cm = tiles_m / xcd_pm, cn = tiles_n / pn;
        const int xi = xcd % xcd_pm, xj = xcd / xcd_pm;
        tm = xi * cm + l2 % cm;
        tn = xj * cn + l2 / cm;
    } else {
        const int total = nwg * ny;
        int bid = blockIdx.x;
        const int xcd = bid & 7, q = total >> 3, r = total & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        yi = bid / nwg;
        const int t = bid - yi * nwg;
        tn = t / tiles_m;
        tm = t - tn * tiles_m;
    }
    const int b = yi / g.phases, ph = yi - b * g.phases;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wid % KS, rp = (wid / KS) % RS, cp = wid / (KS * RS);  // k-part, row part, column part
    const int M = g.Nn, N = g.Cout;
    const int nk = (g.K >> 5) / KS;  // slabs per k-part

    X6State<C> c;
    c.wid = wid;
    c.lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem_raw;
    c.full = !C::RAGGED || (wid + NW * (C::LPS - 1) < C::P);
    c.voff = (unsigned)lane * 16u;
    c.magic = g.magic;
    // tap t's A pieces: slab S = t * cpb + cb sits at K block cb of the row blocks dblk[t] further on
    {
        int d[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) d[t] = (g.dblk[ph][t] - t) * g.cpb * (C::NPL * 1024);
        c.doff[0] = __builtin_amdgcn_readfirstlane(d[0]);
        c.doff[1] = __builtin_amdgcn_readfirstlane(d[1] - d[0]);
        c.doff[2] = __builtin_amdgcn_readfirstlane(d[2] - d[1]);
    }
    c.kslab_w = kh * nk;
    {
        const int kbw = g.K >> 5;
        const unsigned short* w3 = g.W3 + (size_t)ph * g.w3_phase;
        const int ablk = b * g.blocks_per_clip + (m0 >> 4) + g.blk0[ph];
#pragma unroll
        for (int i = 0; i < C::LPS; ++i) {
            int p = wid + NW * i;
            if (p >= C::P) p = C::P - 1;  // never issued (c.full == false)
            const int kp = p / C::PPK, q = p - kp * C::PPK;
            const unsigned short* base;
            int am;
            if (q < C::GA) {
                const int plane = q / C::AB, grp = q - plane * C::AB;
                const int rb = min(ablk + grp, g.total_blocks - 1);  // past the tensor: a clamped block, rows unused
                base = g.A3 + (((size_t)rb * g.cpb) * C::NPL + plane) * 512 + (size_t)kp * nk * (C::NPL * 512);
                am = -1;
            } else {
                const int qq = q - C::GA;
                const int plane = qq / C::NBK, grp = qq - plane * C::NBK;
                const int rb = min((n0 >> 4) + grp, (N - 1) >> 4);
                base = w3 + (((size_t)rb * kbw + (size_t)kp * nk) * C::NPL + plane) * 512;
                am = 0;
            }
            c.kslab[i] = __builtin_amdgcn_readfirstlane(kp * nk);
            const unsigned long long v = (unsigned long long)(uintptr_t)base;
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
            c.sb[i] = ((unsigned long long)hi << 32) | lo;
            c.amask[i] = __builtin_amdgcn_readfirstlane(am);
        }
    }
    // ---- fragment addresses: lane l -> row l & 15 of a 16-row block, 16-byte chunk l >> 4 of its 64-byte row; A per
    // tap: the window starts sh rows into the stage's first block
    {
        const int frow = lane & 15, kq = lane >> 4;
        unsigned ar[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int row = rp * (BM / RS) + frow + g.sh[ph][t];
            ar[t] = c.lds0 + (unsigned)(kh * C::PART + row * 64) + (unsigned)((kq ^ swz4((row >> 2) & 3)) * 16);
        }
        c.a_rd = c.a_rd3[0] = ar[0];
        c.a_rd3[1] = ar[1] - ar[0];
        c.a_rd3[2] = ar[2] - ar[1];
        c.w_rd = c.lds0 + (unsigned)(kh * C::PART + C::GA * 1024 + (cp * (BN / C::CP) + frow) * 64) +
                 (unsigned)((kq ^ swz4((frow >> 2) & 3)) * 16);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) c.acc[0][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- the residual tile is requested BEFORE the ring fill: it lands with slab 0 (vmcnt retires in order) and waits
    // in registers, instead of an exposed round trip in the epilogue of the CU's only workgroup (measured, 384 -> 384
    // k = 1 at eight clips: 181 us with the loads in the epilogue against 110 us without a residual)
    const int crow = lane & 15, ccol0 = 4 * (lane >> 4);
    const int row0 = m0 + rp * (BM / RS), col0 = n0 + cp * (BN / C::CP);
    const float* rb = g.res ? g.res + (size_t)b * g.res_bs + g.res_coff : nullptr;
    f32x4 rv[MT][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            rv[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int gm = row0 + i * 16 + crow, gn = col0 + j * 16 + ccol0;
            const int trow = gm * g.ostride + g.ooff[ph];
            if (rb && (i * NT + j) % KS == kh && gm < M && gn < N && trow < g.Tout) rv[i][j] = *reinterpret_cast<const f32x4*>(rb + (size_t)trow * g.res_ld + gn);
        }

    // ---- prologue: fill the ring, wait for slab 0, read its fragments; then the slab steps
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < nk) x6_issue_slab<C>(c, s, s);
    x6_wait<C>(c, (nk < NS ? nk : NS) - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const int t0 = x6_tap<C>(c, c.kslab_w);
        const unsigned a0 = c.a_rd3[0] + (c.a_rd3[1] & (unsigned)((0 - t0) >> 31)) + (c.a_rd3[2] & (unsigned)((1 - t0) >> 31));
        x6_sides<C, 0, C::LPS, C::NWORK, true>(c, false, true, 0, 0, a0, c.w_rd);
    }
    int kt = 0;
    for (; kt + 1 + NS < nk; kt += 2) {
        x6_step<C, 0, true>(c, kt, nk);
        x6_step<C, 1, true>(c, kt + 1, nk);
    }
    for (; kt < nk; kt += 2) {
        x6_step<C, 0, false>(c, kt, nk);
        if (kt + 1 < nk) x6_step<C, 1, false>(c, kt + 1, nk);
    }

    // ---- epilogue.  accumulator layout (W fragment as srcA): lane l holds C[row = l & 15][col = 4 (l >> 4) + r]
    // Statistics of the next GroupNorm: every lane's partial sums go through a 16-row butterfly, then into the
    // workgroup's binned INTEGER accumulators in LDS (ds_add_u64: exact, order-independent), whose non-zero words are
    // added to the global accumulators (conv.h: stat_bins_add).  (conv_tm's first form -- per-wave float slots added by
    // one lane per group in a serial loop of dependent LDS reads -- cost 8-11 us per tile.)
    // k-parts are summed through LDS in k-part order (bit-deterministic), as in gemm_x6
    constexpr size_t kRedBytes = KS > 1 ? (size_t)NW * MT * NT * 1024 : 0;
    static_assert(kRedBytes + 24 * kStatWords * 8 <= (size_t)NS * C::STAGE, "epilogue scratch exceeds the ring");
    float* red = reinterpret_cast<float*>(smem_raw);
    long long* lbins = reinterpret_cast<long long*>(smem_raw + kRedBytes);  // [groups of this tile][kStatWords], over the idle ring
    const int Cg = g.stats ? N / g.G : 4;
    const int g0 = n0 / Cg;
    const int ng = g.stats ? (min(n0 + BN, N) - 1) / Cg - g0 + 1 : 0;
    if (g.stats || KS > 1) __syncthreads();  // every wave is past its last ring read
    if (g.stats)
        for (int i = tid; i < ng * kStatWords; i += 64 * NW) lbins[i] = 0;
    if constexpr (KS > 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                *reinterpret_cast<f32x4*>(red + ((wid * MT * NT + i * NT + j) * 64 + lane) * 4) = c.acc[0][i][j];
        __syncthreads();
    }
    const int w0 = (cp * RS + rp) * KS;  // first wave of the k-parts of this wave's blocks
    float* yb = g.y + (size_t)b * g.y_bs + g.y_coff;
    float ssum[NT], qsum[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int gn = col0 + j * 16 + ccol0;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (g.bias && gn < N) bv = *reinterpret_cast<const f32x4*>(g.bias + gn);  // (Cout % 4 == 0: launcher)
        ssum[j] = qsum[j] = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if ((i * NT + j) % KS != kh) continue;  // the blocks' k-parts are finished by one of their waves each
            const int gm = row0 + i * 16 + crow;
            if (gm >= M || gn >= N) continue;
            const int trow = gm * g.ostride + g.ooff[ph];
            if (trow >= g.Tout) continue;
            f32x4 a4 = c.acc[0][i][j];
            if constexpr (KS > 1) {
                a4 = *reinterpret_cast<const f32x4*>(red + ((w0 * MT * NT + i * NT + j) * 64 + lane) * 4);
#pragma unroll
                for (int q = 1; q < KS; ++q)
                    a4 += *reinterpret_cast<const f32x4*>(red + (((w0 + q) * MT * NT + i * NT + j) * 64 + lane) * 4);
            }
            if constexpr (C::SPLIT) a4 = a4 * g.oscale;
            const f32x4 o = (a4 + bv) + rv[i][j];
            *reinterpret_cast<f32x4*>(yb + (size_t)trow * g.y_ld + gn) = o;
            ssum[j] += (o[0] + o[1]) + (o[2] + o[3]);
            qsum[j] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
        }
    }
    if (g.stats) {
        // the 16 lanes l & 15 of a quad share the channel quad (one group: Cg % 4 == 0): butterfly over the rows
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int o2 = 1; o2 < 16; o2 <<= 1) {
                ssum[j] += __shfl_xor(ssum[j], o2, 64);
                qsum[j] += __shfl_xor(qsum[j], o2, 64);
            }
        __syncthreads();  // the bins are zero
        if (crow == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int gn = col0 + j * 16 + ccol0;
                if (gn < N) {
                    long long* bp = lbins + (gn / Cg - g0) * kStatWords;
                    stat_bins_add(bp, ssum[j]);
                    stat_bins_add(bp + kStatBins, qsum[j]);
                }
            }
        }
        __syncthreads();
        if (tid < ng * kStatWords) {
            const long long v = lbins[tid];
            if (v) {
                const int grp = g0 + tid / kStatWords, k = tid % kStatWords;
                long long* sp = reinterpret_cast<long long*>(g.stats) + (size_t)(blockIdx.x % kStatSubX6) * g.sub_stride +
                                ((size_t)b * g.G + grp) * kStatWords + k;
                atomicAdd(reinterpret_cast<unsigned long long*>(sp), (unsigned long long)v);
            }
        }
    }
}

// tile configurations: rows x 96 columns, eight waves (4 row parts x 2 column parts), two-stage ring
using CfgX6_128 = X6Cfg<8, 6, 1, 4, 2, 2, 0, 1, 0, 0, 1>;   // 128 x 96: 256 workgroups for 8192 x 384 (one clip's 384-channel stage)
using CfgX6_192 = X6Cfg<12, 6, 1, 4, 2, 2, 0, 1, 0, 0, 1>;  // 192 x 96: the many-row tile of gemm_x6 (9 accumulators per wave)
using CfgX6_128n = X6Cfg<8, 4, 1, 4, 2, 2, 0, 1, 0, 0, 1>;  // 128 x 64: 64- and 192-channel outputs
using CfgX6_96 = X6Cfg<6, 6, 1, 2, 2, 2, 0, 2, 0, 0, 1>;    // 96 x 96, four waves, two workgroups per CU
using CfgX6_128s3 = X6Cfg<8, 6, 1, 4, 2, 3, 0, 1, 0, 0, 1>; // 128 x 96 with a three-stage ring
// (32 x 96 with two k-parts -- 256 workgroups for the 768-channel stage at T = 1024 -- measured 37 us against 39 for conv_tm's
//  split-K tile: not worth its operand conversion; dropped)
using CfgX6_128w = X6Cfg<8, 8, 1, 4, 2, 2, 0, 1, 0, 0, 1>;   // 128 x 128: the encoder's widths (128 / 256 / 512 channels)
using CfgX6_128w3 = X6Cfg<8, 8, 1, 4, 2, 3, 0, 1, 0, 0, 1>;  // the same with a three-stage ring (153 KB)
using CfgX6_64k2 = X6Cfg<4, 6, 2, 1, 6, 2, 0, 1, 0, 0, 1>;  // 64 x 96, two k-parts, twelve waves: 256 workgroups for 4096 x 384
// the same tiles on two-piece fp16 operands (SPLIT: stages of two thirds the size -- the two-stage rings become three-stage ones)
using CfgH3_128 = X6Cfg<8, 6, 1, 4, 2, 3, 0, 1, 0, 0, 1, 0, 0, 1>;
using CfgH3_192 = X6Cfg<12, 6, 1, 4, 2, 3, 0, 1, 0, 0, 1, 0, 0, 1>;
using CfgH3_128n = X6Cfg<8, 4, 1, 4, 2, 3, 0, 1, 0, 0, 1, 0, 0, 1>;
using CfgH3_96 = X6Cfg<6, 6, 1, 2, 2, 3, 0, 2, 0, 0, 1, 0, 0, 1>;
using CfgH3_128w = X6Cfg<8, 8, 1, 4, 2, 3, 0, 1, 0, 0, 1, 0, 0, 1>;
using CfgH3_64k2 = X6Cfg<4, 6, 2, 1, 6, 2, 0, 1, 0, 0, 1, 0, 0, 1>;

template <class C>
int launch_x6_cfg(const ConvX6Args& a, int B, hipStream_t s) {
    const int tiles_m = cdiv(a.Nn, C::BM), tiles_n = cdiv(a.Cout, C::BN);
    const int nwg = tiles_m * tiles_n, ny = B * a.phases;
    size_t lds = (size_t)C::NS * C::STAGE;
    int pm = 0;
    if ((nwg & 7) == 0) {
        double best = 0;
        for (int cdv = 1; cdv <= 8; cdv *= 2) {
            if (tiles_m % cdv || tiles_n % (8 / cdv)) continue;
            const double cost = (double)a.Nn * a.Cp / cdv + (double)a.Cout * a.K / (8 / cdv);
            if (pm == 0 || cost < best) {
                pm = cdv;
                best = cost;
            }
        }
    }
    static LdsAttr attr;
    AFTER_TRY(ensure_lds_attr(attr, reinterpret_cast<const void*>(conv_x6_kernel<C>), lds));
    hipLaunchKernelGGL(conv_x6_kernel<C>, dim3(nwg * ny), dim3(64 * C::NW), lds, s, a, tiles_m, tiles_n, pm, ny);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

int g_x6_force = -1;  // AFTER_CONV_X6: 0 = never, 1 = by size (default), 2 = wherever eligible
long long g_x6_launches = 0;  // after_conv_x6_launches(): the tests check that the default path really takes this kernel
long long g_h3_launches = 0;  // ... of them on two-piece fp16 operands
int g_x6_tile = -1;   // AFTER_CONV_X6_TILE / after_convtm_set_x6_tile: 0 = by shape, 1 = 128 x 96, 2 = 192 x 96, 3 = 128 x 64

}  // namespace

int conv_x6_rows(int T) { return (conv_tm_rows(T) + 15) & ~15; }
size_t conv_x6_plane_elems(int B, int T, int C) { return (size_t)B * conv_x6_rows(T) * 3 * conv_tm_cp(C); }
size_t conv_x6_weight_elems(const ConvDmaPlanIn& in, const ConvTmPlan& p) { return (size_t)in.phases * x6_elems(in.Cout, p.K); }

bool conv_x6_eligible(const ConvDmaPlanIn& in, const ConvTmPlan& p) {
    return p.ok && in.istride == 1 && in.taps <= 3 && (in.Cout & 3) == 0 && in.phases <= kMaxPhases;
}

size_t conv_h3_weight_elems(const ConvDmaPlanIn& in, const ConvTmPlan& p) { return (size_t)in.phases * h3_elems(in.Cout, p.K); }

int conv_h3_split(const float* w_tm, unsigned short* w2, const ConvDmaPlanIn& in, const ConvTmPlan& p, float scale, hipStream_t s) {
    for (int ph = 0; ph < in.phases; ++ph)
        AFTER_TRY(gemm_h3_split(w_tm + (size_t)ph * in.Cout * p.K, p.K, w2 + (size_t)ph * h3_elems(in.Cout, p.K), in.Cout, p.K, scale, s));
    return AFTER_OK;
}

int conv_x6_split(const float* w_tm, unsigned short* w3, const ConvDmaPlanIn& in, const ConvTmPlan& p, hipStream_t s) {
    for (int ph = 0; ph < in.phases; ++ph)
        AFTER_TRY(gemm_x6_split(w_tm + (size_t)ph * in.Cout * p.K, p.K, w3 + (size_t)ph * x6_elems(in.Cout, p.K), in.Cout, p.K, s));
    return AFTER_OK;
}

int conv_x6_mode() {
    if (g_x6_force < 0) {
        const char* e = getenv("AFTER_CONV_X6");
        g_x6_force = e ? atoi(e) : 1;
    }
    return g_x6_force;
}

// does this launch run on the bf16 pipe?  (the caller then hands act_pad_tm a plane buffer instead of the fp32 one)
bool conv_x6_wins(const ConvTmRun& r, const ConvDmaPlanIn& in, const ConvTmPlan& p, bool h3) {
    const int mode = conv_x6_mode();
    if (mode == 0 || !conv_x6_eligible(in, p)) return false;
    if (r.y2 || r.y_cm || r.res_cm || r.post_scale || r.out_act || r.bias_bstride || r.x_ld || r.x_bs) return false;
    if (r.stats && (in.Cout % r.G || ((in.Cout / r.G) & 3) || r.G > 16)) return false;
    if ((r.y_ld & 3) || (r.y_coff & 3) || (r.res_ld & 3) || (r.res_coff & 3)) return false;
    if (mode >= 2) return true;
    if (h3 && in.Cout % 96) {
        // two fp16 pieces: the operand tensors are 4 bytes per element like the fp32 path's, and a product block is three MFMAs of 16
        // cycles against eight of 32 -- the encoder's widths (64 / 128 / 256 / 512: the 128 x 128 and 128 x 64 tiles) win wherever the
        // launch fills the chip (profiles/r6_ab_conv_h3_wide.txt: encode at 8 / 32 clips 4.46 -> 3.80 / 16.2 -> 13.2 ms; half-filled
        // launches -- one clip -- lose: 1.12 -> 1.21)
        if (in.Cout % 64) return false;
        const int tn = (in.Cout & 127) == 0 ? 128 : 64;
        return (double)cdiv(r.Nn, 128) * cdiv(in.Cout, tn) * r.B * in.phases >= 200 && p.K >= 192;
    }
    // MFMA-bound launches only: enough 128 x 96 tiles to fill the chip, enough K for the ring to run
    const double wgs = (double)cdiv(r.Nn, 128) * cdiv(in.Cout, 96) * r.B * in.phases;
    // half-filled launches of the 128-row tile: the 64 x 96 tile with two k-parts wins the k = 3 convs (384 channels at
    // T = 4096: 28 against 39 us), not the k = 1 ones
    if (wgs < 200 && in.taps >= 2 && (p.K & 63) == 0 && in.Cout % 96 == 0 &&
        (double)cdiv(r.Nn, 64) * cdiv(in.Cout, 96) * r.B * in.phases >= 200)
        return true;
    // widths that are not multiples of 96 (the encoder's 64 / 128 / 256 / 512) stay on conv_tm.  Isolated, the 128 x 128
    // tile wins their k = 3 convs from two clips on (eight clips: 256 channels at T = 8192 199 -> 134 us, 512 at 1024
    // 97 -> 60, on every lease); inside the encoder the gain did not reproduce from lease to lease (encode at eight clips
    // 4.61 -> 4.41 ms on one box, 4.61 -> 4.71 on two others, the fp32 path within 0.3 % on all of them: the 1.5x
    // larger plane tensors push the pass's working set past the 256-MB Infinity Cache).  AFTER_CONV_X6=2 runs them here.
    if (in.Cout % 96) return false;
    // (K = 192 -- the 192-channel k = 1 convs, six slabs -- only where the launch is several rounds of tiles: eight clips at
    //  T = 16384 act_pad 52 + conv 137 us on the fp32 path; decode 6.60 -> 6.42 ms; at one clip the fp32 conv's 16 us stand)
    return wgs >= 200 && (p.K >= 256 || (p.K >= 192 && wgs >= 1024)) && in.Cout >= 64;
}

int launch_conv_x6(const ConvTmRun& r, const ConvDmaPlanIn& in, const ConvTmPlan& p, hipStream_t s) {
    AFTER_REQUIRE(conv_x6_eligible(in, p) && r.xp3 && r.w3, AFTER_E_INVALID, "conv_x6: not an x6 launch");
    AFTER_REQUIRE(conv_tm_stat_sub() == kStatSubX6, AFTER_E_INVALID, "conv_x6: statistics sub-slot count");
    ConvX6Args a;
    memset(&a, 0, sizeof(a));
    a.A3 = r.xp3;
    a.W3 = r.w3;
    a.bias = r.bias;
    a.res = r.res;
    a.y = r.y;
    a.stats = r.stats;
    a.Cp = p.Cp;
    a.Cout = in.Cout;
    a.K = p.K;
    a.Tout = r.Tout;
    a.Nn = r.Nn;
    a.phases = in.phases;
    a.ostride = in.ostride;
    a.G = r.G;
    a.sub_stride = r.sub_stride;
    a.cpb = p.Cp / 32;
    a.magic = (65536 + a.cpb - 1) / a.cpb;
    const int rows16 = (r.Tp + 15) & ~15;
    a.blocks_per_clip = rows16 / 16;
    a.total_blocks = r.B * a.blocks_per_clip;
    a.y_ld = r.y_ld > 0 ? r.y_ld : in.Cout;
    a.y_coff = r.y_coff;
    a.y_bs = r.y_bs > 0 ? r.y_bs : (long long)r.Tout * in.Cout;
    a.res_ld = r.res_ld > 0 ? r.res_ld : in.Cout;
    a.res_coff = r.res_coff;
    a.res_bs = r.res_bs > 0 ? r.res_bs : (long long)r.Tout * in.Cout;
    const bool h3 = r.hscale != 0.f;
    a.w3_phase = h3 ? h3_elems(in.Cout, p.K) : x6_elems(in.Cout, p.K);
    a.oscale = r.oscale;
    const int halo = conv_tm_halo();
    for (int ph = 0; ph < in.phases; ++ph) {
        const int s0 = halo + in.toff[ph][0];
        a.blk0[ph] = s0 >> 4;
        for (int t = 0; t < 3; ++t) {
            const int st = halo + in.toff[ph][t < in.taps ? t : in.taps - 1];
            a.sh[ph][t] = st & 15;
            a.dblk[ph][t] = (st >> 4) - (s0 >> 4);
        }
        a.ooff[ph] = in.ooff[ph];
    }
    AFTER_REQUIRE(((uintptr_t)r.y & 15) == 0 && (!r.res || ((uintptr_t)r.res & 15) == 0) && (!r.bias || ((uintptr_t)r.bias & 15) == 0),
                  AFTER_E_INVALID, "conv_x6: operand alignment");
    if (g_x6_tile < 0) {
        const char* e = getenv("AFTER_CONV_X6_TILE");
        g_x6_tile = e ? atoi(e) : 0;
    }
    int t = g_x6_tile;
    if (t == 0) {
        // measured per layer (scripts/bench_conv.py --x6, profiles/r4_bench_conv_x6_b{1,8}.jsonl): the three-stage 128 x 96
        // ring wins the k = 3 / two-tap launches at every size; the k = 1 convs (12-24 slabs, residual + statistics
        // epilogue) run best as two 96 x 96 workgroups per CU once those fill the chip twice over; widths that are
        // multiples of 64 but not of 96 take the 128 x 64 tile; 192 x 96 never won
        const long long ny = (long long)r.B * in.phases;
        if (in.Cout % 96 && (in.Cout & 127) == 0) t = p.K >= 1536 ? 8 : 7;
        else if (in.Cout % 96 && in.Cout % 64 == 0) t = 3;
        else if ((double)cdiv(r.Nn, 128) * cdiv(in.Cout, 96) * ny < 200) t = 6;  // (conv_x6_wins: taps >= 2, K % 64 == 0)
        else if (in.taps == 1 && (double)cdiv(r.Nn, 96) * cdiv(in.Cout, 96) * ny >= 1024) t = 4;
        else {
            t = 5;
            // widths that both tiles divide (384, 768): rounds of 256 workgroups x tile area, the 128 x 128 tile moving 7 %
            // fewer operand bytes per MFMA (eight clips: 384 channels at T = 8192 300 -> 273 us, 768 at 1024 stays: 127 vs 144)
            if ((in.Cout & 127) == 0) {
                const double w5 = (double)cdiv(r.Nn, 128) * (in.Cout / 96) * ny, w7 = (double)cdiv(r.Nn, 128) * (in.Cout / 128) * ny;
                const double c5 = (double)cdivll((long long)w5, 256) * 96, c7 = (double)cdivll((long long)w7, 256) * 128 * 0.93;
                if (c7 < c5) t = 7;
            }
        }
    }
    if (t == 6 && (p.K & 63)) t = 5;  // two k-parts need an even slab count
    ++g_x6_launches;
    if (h3) {
        ++g_h3_launches;
        switch (t) {
            case 1: return launch_x6_cfg<CfgH3_128>(a, r.B, s);
            case 2: return launch_x6_cfg<CfgH3_192>(a, r.B, s);
            case 3: return launch_x6_cfg<CfgH3_128n>(a, r.B, s);
            case 4: return launch_x6_cfg<CfgH3_96>(a, r.B, s);
            case 5: return launch_x6_cfg<CfgH3_128>(a, r.B, s);
            case 6: return launch_x6_cfg<CfgH3_64k2>(a, r.B, s);
            case 7: case 8: return launch_x6_cfg<CfgH3_128w>(a, r.B, s);
            default: break;
        }
        set_error("conv_x6: no two-piece tile configuration %d", t);
        return AFTER_E_INVALID;
    }
    switch (t) {
        case 1: return launch_x6_cfg<CfgX6_128>(a, r.B, s);
        case 2: return launch_x6_cfg<CfgX6_192>(a, r.B, s);
        case 3: return launch_x6_cfg<CfgX6_128n>(a, r.B, s);
        case 4: return launch_x6_cfg<CfgX6_96>(a, r.B, s);
        case 5: return launch_x6_cfg<CfgX6_128s3>(a, r.B, s);
        case 6: return launch_x6_cfg<CfgX6_64k2>(a, r.B, s);
        case 7: return launch_x6_cfg<CfgX6_128w>(a, r.B, s);
        case 8: return launch_x6_cfg<CfgX6_128w3>(a, r.B, s);
        default: break;
    }
    set_error("conv_x6: no tile configuration %d", t);
    return AFTER_E_INVALID;
}

}  // namespace after

extern "C" void after_convtm_set_x6_tile(int id) { after::g_x6_tile = id; }
extern "C" long long after_conv_x6_launches(void) { return after::g_x6_launches; }
extern "C" long long after_conv_h3_launches(void) { return after::g_h3_launches; }
