// Winograd F(2x2, 3x3) convolution for gfx950 (fp32 MFMA 32x32x2) - round 5.
//
// For the 3x3 / stride 1 / pad 1 C -> C convolutions of HRNet's high-resolution branches (pose_hrnet.py:22-57: 32 -> 32
// @64x48 and 64 -> 64 @32x24 at 256x192 are 47 % of the forward FLOPs) the direct kernel (conv_direct.hip) sits at 0.41-0.51
// of the fp32 matrix peak with its main loop 77 % MFMA-busy: the remaining lever is the multiply count itself.
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A          (Lavin & Gray; 16 multiplies per 2x2 outputs and channel pair, not 36)
// turns the conv into 16 independent [tiles x Ci] . [Ci x Co] products, one per position xi of the 4x4 transformed patch:
// 2.25x fewer MFMAs, pure fp32 (the transforms are additions and two multiplications by 1/2).
//
//   * Weights: U = G g G^T is computed ONCE per forward pass for every eligible conv of a network by ONE launch
//     (wino_weights; advmix_wino_weights) into a side buffer laid out in MFMA B-fragment order,
//     U[n tile][xi][k group q][lane][4] - a wave's B operand of (xi, q) is one contiguous 1 KB load, never staged in LDS
//     (the four waves of a workgroup multiply DIFFERENT xi, so there is nothing to share).  Two images per conv: the
//     forward one (n = Cout, k = Cin) and the input-gradient one (n = Cin, k = Cout, taps rotated by 180 degrees): a
//     stride-1 input gradient is the same convolution with the rotated, transposed filter.
//   * A workgroup = 32 tiles (128 output pixels) x 32 output channels, four waves.  Wave i owns row i of the transformed
//     patch, xi = (i, 0..3): row i of B^T d needs two of the patch's four pixel rows (8 of its 16 pixels), loaded
//     straight into the MFMA A-fragment layout (lane = tile, 16 bytes = 4 channels) with bounds-checked buffer loads -
//     an out-of-image pixel returns 0.0f.  Per 8-channel group: 8 + 4 sixteen-byte loads, 32 additions, 16 MFMAs.
//   * Inverse transform: the column half (A applied from the right) stays inside the wave's accumulators; the row half
//     (A^T from the left) adds across the four waves through LDS - one exchange per workgroup, written as a [tile][channel]
//     image, so that what comes out is the NATURAL layout: thread = (tile, output position, 4 channels).  There the SAME
//     fused epilogues as conv_direct run with 16-byte operand loads and stores and no transposer: BatchNorm column sums (fp64
//     slots), eval-mode BatchNorm + residual + activation, or - input-gradient role - addend, activation slope from the bit
//     mask / from c, BatchNorm-backward sums.
#include "common.h"
#include <stdio.h>

namespace wino {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;      // >= any buffer size accepted -> loads return 0, stores are dropped
constexpr int STORE_AUX = 16;              // sc1 (write-through), as conv_direct's epilogue

struct WinoP {
    const float* x;
    const float* u;           // transformed weights, fragment order (see wino_weights)
    float* y;
    int N, H, W, Ci, Co;      // 3x3, stride 1, pad 1: input and output are both H x W
    int Ht, Wt;               // 2x2 output tiles per column / row
    int nbw, nblk;            // blocks of BW x BH tiles per row / per image
    int xbytes, ybytes, ubytes;
    // role 0 (forward): column sums of the raw output and / or eval-mode BatchNorm, residual, activation
    const float *bn_gamma, *bn_beta, *bn_rm, *bn_rv, *res;
    float bn_eps;
    int act;
    double* stats;            // [2][stats_nbg][Co] fp64 slots (slot-major), zero on entry
    int stats_nbg;
    // role 1 (input gradient): ``res`` is the addend; with bnb_c the epilogue is the BatchNorm-backward one (ConvD in
    // conv_direct.hip: same fields, same arithmetic)
    const unsigned char* bnb_mask;
    const float *bnb_c, *bnb_mean, *bnb_invstd, *bnb_gamma, *bnb_beta;
    int bnb_act;
    int xcd_remap;
    // INBN (round 6): the input is the RAW output c of the producing conv; its train-mode BatchNorm + ReLU is applied while the
    // patch is staged.  in_slots: the producer's column sums [2][in_ns][Ci] (fp64, as its epilogue left them), in_rows = N H W.
    // Every workgroup derives mean / invstd itself; workgroup (0, 0) also publishes them (saved for the backward pass) and
    // updates the running statistics - what advmix_norm_apply_slots does in its own launch.
    const double* in_slots;
    int in_ns;
    double in_rows;
    float in_eps, in_momentum;
    const float *in_gamma, *in_beta;
    float *in_mean, *in_invstd, *in_rmean, *in_rvar;
    long long* in_nbt;
};

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ int acc_row(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }   // v_mfma_f32_32x32x2 D layout

// INBN: the producer's slots reduced by the whole workgroup - thread (part, channel) adds every NP-th slot of its channel (all
// loads issued before the first add: one L2 round trip, under the patch loads already in flight), the parts meet in LDS
// (``red``: the patch region, not yet written) - then thread ch < C derives mean / invstd exactly as norm_apply_slots_kernel
// does (norm.hip) and leaves (mean, invstd, gamma, beta) in ``bnp``; workgroup (0, 0) publishes mean / invstd and updates the
// running statistics.  Contains one __syncthreads(); the caller adds the one that makes ``bnp`` visible.
template <int C, int NT>
__device__ __forceinline__ void in_bn_params(const WinoP& p, float* bnp, double* red, int tid) {
    constexpr int NP = NT / C >= 4 ? 4 : (NT / C >= 2 ? 2 : 1);
    constexpr int PER = 16 / NP;                            // slots per part at most (in_ns <= 16: the entry point checks)
    const int ns = p.in_ns;
    if (tid < NP * C) {
        const int part = tid / C, ch = tid - part * C;
        double v0[PER], v1[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int k = part + NP * i;
            v0[i] = k < ns ? p.in_slots[(int64_t)k * C + ch] : 0.0;
            v1[i] = k < ns ? p.in_slots[((int64_t)ns + k) * C + ch] : 0.0;
        }
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < PER; ++i) { s0 += v0[i]; s1 += v1[i]; }
        red[(2 * part) * C + ch] = s0;
        red[(2 * part + 1) * C + ch] = s1;
    }
    __syncthreads();
    if (tid < C) {
        double s0 = red[tid], s1 = red[C + tid];
#pragma unroll
        for (int q = 1; q < NP; ++q) { s0 += red[(2 * q) * C + tid]; s1 += red[(2 * q + 1) * C + tid]; }
        const double m = s0 / p.in_rows;
        double var = s1 / p.in_rows - m * m;
        if (var < 0) var = 0;
        const float mu = (float)m, is = (float)(1.0 / sqrt(var + (double)p.in_eps));
        bnp[tid] = mu;
        bnp[C + tid] = is;
        bnp[2 * C + tid] = p.in_gamma[tid];
        bnp[3 * C + tid] = p.in_beta[tid];
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            p.in_mean[tid] = mu;
            p.in_invstd[tid] = is;
            if (p.in_rmean) {
                const double unb = p.in_rows > 1 ? var * p.in_rows / (p.in_rows - 1) : var;
                p.in_rmean[tid] = (float)((1.0 - p.in_momentum) * (double)p.in_rmean[tid] + (double)p.in_momentum * m);
                p.in_rvar[tid] = (float)((1.0 - p.in_momentum) * (double)p.in_rvar[tid] + (double)p.in_momentum * unb);
            }
            if (tid == 0 && p.in_nbt) *p.in_nbt += 1;
        }
    }
}

// Geometry of a workgroup's block of BW x BH = 32 tiles and of its input patch in LDS.  Pixel (pr, pc) of the
// (2 BH + 2) x (2 BW + 2) patch lives at position pr * PWL + (pc & 1) * HALF + (pc >> 1), PP = C + 4 floats per position:
// a lane's 16-byte read of column c of ITS tile touches position ... + bx' + (c >> 1) - even and odd pixel columns are
// stored apart, so the 32 tiles of a wave read CONSECUTIVE positions (not every second one), and with HALF chosen so that
// a tile row (two pixel rows = 4 HALF positions) advances the 16-byte slot by BW, the 16 lanes the hardware serves per
// cycle of a ds_read_b128 fall on 16 different slots: no bank conflicts.
template <int LBW, int C, int KS = 1>
struct Geo {
    static constexpr int BW = 1 << LBW, BH = 32 >> LBW;
    static constexpr int PH = 2 * BH + 2, PW = 2 * BW + 2;
    static constexpr int HALF = LBW == 3 ? 10 : (LBW == 2 ? 5 : (LBW == 4 ? 20 : PW / 2));
    static constexpr int PWL = 2 * HALF, PP = C + 4;
    static constexpr int PATCH = PH * PWL * PP;             // floats
    static constexpr int XCH = KS * 4 * 2 * 32 * 40;        // floats of the inverse transform's exchange image ([wave][b][32 tiles][40])
    static constexpr int LDS = PATCH > XCH ? PATCH : XCH;
};

// NQ = Ci / 8 (k groups of 8 channels: two k-lanes x 4 channels per 16-byte load).
// KS: K is split over KS sets of four waves (workgroup = 4 KS waves): the 128-channel branch (16 x 12 maps: 64 blocks x 4
// column tiles = one workgroup per CU at B = 32) gets two waves per SIMD that way, each multiplying half of the channels;
// the halves meet in the exchange of the inverse transform, which adds across waves anyway.
// waves per SIMD the register allocator may assume: what the LDS footprint admits anyway (three workgroups of <= 53 KB per CU);
// round 6: FOUR for the 32-channel forward variants - their 40 KB admit it for the eval / plain epilogues, and the tighter register
// budget (79-124 VGPRs beside the accumulators) schedules all three better: 13.0 -> 12.4 us forward + sums, 12.8 -> 11.5 eval,
// launch to launch (the BatchNorm-backward variant loses at that budget and keeps three; in the step: no difference, r06j)
template <int NQ, int LBW, int KS, int NC = 1, int VAR = 0>
constexpr int wino_waves() {
    if (NQ == 4 && NC == 1 && VAR != 2) return 4;                       // (64 / 128 channels: no difference measured)
    if (KS > 1 || Geo<LBW, NQ * 8, KS>::LDS * 4 + 4096 > 80 * 1024) return 1;
    if (NC > 1) return 2;
    return Geo<LBW, NQ * 8, KS>::LDS * 4 + 1024 <= 53 * 1024 ? 3 : 2;
}

// VAR: the epilogue compiled in - 0 forward + BatchNorm column sums, 1 forward + eval-mode BatchNorm (+ residual) + activation,
// 2 input gradient (+ addend) + BatchNorm-backward epilogue, 3 plain (forward or input gradient, + residual / addend)
// NC (round 6): column tiles of 32 output channels per workgroup.  With NC = 1 the grid's y dimension walks the column tiles,
// so a 64-channel conv stages every patch and transforms every input group TWICE (0.73-0.80 in the step
// where the 32-channel instance is at 0.85-0.88); NC = 2 keeps both tiles' accumulators (2 x 64 registers) in the wave: one
// staging, one B^T d B per 8-channel group, 32 MFMAs behind it instead of 16; the epilogue runs once per column tile.
// INBN (round 6): 1 = the input tensor is the raw output c of a conv whose train-mode BatchNorm + ReLU has NOT
// been applied: act(fma((c - mean) * invstd, gamma, beta)) - norm_apply_slots' very expression, so that the backward pass's
// "sign from c" epilogue and the weight gradient's staging see the same activation bit for bit - is applied to every element
// on its way into LDS (~1.3 loads per input element, four VALU operations each), and the separate norm_apply_slots launch with
// its y tensor disappears: +0.3 ... +1.3 us on the conv against 6.2 ... 10.2 us for the launch (profiles/r06d_microbench_wino_inbn.log,
// tools/microbench_wino_inbn.py).
template <int NQ, int VAR, int LBW, int KS = 1, int NC = 1, int INBN = 0>
__global__ __launch_bounds__(256 * KS, (wino_waves<NQ, LBW, KS, NC, VAR>())) void conv_wino(const WinoP p) {
    using G = Geo<LBW, NQ * 8, KS>;
    constexpr int NT = 256 * KS, NQW = NQ / KS;            // threads; k groups per wave
    static_assert(NQ % KS == 0, "K splits evenly over the wave sets");
    constexpr int C = NQ * 8, BW = G::BW, BH = G::BH, PW = G::PW, PH = G::PH, HALF = G::HALF, PWL = G::PWL, PP = G::PP;
    // one region, four lives (separated by workgroup barriers): the input patch, the exchange of the inverse transform's
    // row half ([wave][b][r / 4][lane][4]), then the wave-private transposers of the epilogue
    __shared__ __attribute__((aligned(16))) float L[G::LDS];
    __shared__ float sred[2 * 4 * KS * 32];
    __shared__ __attribute__((aligned(16))) float bnp[INBN ? 4 * NQ * 8 : 4];      // INBN: mean, invstd, gamma, beta per input channel
    float* const X = L;
    static_assert(NC == 1 || KS == 1, "column tiles per workgroup and the K split are alternatives");

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int wid = wv & 3, kh = wv >> 2;                  // row of the transformed patch this wave multiplies; its share of K
    const int l31 = lane & 31, lh = lane >> 5;
    int bm = blockIdx.x;
    if (p.xcd_remap && (gridDim.x & 7) == 0 && gridDim.x >= 16)            // each XCD (and its L2) works through a contiguous
        bm = (bm & 7) * ((int)gridDim.x >> 3) + (bm >> 3);                  // range of blocks: halos are L2 hits
    const int n00 = blockIdx.y * (32 * NC);
    const int img = bm / p.nblk;                                            // (uniform: scalar arithmetic)
    const int rblk = bm - img * p.nblk;
    const int bby = rblk / p.nbw, bbx = rblk - bby * p.nbw;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.u, 0, p.ubytes, 0x00020000);
    const unsigned bo = (unsigned)((((blockIdx.y * NC * 16 + wid * 4) * NQ) * 64 + lane) * 16);
    f32x4 bn[NC][4];
    auto issue_b = [&](int q) {
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) bn[ct][j] = bload(ur, bo + (unsigned)(((ct * 16 + j) * NQ + kh * NQW + q) * 1024));
    };
    issue_b(0);

    // ---- the block's input patch -> LDS: every input byte is requested ONCE per workgroup (the first version loaded each
    // lane's 8 patch pixels straight into registers: 4x the bytes - neighbouring tiles and the four waves overlap - at a
    // quarter of a cache line per request; the loads, not the MFMAs, set its time: profiles/r05a_knockout_wino_v1.log) ----
    {
        constexpr int NS = PH * PW * (C / 4), NIT = (NS + NT - 1) / NT;
        const int hb = 2 * BH * bby - 1, wb = 2 * BW * bbx - 1;
        f32x4 stg[NIT];
        if constexpr (NT % (C / 4) == 0) {
            // a thread keeps its channel quad and walks the patch NT / (C / 4) pixels at a time: (row, column) carried along
            // instead of divided out of the slot index in every iteration (the SIMD's VALU time adds to its MFMA time)
            constexpr int Q = C / 4, DPX = NT / Q, DPR = DPX / PW, DPC = DPX % PW;
            const int cs = tid % Q, px0 = tid / Q;
            int pr = px0 / PW, pc = px0 - pr * PW;
            const int gbase = ((img * p.H + hb) * p.W + wb) * C + cs * 4;        // element offset of patch pixel (0, 0) (may be negative)
            unsigned lo[NIT];
            unsigned okm = 0u;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const bool ok = pr < PH && (unsigned)(hb + pr) < (unsigned)p.H && (unsigned)(wb + pc) < (unsigned)p.W;
                stg[it] = bload(xr, ok ? (unsigned)((gbase + (pr * p.W + pc) * C) * 4) : OOB);
                okm |= ok ? (1u << it) : 0u;
                lo[it] = pr < PH ? (unsigned)((pr * PWL + (pc & 1) * HALF + (pc >> 1)) * PP + cs * 4) : 0xffffffffu;
                pc += DPC; pr += DPR;
                if (pc >= PW) { pc -= PW; ++pr; }
            }
            if constexpr (INBN) {
                static_assert(NIT <= 32, "one bit per staged piece");
                in_bn_params<C, NT>(p, bnp, reinterpret_cast<double*>(L), tid);      // (the patch loads are in flight meanwhile)
                __syncthreads();
                const f32x4 mu = *reinterpret_cast<const f32x4*>(&bnp[cs * 4]), is = *reinterpret_cast<const f32x4*>(&bnp[C + cs * 4]);
                const f32x4 g = *reinterpret_cast<const f32x4*>(&bnp[2 * C + cs * 4]), b = *reinterpret_cast<const f32x4*>(&bnp[3 * C + cs * 4]);
#pragma unroll
                for (int it = 0; it < NIT; ++it)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = fmaxf(__builtin_fmaf((stg[it][e] - mu[e]) * is[e], g[e], b[e]), 0.f);
                        stg[it][e] = ((okm >> it) & 1u) ? t : 0.f;      // (the zero ring pads the ACTIVATION, not c)
                    }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (lo[it] != 0xffffffffu) *reinterpret_cast<f32x4*>(&L[lo[it]]) = stg[it];
        } else {
            unsigned okm = 0u;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int s = tid + NT * it;
                const int px = s / (C / 4), cs = s % (C / 4);
                const int pr = px / PW, pc = px % PW;
                const int h = hb + pr, w = wb + pc;
                const bool ok = s < NS && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                stg[it] = bload(xr, ok ? (unsigned)((((img * p.H + h) * p.W + w) * C + cs * 4) * 4) : OOB);
                okm |= ok ? (1u << it) : 0u;
            }
            if constexpr (INBN) {
                static_assert(NIT <= 32, "one bit per staged piece");
                in_bn_params<C, NT>(p, bnp, reinterpret_cast<double*>(L), tid);
                __syncthreads();
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int s = tid + NT * it;
                const int px = s / (C / 4), cs = s % (C / 4);
                const int pr = px / PW, pc = px % PW;
                if constexpr (INBN) {
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(&bnp[cs * 4]), is = *reinterpret_cast<const f32x4*>(&bnp[C + cs * 4]);
                    const f32x4 g = *reinterpret_cast<const f32x4*>(&bnp[2 * C + cs * 4]), b = *reinterpret_cast<const f32x4*>(&bnp[3 * C + cs * 4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = fmaxf(__builtin_fmaf((stg[it][e] - mu[e]) * is[e], g[e], b[e]), 0.f);
                        stg[it][e] = ((okm >> it) & 1u) ? t : 0.f;
                    }
                }
                if (s < NS) *reinterpret_cast<f32x4*>(&L[(pr * PWL + (pc & 1) * HALF + (pc >> 1)) * PP + cs * 4]) = stg[it];
            }
        }
    }

    // ---- this lane's tile ------------------------------------------------------------------------------------------
    const int bxl = l31 & (BW - 1), byl = l31 >> LBW;

    // wave i multiplies row i of B^T d B:   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]  ->  d[ra] + sg * d[rb]
    const int ra = wid == 0 ? 0 : (wid == 2 ? 2 : 1);
    const int rb = wid == 3 ? 3 : (wid == 2 ? 1 : 2);
    const float sg = wid == 1 ? 1.f : -1.f;
    const float* const la = L + ((2 * byl + ra) * PWL + bxl) * PP + 4 * lh + kh * (NQW * 8);
    const float* const lb = L + ((2 * byl + rb) * PWL + bxl) * PP + 4 * lh + kh * (NQW * 8);

    f32x16 acc[NC][4];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][j][r] = 0.f;

    __syncthreads();                                        // the patch is complete
#pragma unroll
    for (int q = 0; q < NQW; ++q) {
        f32x4 rc[4], v[4], bc[NC][4];
        __builtin_amdgcn_sched_barrier(0);                  // (nothing of group q + 1 is hoisted above group q's MFMAs)
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) bc[ct][j] = bn[ct][j];
        if (q + 1 < NQW) issue_b(q + 1);                     // the next group's filters fly under this group's 16 NC MFMAs
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int o = ((c & 1) * HALF + (c >> 1)) * PP + 8 * q;
            const f32x4 da = *reinterpret_cast<const f32x4*>(la + o);
            const f32x4 db = *reinterpret_cast<const f32x4*>(lb + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) rc[c][e] = __builtin_fmaf(sg, db[e], da[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {                       // (B^T d) B:  columns [1 0 0 0; 0 1 -1 1; -1 1 1 0; 0 0 0 -1]
            v[0][e] = rc[0][e] - rc[2][e];
            v[1][e] = rc[1][e] + rc[2][e];
            v[2][e] = rc[2][e] - rc[1][e];
            v[3][e] = rc[1][e] - rc[3][e];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[ct][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][e], bc[ct][j][e], acc[ct][j], 0, 0, 0);
    }

    // ---- inverse transform, column half (inside the wave):  M A,  A^T = [1 1 1 0; 0 1 -1 -1] ---------------------------
    // From here on the workgroup works in the NATURAL layout: item = (tile, output position (oa, ob) of its 2 x 2, 4 consecutive
    // channels); thread tid owns items tid + NT k, all with the same channel quad cq = tid % 8.  (Until round 5's last kernels
    // the four finishing waves kept the MFMA accumulator layout - one column, 16 scattered rows per lane - and moved every
    // read operand and the result through wave-private LDS transposers: 80 LDS operations and five dependent round trips per
    // lane for the BatchNorm-backward role; here the exchange image itself is the transposer: 32 four-byte writes, 3 KS
    // sixteen-byte reads per item, operands and stores straight from / to memory in 16-byte pieces.)
    constexpr int XP = 40;                                  // pitch of a tile row in the exchange image [wave][b][32 tiles][XP]
    constexpr int NI = 1024 / NT;                           // items per thread (32 tiles x 4 positions x 8 quads)
    static_assert(KS * 4 * 2 * 32 * XP <= G::LDS, "the exchange image fits the region");
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.y), 0, p.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bnb_c ? p.bnb_c : p.y), 0, p.ybytes, 0x00020000);
    const bool mask_on = VAR == 2 && p.bnb_mask != nullptr && p.bnb_act != ADVMIX_ACT_NONE;
    const bool recompute = VAR == 2 && !mask_on && p.bnb_act != ADVMIX_ACT_NONE;
    const bool has_res = VAR != 0 && p.res != nullptr;
    const int cq = tid & 7;
    // item k of this thread: tile tl0 + (NT / 32) k of the block, output position (oa, ob) = bits 1, 0 of tid / 8 - the SAME for all
    // its items; the tile walks down the block: BW divides NT / 32, so tx stays and ty advances by (NT / 32) / BW per item
    const int tl0 = tid >> 5, oa = (tid >> 4) & 1, ob = (tid >> 3) & 1;
    constexpr int TSTEP = (NT / 32) / BW;                   // tile rows per item
    static_assert((NT / 32) % BW == 0, "an item step is whole tile rows");
    const int tx = bbx * BW + (tl0 & (BW - 1)), ty0 = bby * BH + (tl0 >> LBW);
    const bool xok_t = tx < p.Wt;
    const int pix0 = (img * p.H + 2 * ty0 + oa) * p.W + 2 * tx + ob;
    const int pstep = 2 * TSTEP * p.W;                      // pixels per item
    // the epilogue's read operands (addend / residual, c, mask bytes) of column tile ct + 1 are requested before the items of
    // column tile ct are worked through: with NC = 1 they arrive under the exchange, with NC = 2 the second set arrives under
    // the first tile's epilogue (requested all at once they would not fit the registers beside 128 accumulators)
    unsigned yo[NC][NI];
    f32x4 oa4[NC][NI], oc4[NC][NI];
    unsigned mb[NC][NI];
    auto fetch = [&](const int ct) {
        const int col = n00 + 32 * ct + 4 * cq;
        const bool xok = xok_t && col < p.Co;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const bool ok = xok && ty0 + TSTEP * k < p.Ht;
            const int pix = pix0 + pstep * k;
            yo[ct][k] = ok ? (unsigned)((pix * p.Co + col) * 4) : OOB;
            oa4[ct][k] = f32x4{0.f, 0.f, 0.f, 0.f};
            oc4[ct][k] = oa4[ct][k];
            mb[ct][k] = 0u;
            if (VAR != 0 && has_res) oa4[ct][k] = bload(rr, yo[ct][k]);  // requested now, they arrive under the exchange
            if constexpr (VAR == 2) {
                oc4[ct][k] = bload(cr, yo[ct][k]);
                if (mask_on && ok) mb[ct][k] = p.bnb_mask[yo[ct][k] >> 4];      // byte (pixel * Co + column) / 4; bit e: channel col + e
            }
        }
    };
    fetch(0);
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        const int n0 = n00 + 32 * ct;
        const int col = n0 + 4 * cq;
        const bool cvalid = col < p.Co;                         // (Co = 48: the second column tile is half empty - its filters are zero)

        __syncthreads();                                        // every wave is done with the patch (with the previous column tile's image): the region becomes the exchange image
        {
            float* const xw = &X[(wv * 2 * 32 + 4 * lh) * XP + l31];      // D[tile row acc_row(r, lh)][column l31]
    #pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float t0 = (acc[ct][0][r] + acc[ct][1][r]) + acc[ct][2][r];
                const float t1 = (acc[ct][1][r] - acc[ct][2][r]) - acc[ct][3][r];
                xw[((r & 3) + 8 * (r >> 2)) * XP] = t0;
                xw[(32 + (r & 3) + 8 * (r >> 2)) * XP] = t1;
            }
        }
        __syncthreads();
        if (ct + 1 < NC) fetch(ct + 1);
        // ---- row half (across the waves) and the fused epilogue (conv_direct.hip's arithmetic), per item.  The epilogue VARIANT
        // is a template parameter: a SIMD issues VALU and MFMA instructions one after the other (csrc/conv_pw.hip's knock-outs), so
        // every instruction of an epilogue the launch does not use was paid for in the run-time-generic first version ------------
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        const float bb_slope = act_neg_slope(p.bnb_act);
        const int colc = cvalid ? col : 0;
        auto ld4 = [&](const float* q) { return *reinterpret_cast<const f32x4*>(q + colc); };   // (colc % 4 == 0: one 16-byte load)
        f32x4 bn_is = {1.f, 1.f, 1.f, 1.f}, bn_g = bn_is, bn_b = {0.f, 0.f, 0.f, 0.f}, bn_m = bn_b;
        if constexpr (VAR == 1) {
            const f32x4 rv = ld4(p.bn_rv);
    #pragma unroll
            for (int e = 0; e < 4; ++e) bn_is[e] = 1.0f / sqrtf(rv[e] + p.bn_eps);
            bn_g = ld4(p.bn_gamma); bn_b = ld4(p.bn_beta); bn_m = ld4(p.bn_rm);
        }
        f32x4 bb_mu = {0.f, 0.f, 0.f, 0.f}, bb_is = bb_mu, bb_g = bb_mu, bb_b = bb_mu;
        if constexpr (VAR == 2) {
            bb_mu = ld4(p.bnb_mean); bb_is = ld4(p.bnb_invstd);
            if (recompute) { bb_g = ld4(p.bnb_gamma); bb_b = ld4(p.bnb_beta); }
        }
        const float sg2 = oa ? -1.f : 1.f;                      // Y[0][b] = T0 + T1 + T2,  Y[1][b] = T1 - T2 - T3
        const float* const xr0 = &X[((oa * 2 + ob) * 32 + tl0) * XP + 4 * cq];
    #pragma unroll
        for (int k = 0; k < NI; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
    #pragma unroll
            for (int kk = 0; kk < KS; ++kk) {                   // (the K shares of the wave sets add up here)
                const float* const xp = xr0 + (4 * kk * 2 * 32 + (NT / 32) * k) * XP;
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(xp);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(xp + 2 * 32 * XP);
                const f32x4 x2 = *reinterpret_cast<const f32x4*>(xp + 4 * 32 * XP);
    #pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += __builtin_fmaf(sg2, x1[e] + x2[e], x0[e]);
            }
            const bool valid = yo[ct][k] != OOB;
            if constexpr (VAR == 0) {
                if (valid) {
    #pragma unroll
                    for (int e = 0; e < 4; ++e) { s1[e] += v[e]; s2[e] = __builtin_fmaf(v[e], v[e], s2[e]); }
                }
            } else {
    #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float o = v[e];
                    if constexpr (VAR == 1) {
                        o = (o - bn_m[e]) * bn_is[e] * bn_g[e] + bn_b[e];
                        o += oa4[ct][k][e];
                        o = act_fwd(o, p.act);
                    } else {
                        o += oa4[ct][k][e];
                        if constexpr (VAR == 2) {
                            const float xh = (oc4[ct][k][e] - bb_mu[e]) * bb_is[e];
                            if (mask_on) o = ((mb[ct][k] >> e) & 1u) ? o : o * bb_slope;
                            else if (recompute) o = __builtin_fmaf(xh, bb_g[e], bb_b[e]) > 0.f ? o : o * bb_slope;
                            if (valid) { s1[e] += o; s2[e] = __builtin_fmaf(o, xh, s2[e]); }
                        }
                    }
                    v[e] = o;
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, yo[ct][k], 0, STORE_AUX);
        }
        if constexpr (VAR == 0 || VAR == 2) {
            // a wave = 8 (tile, position) items x 8 channel quads (lane = 8 i + cq): the items add up by shuffles, the waves in LDS
    #pragma unroll
            for (int e = 0; e < 4; ++e) {
    #pragma unroll
                for (int d = 8; d < 64; d <<= 1) {
                    s1[e] += __shfl_xor(s1[e], d, 64);
                    s2[e] += __shfl_xor(s2[e], d, 64);
                }
            }
            if (lane < 8) {
    #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sred[wv * 32 + 4 * lane + e] = s1[e];
                    sred[(4 * KS + wv) * 32 + 4 * lane + e] = s2[e];
                }
            }
            __syncthreads();
            if (tid < 32) {
                double d1 = 0.0, d2 = 0.0;
    #pragma unroll
                for (int k = 0; k < 4 * KS; ++k) {
                    d1 += (double)sred[k * 32 + tid];
                    d2 += (double)sred[(4 * KS + k) * 32 + tid];
                }
                const int sl = (int)blockIdx.x % p.stats_nbg;  // slot-major [2][slots][Co]: consecutive doubles per workgroup
                if (n0 + tid < p.Co) {
                    atomicAdd(p.stats + (int64_t)sl * p.Co + n0 + tid, d1);
                    atomicAdd(p.stats + ((int64_t)p.stats_nbg + sl) * p.Co + n0 + tid, d2);
                }
            }
        }
    }
}

// ---- weight transform -------------------------------------------------------------------------------------------------
// One block of 256 threads = one (n tile of 32, k group of 8) of one image: thread t = 4 * (32 * lh + n) + e holds
// (n, k = 8 q + 4 lh + e) and writes its 16 values U[xi] at ((ntile * 16 + xi) * NQ + q) * 256 + t - the order in which
// the conv's lanes read them.  role 0: g[r][s] = w[n][r][s][k] (forward; n = Cout, k = Cin); role 1: g[r][s] = w[k][2 - r][2 - s][n]
// (input gradient; n = Cin, k = Cout).  G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1].
struct WinoEnt {
    const float* w;
    float* u;
    int Cn, Ck, role, blk0;
};

__global__ __launch_bounds__(256) void wino_weights(const WinoEnt* __restrict__ ents, const int* __restrict__ blk_ent) {
    const WinoEnt e = ents[blk_ent[blockIdx.x]];
    const int lb = (int)blockIdx.x - e.blk0;
    const int NQ = e.Ck >> 3;
    const int nt = lb / NQ, q = lb - nt * NQ;
    const int t = threadIdx.x;
    const int cn = nt * 32 + ((t >> 2) & 31), ck = q * 8 + (t >> 7) * 4 + (t & 3);
    const bool live = cn < e.Cn;                            // (Cn = 48: columns 48 .. 63 of the second tile are zero filters)
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s)
            g[r][s] = !live ? 0.f : (e.role == 0 ? e.w[((int64_t)(cn * 3 + r) * 3 + s) * e.Ck + ck]
                                                 : e.w[((int64_t)(ck * 3 + (2 - r)) * 3 + (2 - s)) * e.Cn + cn]);
    float a[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        a[0][s] = g[0][s];
        a[1][s] = 0.5f * ((g[0][s] + g[2][s]) + g[1][s]);
        a[2][s] = 0.5f * ((g[0][s] + g[2][s]) - g[1][s]);
        a[3][s] = g[2][s];
    }
    float* const out = e.u + ((int64_t)nt * 16 * NQ + q) * 256 + t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float u0 = a[i][0];
        const float u1 = 0.5f * ((a[i][0] + a[i][2]) + a[i][1]);
        const float u2 = 0.5f * ((a[i][0] + a[i][2]) - a[i][1]);
        const float u3 = a[i][2];
        out[(int64_t)(i * 4 + 0) * NQ * 256] = u0;
        out[(int64_t)(i * 4 + 1) * NQ * 256] = u1;
        out[(int64_t)(i * 4 + 2) * NQ * 256] = u2;
        out[(int64_t)(i * 4 + 3) * NQ * 256] = u3;
    }
}

}  // namespace wino

// Which problems the Winograd kernel serves at all (3x3 / stride 1 / pad 1 is implied by the entry points).
static bool wino_shape_ok(int N, int H, int W, int Ci, int Co) {
    if (N <= 0 || H < 4 || W < 4 || (H & 1) || (W & 1)) return false;
    if (!(Ci == 32 || Ci == 48 || Ci == 64 || Ci == 96 || Ci == 128) || Co % 16 != 0 || Co > 4096) return false;
    if ((int64_t)N * H * W * (Ci > Co ? Ci : Co) * 4 >= 0x7fffffffLL) return false;
    return true;
}

// Block shape: 8 x 4 or 4 x 8 tiles, whichever wastes fewer tiles at the borders; on a tie 8 x 4 for 32 channels (longer
// contiguous pixel runs), 4 x 8 for 64 (its patch is 49 KB of LDS against 54: three workgroups per CU).
static int wino_lbw(int Ht, int Wt, int Ci) {
    const int64_t w3 = (int64_t)cdiv(Wt, 8) * cdiv(Ht, 4), w2 = (int64_t)cdiv(Wt, 4) * cdiv(Ht, 8);
    if (w3 != w2) return w3 < w2 ? 3 : 2;
    return Ci <= 48 ? 3 : 2;                                // (96 / 128 channels: 72 / 95 KB against 80 / 106)
}

// Column tiles per workgroup (template parameter NC).  Default 1.  ADVMIX_WINO_NC=2: both column tiles of a 64-channel conv in one
// workgroup (one staging, one input transform - VERDICT r5 next 3).  Built, parity-green and measured in round 6
// (profiles/r06c_ab_wino_nc.log, EXPERIMENTS M): 64->64 @32x24 13.7 / 13.7 / 15.3 / 15.9 us against 13.4 / 13.4 / 14.2 / 14.5
// (forward + sums / eval / input gradient with mask / sign from c), @64x48 33.4 / 43.2 / 47.3 / 42.5 against 34.9 / 35.5 / 38.8 /
// 38.4, the step 835.2 / 834.6 against 841.6 / 841.5 images/s: half as many workgroups (192: a quarter of the CUs idle) at two
// waves per SIMD and two epilogues in a row cost more than the second staging saved.  Kept as the A/B switch.
static int wino_nc(int Ci, int Co) {
    static const int want = [] { const char* e = getenv("ADVMIX_WINO_NC"); return e ? atoi(e) : 1; }();
    return (want >= 2 && Ci == 64 && Co % 64 == 0) ? 2 : 1;
}

// 0: not served; otherwise the launch's work units (blocks of 32 tiles x column tiles of 32; a workgroup holds wino_nc() of
// them) - what a caller needs to decide whether the chip is filled (ops.py: WINO_MIN_WGS).
extern "C" int advmix_conv_wino_config(int N, int H, int W, int Ci, int Co) {
    if (!wino_shape_ok(N, H, W, Ci, Co)) return 0;
    const int lbw = wino_lbw(H / 2, W / 2, Ci);
    const int64_t wgs = (int64_t)N * cdiv(W / 2, 1 << lbw) * cdiv(H / 2, 32 >> lbw) * cdiv(Co, 32);
    return wgs > 0x7fffffff ? 0x7fffffff : (int)wgs;
}

// floats of one transformed image (forward or input gradient) of a 3x3 Cn x Ck filter bank
extern "C" int64_t advmix_wino_u_floats(int Co, int Ci) { return (int64_t)16 * cdiv(Co, 32) * 32 * Ci; }   // (n padded to whole column tiles)

// Transform the filters of n convs in one launch.  ``ents`` (device): n records {w, u, Cn, Ck, role, first block}, ``blk_ent``
// (device): the record index of each of the ``blocks`` workgroups (a record owns (Cn / 32) * (Ck / 8) consecutive ones).
// The caller builds both once (ops.py: WinoBank).  Replaces nothing in the reference - cuDNN picks and prepares its own
// algorithm behind nn.Conv2d (lib/models/pose_hrnet.py:22-25).
extern "C" int advmix_wino_weights(const void* ents, const int* blk_ent, int blocks, void* stream) {
    if (!ents || !blk_ent || blocks <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(wino::wino_weights, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const wino::WinoEnt*)ents, blk_ent);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

static int wino_launch(int role, wino::WinoP& p, hipStream_t st) {
    const int lbw = wino_lbw(p.Ht, p.Wt, p.Ci);
    p.nbw = cdiv(p.Wt, 1 << lbw);
    p.nblk = p.nbw * cdiv(p.Ht, 32 >> lbw);
    const int nc = wino_nc(p.Ci, p.Co);
    dim3 g(p.N * p.nblk, cdiv(p.Co, 32 * nc));
    const int NQ = p.Ci / 8;
    // the epilogue variant (template parameter VAR): forward + sums / forward + eval BatchNorm / input gradient + BatchNorm backward / plain
    int var;
    if (role == 0 && p.stats && !p.bn_gamma && !p.res && p.act == ADVMIX_ACT_NONE) var = 0;
    else if (role == 0 && p.bn_gamma && !p.stats) var = 1;
    else if (role == 1 && p.bnb_c) var = 2;
    else if (!p.stats && !p.bn_gamma && (role == 1 || p.act == ADVMIX_ACT_NONE)) var = 3;
    else return ADVMIX_EINVAL;                              // (e.g. sums AND an eval epilogue in one launch: not a combination the step uses)
#define WL(NQ_, VAR_, LBW_, KS_, NC_) hipLaunchKernelGGL((wino::conv_wino<NQ_, VAR_, LBW_, KS_, NC_>), g, dim3(256 * KS_), 0, st, p)
#define WI(NQ_, LBW_, KS_, NC_) hipLaunchKernelGGL((wino::conv_wino<NQ_, 0, LBW_, KS_, NC_, 1>), g, dim3(256 * KS_), 0, st, p)
#define WR(NQ_, LBW_, KS_, NC_) do { if (inbn) WI(NQ_, LBW_, KS_, NC_); else if (var == 0) WL(NQ_, 0, LBW_, KS_, NC_); else if (var == 1) WL(NQ_, 1, LBW_, KS_, NC_); else if (var == 2) WL(NQ_, 2, LBW_, KS_, NC_); else WL(NQ_, 3, LBW_, KS_, NC_); } while (0)
    const bool inbn = p.in_slots != nullptr;                // (forward + column sums only: the entry point checks)
    if (inbn && var != 0) return ADVMIX_EINVAL;
    if (NQ == 4 && lbw == 3) WR(4, 3, 1, 1);
    else if (NQ == 4) WR(4, 2, 1, 1);
    else if (NQ == 8 && lbw == 3 && nc == 2) WR(8, 3, 1, 2);
    else if (NQ == 8 && nc == 2) WR(8, 2, 1, 2);
    else if (NQ == 8 && lbw == 3) WR(8, 3, 1, 1);
    else if (NQ == 8) WR(8, 2, 1, 1);
    else if (NQ == 16 && lbw == 3) WR(16, 3, 2, 1);
    else if (NQ == 16) WR(16, 2, 2, 1);
    else if (NQ == 6 && lbw == 3) WR(6, 3, 1, 1);           // HRNet-W48's 48- and 96-channel branches
    else if (NQ == 6) WR(6, 2, 1, 1);
    else if (NQ == 12 && lbw == 3) WR(12, 3, 1, 1);
    else if (NQ == 12) WR(12, 2, 1, 1);
    else return ADVMIX_EINVAL;
#undef WI
#undef WR
#undef WL
    if (advmix_opts().trace_shapes) {
        char nm[48];
        snprintf(nm, sizeof nm, "conv_wino<%d, %d, %d, %d, %d, %d>", NQ, var, lbw, NQ == 16 ? 2 : 1, nc, inbn ? 1 : 0);
        advmix_trace_launch(nm, g, role == 0 ? (p.stats ? (inbn ? "bn_in+fwd+sums" : "fwd+sums") : (p.bn_gamma ? "fwd+bn_eval" : "fwd")) : (p.bnb_c ? "dgrad+bnb" : "dgrad"),
                            p.N, p.H, p.W, p.Ci, p.H, p.W, p.Co, 3, 3, 1, 2.0 * p.N * (double)p.H * p.W * p.Co * p.Ci * 9);
    }
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

static int wino_fill(wino::WinoP& p, const float* x, const float* u, float* y, int N, int H, int W, int Ci, int Co) {
    if (!x || !u || !y || !wino_shape_ok(N, H, W, Ci, Co)) return ADVMIX_EINVAL;
    p = wino::WinoP{};
    p.x = x; p.u = u; p.y = y;
    p.N = N; p.H = H; p.W = W; p.Ci = Ci; p.Co = Co;
    p.Ht = H / 2; p.Wt = W / 2;
    p.xbytes = (int)((int64_t)N * H * W * Ci * 4);
    p.ybytes = (int)((int64_t)N * H * W * Co * 4);
    p.ubytes = (int)((int64_t)16 * cdiv(Co, 32) * 32 * Ci * 4);
    static const int xcd_remap = [] { const char* e = getenv("ADVMIX_XCD_REMAP"); return e ? atoi(e) : 1; }();
    p.xcd_remap = xcd_remap;
    return ADVMIX_OK;
}

static int wino_slots(const int* stats_ns) {
    int ns = stats_ns && *stats_ns > 0 ? *stats_ns : advmix_opts().stat_slots;
    if (ns <= 0 || ns > ADVMIX_STAT_SLOTS_MAX || (ns & (ns - 1))) ns = 16;
    return ns;
}

// advmix_conv_fwd_ex for a 3x3 / stride 1 / pad 1 conv whose filters were transformed by advmix_wino_weights (role 0 image
// ``u``): y = act(BN_eval(conv(x)) + residual) and / or the column (sum, sum of squares) of the raw output into ``stats``
// ([2][*stats_ns][Co], zero on entry; *stats_ns: in = slots to use or 0, out = slots used).  Returns ADVMIX_EINVAL (nothing
// launched) for shapes the kernel does not serve: the caller runs advmix_conv_fwd_ex on the untransformed filters.
// Semantics: lib/models/pose_hrnet.py:22-57 (conv3x3 + BatchNorm2d (+ residual) + ReLU of a BasicBlock).
extern "C" int advmix_conv3x3_wino_fwd(const float* x, const float* u, float* y, int N, int H, int W, int Ci, int Co,
                                       const float* bn_gamma, const float* bn_beta, const float* bn_rm, const float* bn_rv,
                                       float bn_eps, const float* residual, int act, double* stats, int* stats_ns, void* stream) {
    if ((bn_gamma != nullptr) != (bn_beta && bn_rm && bn_rv)) return ADVMIX_EINVAL;
    if (stats && !stats_ns) return ADVMIX_EINVAL;
    if (advmix_opts().deterministic && stats) return ADVMIX_EINVAL;        // fp64 atomics: the ordered form is conv_direct's
    wino::WinoP p;
    int rc = wino_fill(p, x, u, y, N, H, W, Ci, Co);
    if (rc) return rc;
    p.bn_gamma = bn_gamma; p.bn_beta = bn_beta; p.bn_rm = bn_rm; p.bn_rv = bn_rv; p.bn_eps = bn_eps;
    p.res = residual; p.act = act; p.stats = stats;
    p.stats_nbg = wino_slots(stats_ns);
    rc = wino_launch(0, p, (hipStream_t)stream);
    if (rc == ADVMIX_OK && stats_ns) *stats_ns = p.stats_nbg;
    return rc;
}

// advmix_conv3x3_wino_fwd (train role: raw output + column sums into ``stats``) whose input is the RAW output c of the preceding
// conv, with that conv's train-mode BatchNorm + ReLU applied while the patch is staged: y = conv(relu(BN_train(c))).  in_slots:
// the column (sum, sum of squares) of c as the producer's epilogue left them, [2][in_ns][Ci] with in_ns <= 16; the launch derives
// mean / invstd (biased variance, eps) from them, writes them to in_mean / in_invstd (the backward pass's saved statistics),
// updates in_rmean / in_rvar (momentum, unbiased variance; may be NULL) and increments *in_nbt (may be NULL) - everything
// advmix_norm_apply_slots does, without its launch and without the activation tensor.  One launch replaces
// norm_apply_slots + conv for the inner edge conv1 -> bn1 -> relu -> conv2 of lib/models/pose_hrnet.py:41-57 (BasicBlock) and
// :77-88 (Bottleneck conv1 -> conv2).  ADVMIX_EINVAL (nothing launched) where advmix_conv3x3_wino_fwd refuses, or in_ns > 16.
extern "C" int advmix_conv3x3_wino_fwd_inbn(const float* c_in, const float* u, float* y, int N, int H, int W, int Ci, int Co,
                                            const double* in_slots, int in_ns, const float* in_gamma, const float* in_beta,
                                            float in_eps, float* in_mean, float* in_invstd, float* in_rmean, float* in_rvar,
                                            long long* in_nbt, float in_momentum, double* stats, int* stats_ns, void* stream) {
    if (!in_slots || !in_gamma || !in_beta || !in_mean || !in_invstd || !stats || !stats_ns) return ADVMIX_EINVAL;
    if (in_ns < 1 || in_ns > 16 || (in_ns & (in_ns - 1)) || (in_rmean != nullptr) != (in_rvar != nullptr)) return ADVMIX_EINVAL;
    if (advmix_opts().deterministic) return ADVMIX_EINVAL;
    wino::WinoP p;
    int rc = wino_fill(p, c_in, u, y, N, H, W, Ci, Co);
    if (rc) return rc;
    p.stats = stats;
    p.stats_nbg = wino_slots(stats_ns);
    p.in_slots = in_slots; p.in_ns = in_ns; p.in_rows = (double)N * H * W; p.in_eps = in_eps; p.in_momentum = in_momentum;
    p.in_gamma = in_gamma; p.in_beta = in_beta; p.in_mean = in_mean; p.in_invstd = in_invstd;
    p.in_rmean = in_rmean; p.in_rvar = in_rvar; p.in_nbt = in_nbt;
    rc = wino_launch(0, p, (hipStream_t)stream);
    if (rc == ADVMIX_OK) *stats_ns = p.stats_nbg;
    return rc;
}

// advmix_conv_tr_w_add / advmix_conv_tr_w_bnb for the same convs: dx = conv(dy, rotated transposed filters) + addend from the
// role 1 image ``u`` (n = Cin, k = Cout); with ``bn_c`` the BatchNorm-backward epilogue of advmix_conv_tr_w_bnb (same
// arguments, same arithmetic).  dy: [N,H,W,Co], dx / addend / bn_c / mask: [N,H,W,Ci].
extern "C" int advmix_conv3x3_wino_dgrad(const float* dy, const float* u, const float* addend, float* dx, int N, int H, int W,
                                         int Co, int Ci, const unsigned char* act_mask, const float* bn_c, const float* bn_mean,
                                         const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int act,
                                         double* stats, int* stats_ns, void* stream) {
    if (bn_c) {
        if (!bn_mean || !bn_invstd || !stats || !stats_ns) return ADVMIX_EINVAL;
        if (act != ADVMIX_ACT_NONE && !act_mask && !(bn_gamma && bn_beta)) return ADVMIX_EINVAL;
        if (advmix_opts().deterministic) return ADVMIX_EINVAL;
    } else if (stats) {
        return ADVMIX_EINVAL;
    }
    wino::WinoP p;
    int rc = wino_fill(p, dy, u, dx, N, H, W, Co, Ci);      // the gradient conv reads Co channels and writes Ci
    if (rc) return rc;
    p.res = addend;
    if (bn_c) {
        p.stats = stats; p.stats_nbg = wino_slots(stats_ns);
        p.bnb_mask = act_mask; p.bnb_c = bn_c; p.bnb_mean = bn_mean; p.bnb_invstd = bn_invstd;
        p.bnb_gamma = bn_gamma; p.bnb_beta = bn_beta; p.bnb_act = act;
    }
    rc = wino_launch(1, p, (hipStream_t)stream);
    if (rc == ADVMIX_OK && bn_c) *stats_ns = p.stats_nbg;
    return rc;
}
