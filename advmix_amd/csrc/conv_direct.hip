// Second-generation implicit-GEMM convolution for gfx950 (fp32 MFMA 32x32x2).
//
// What limited conv_igemm (rocprofv3 PMC, profiles/r01_pmc_conv32.txt): the MFMA pipe was 48 %
// busy while waves sat 29 % of their life at barriers / s_waitcnt - every 16-deep k tile cost
// two workgroup barriers and an LDS round trip for BOTH operands.  Here:
//   * the activation operand never touches LDS: each lane loads its own MFMA A-fragment
//     (row = output pixel, 4 consecutive channels = 16 B) straight from HBM/L2 into VGPRs with
//     bounds-checked buffer loads - an out-of-image tap gets an out-of-range offset and the
//     hardware returns 0.0f, so zero padding costs no select instructions;
//   * loads for tap-chunk c+1 are issued before the MFMAs of chunk c (two named register sets,
//     statically indexed), so a chunk's ~1000-2000 MFMA cycles cover the load latency;
//     (a one-workgroup-per-CU persistent variant with cross-tile prefetch was measured and is
//     SLOWER - 41.7 vs 26.2 us - one wave per SIMD cannot cover its own barrier/stage gaps);
//   * only the weight chunk [BN][KC] is staged through LDS, double-buffered: ONE barrier per
//     KC-deep chunk (16-32 MFMAs per wave) instead of two per 16-deep tile;
//   * optional split over the K chunks (gridDim.z slices, fp32 atomics into a pre-zeroed
//     output) for the low-resolution HRNet branches whose M x N tile grid is only 24-192
//     workgroups on a 256-CU chip.
// MODE 0 = forward gather (Conv2d fwd / ConvTranspose2d dgrad), MODE 1 = phase-decomposed
// transposed gather (Conv2d dgrad / ConvTranspose2d fwd), exactly as conv_mfma.hip.
#include "common.h"
#include <stdlib.h>
#include <stdio.h>



namespace direct {

struct ConvD {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    int N, Hi, Wi, Ci;
    int Ho, Wo, Co;
    int R, S, stride, pad;
    int xbytes, wbytes, ybytes;   // buffer sizes for the hardware range check
    int nsplit;             // K-chunk slices (gridDim.z = phases * nsplit)
    // optional fused epilogue (MODE 0, no K split): eval-mode BatchNorm + residual + activation, and/or
    // per-wave column sums of the raw conv output for a following train-mode BatchNorm
    const float *bn_gamma, *bn_beta, *bn_rm, *bn_rv, *res;
    float bn_eps;
    int act;
    double* stats;          // [2][stats_nbg][Co] (stats_tiles: [2][Co][stats_nbg]): (sum, sum of squares), wave slabs folded onto
    int stats_nbg;          // stats_nbg slots with fp64 atomics; must be zero on entry
    // MODE 1 + EPI: this launch is the input gradient g0 = dL/dy of a tensor y = act(BN(c) + residual) (train mode).
    // The epilogue multiplies by the activation's slope (through y), writes g = g0 * act'(y) and accumulates the two
    // BatchNorm-backward channel sums (sum g, sum g * xhat) into ``stats``: the separate statistics pass over
    // (dy, y, c) and its finalize launch disappear (norm.hip: norm_bwd_apply_slots consumes the slots).
    // The sign of y comes from ``bnb_mask`` (one bit per element, a byte per 4 channels, written by norm_apply_slots: 1/16
    // of y's bytes) or, for y = act(BN(c)) WITHOUT a residual, from c itself: fmaf((c - mean) * invstd, gamma, beta) > 0
    // is exactly the expression norm_apply_slots evaluated (round 4; the fp32 y was read for its sign alone).
    const unsigned char* bnb_mask;
    const float *bnb_c, *bnb_mean, *bnb_invstd, *bnb_gamma, *bnb_beta;
    int bnb_act;
    int stats_tiles;        // 1 (deterministic mode): every workgroup STORES its column sums in a slot of its own -
                            // stats[2][Co][stats_nbg] with stats_nbg = row tiles x phases - instead of fp64 atomics
    int xcd_remap;          // 1: row-tile order remapped so that each XCD (and its L2) owns a CONTIGUOUS range of row tiles
    int wimg;               // > 0 (MODE 0, conv_wino4.hip's batched GEMM): image n multiplies its OWN filters w + n * wimg (floats;
                            // wbytes = one image's); a row tile never straddles two images (the host checks BM | Ho * Wo)
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}

// Cache policy of the epilogue stores (gfx940+ aux bits: 1 = sc0, 2 = nt, 16 = sc1).  sc1 = write-through: the output
// leaves the XCD's L2 while the kernel still runs instead of being flushed at the kernel boundary (the 8 per-XCD L2s
// are not coherent, so every boundary writes back what the predecessor left dirty: ~B / 6 TB/s for B dirty bytes).
// Measured on the dominant conv (12.6 MB output; tools/build_variant.sh, gpurun_out/r2e/store_policy.log):
// launch to launch 25.3 -> 23.7 us (+BN sums), 23.7 -> 22.4 us (plain); nt 24.5 / 22.8; the 4-stream aggregate
// (95.5 TFLOP/s) and the step (536 -> 540 images/s) barely move - concurrent lanes already hide the boundary.
#ifndef CD_STORE_AUX
#define CD_STORE_AUX 16
#endif
constexpr unsigned OOB = 0x80000000u;   // >= any buffer size we accept -> load returns 0

// BT: the weight operand is given k-major, w[Ck][R][S][Cn] (a Conv2d's own [Co][R][S][Ci] seen from its
// input gradient, a ConvTranspose2d's own [Ci][R][S][Co] seen from its forward): a thread loads 4
// consecutive n of one k and scatters them into the [n][k] LDS image, so no re-layout kernel is needed.
// v_mfma_f32_32x32x2: lane l holds row l % 32 and k-lane l / 32 (of 2); a fragment load touches 32 pixel rows x 32 B.
// (The 16x16x4 shape - 16 rows x 64 B per load - was built, measured 25.4 vs 25.9 us alone and 455 vs 468 images/s
// in the step, and removed.)
constexpr int MR = 32;
struct MS {
    typedef f32x16 acc_t;
    static constexpr int NR = 16;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }
};

// LDS floats of one weight buffer (two are used) for a tile configuration - shared by the kernels that instantiate
// conv_body (conv_direct: one configuration; conv_group: the largest of its configurations)
template <int TM, int TN, int WM, int WN, int KC, int NW = 4>
struct Geo {
    static constexpr int WK = NW / (WM * WN);
    static constexpr int RM = TM * (32 / MR), RN = TN * (32 / MR);
    static constexpr int BN = 32 * TN * WN;
    static constexpr int LDB = KC + 4;
    static constexpr int RED = WK > 1 ? NW * RM * RN * MS::NR * 64 : 0;
    static constexpr int BSZ = (WK * BN * LDB * 2 > RED) ? WK * BN * LDB : (RED + 1) / 2;
    static constexpr int TP = 36;                          // pitch of the epilogue's wave-private transposer (floats)
    static constexpr int TSZ = NW * (2 * MS::NR / WK) * TP; // NW waves x the 2 * NR / WK rows a wave finishes x 32 columns
};

// One workgroup's tile (bx, by) of slice / phase bz.  ``Bs0``: 2 * Geo::BSZ floats of LDS, ``Ts0``: Geo::TSZ floats,
// ``taptab``: 64 int4.
template <int TM, int TN, int WM, int WN, int KC, int MODE, bool SPLIT, bool BT, bool EPI, int NW = 4>
__device__ __forceinline__ void conv_body(const ConvD p, const int bx, const int by, const int bz, float* const Bs0,
                                          float* const Ts0, int4* const taptab) {
    // WK waves of the workgroup split K BETWEEN THEM (WM x WN x WK = 4 waves): the low-resolution layers have too
    // few output tiles to fill the chip; instead of slicing K across workgroups (SPLIT: zero-fill + fp32 atomics,
    // no fused epilogue) a 32x32 tile is computed by four waves that each take every fourth K chunk and meet
    // in LDS, so the fused BN / residual / activation / statistics epilogue still applies.
    // NW = 8 (round 3): two row tiles share ONE staging of the weight chunks - the low-resolution layers are bound by
    // weight traffic (3x3 256->256: 2.36 MB of weights re-read by each of 48 row tiles) - and each is K-split over four waves.
    static_assert(NW % (WM * WN) == 0 && (NW == 4 || NW == 8), "WM x WN x WK = NW waves");
    constexpr int NT = 64 * NW;                // threads of the workgroup
    constexpr int WK = NW / (WM * WN);
    static_assert(!(SPLIT && WK > 1), "one kind of K split at a time");
    constexpr int NSUB = 32 / MR;              // MFMA tiles per 32 rows / 32 columns
    constexpr int KL = 64 / MR;                // k-lanes: lanes that hold different k of the same row
    constexpr int RM = TM * NSUB, RN = TN * NSUB;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int KQ = KC / (4 * KL);          // float4 k-groups per chunk (4 k per k-lane each)
    constexpr int LDB = KC + 4;
    constexpr int BCH = BN * KC / 4;           // float4 staging slots per chunk
    constexpr int BSL = (BCH * WK + NT - 1) / NT;
    constexpr int RED = WK > 1 ? NW * RM * RN * MS::NR * 64 : 0;          // floats of the cross-wave reduction (all waves)
    constexpr int BSZ = (WK * BN * LDB * 2 > RED) ? WK * BN * LDB : (RED + 1) / 2;

    static_assert(BSZ == Geo<TM, TN, WM, WN, KC, NW>::BSZ, "Geo mirrors these constants");
    float (*const Bs)[BSZ] = reinterpret_cast<float (*)[BSZ]>(Bs0);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane % MR, lh = lane / MR;             // row within the MFMA tile, k-lane
    const int wk = wid / (WM * WN), wmn = wid % (WM * WN);
    const int wm = wmn / WN, wn = wmn % WN;

    int Hp, Wp, Th, Tw, rh = 0, rw = 0, phh = 0, phw = 0, zsl = bz;
    if (MODE == 0) {
        Hp = p.Ho; Wp = p.Wo; Th = p.R; Tw = p.S;
    } else {
        const int phase = bz / p.nsplit;
        zsl = bz - phase * p.nsplit;
        rh = phase / p.stride; rw = phase % p.stride;
        Hp = p.Ho > rh ? (p.Ho - rh + p.stride - 1) / p.stride : 0;
        Wp = p.Wo > rw ? (p.Wo - rw + p.stride - 1) / p.stride : 0;
        phh = (rh + p.pad) % p.stride; phw = (rw + p.pad) % p.stride;
        Th = phh < p.R ? (p.R - phh + p.stride - 1) / p.stride : 0;
        Tw = phw < p.S ? (p.S - phw + p.stride - 1) / p.stride : 0;
    }
    const int Mp = p.N * Hp * Wp;
    // Workgroups are dealt to the 8 XCDs round-robin in launch order, so neighbouring row tiles - which share their 3x3
    // halo rows - land on eight different L2s and each fetches the halo from HBM again (dominant conv: 22.2 MB read for
    // 12.6 MB of input, x1.76 = (128 + 2 * 48) / 128 pixels per tile, profiles/r02c_pmc_conv32_epi.json).  With the
    // row-tile index remapped (tile = (bx % 8) * (gx / 8) + bx / 8) XCD k works through tiles [k * gx / 8, (k + 1) * gx / 8)
    // in order: a tile's halo is what its predecessor on the SAME XCD just loaded.
    int bxr = bx;
    if (p.xcd_remap && (gridDim.x & 7) == 0 && gridDim.x >= 16 && bx < (int)gridDim.x)
        bxr = (bx & 7) * ((int)gridDim.x >> 3) + (bx >> 3);
    const int m0 = bxr * BM, n0 = by * BN;
    if (m0 >= Mp) return;
    const int ntaps = Th * Tw;
    const int cpt = p.Ci / KC;                              // chunks per tap
    const int nch_all = ntaps * cpt;
    // this slice's chunk range
    int ch_lo = 0, ch_hi = nch_all;
    if (SPLIT) {
        const int per = (nch_all + p.nsplit - 1) / p.nsplit;
        ch_lo = zsl * per;
        ch_hi = min(nch_all, ch_lo + per);
        if (ch_lo >= ch_hi && !(zsl == 0)) return;          // empty slice (slice 0 still writes bias/zeros)
    }
    const int Kfull = p.R * p.S * p.Ci;

    if (tid < ntaps) {
        int4 t;
        if (MODE == 0) {
            t.x = tid / p.S; t.y = tid % p.S; t.z = tid * p.Ci;
        } else {
            int th = tid / Tw, tw = tid % Tw;
            t.x = -th; t.y = -tw; t.z = ((phh + p.stride * th) * p.S + phw + p.stride * tw) * p.Ci;
        }
        t.w = t.z / p.Ci;                                   // r*S + s (BT addressing)
        taptab[tid] = t;
    }

    // ---- per-lane A rows -----------------------------------------------------------------
    int a_nb[RM], a_h[RM], a_w[RM];
    bool a_ok[RM];
#pragma unroll
    for (int t = 0; t < RM; ++t) {
        int m = m0 + wm * TM * 32 + t * MR + l31;
        a_ok[t] = m < Mp;
        int mm = a_ok[t] ? m : 0;
        int n = mm / (Hp * Wp);
        int rem = mm - n * (Hp * Wp);
        int hi_ = rem / Wp, wi_ = rem - hi_ * Wp;
        a_nb[t] = n * p.Hi * p.Wi;
        if (MODE == 0) {
            a_h[t] = hi_ * p.stride - p.pad;
            a_w[t] = wi_ * p.stride - p.pad;
        } else {
            a_h[t] = (rh + hi_ * p.stride + p.pad - phh) / p.stride;
            a_w[t] = (rw + wi_ * p.stride + p.pad - phw) / p.stride;
        }
    }
    // ---- per-thread B staging slots --------------------------------------------------------
    int b_row[BSL], b_k4[BSL], b_kk[BSL];
    unsigned b_off[BSL];
#pragma unroll
    for (int i = 0; i < BSL; ++i) {
        int s = tid + NT * i;
        b_kk[i] = s / BCH;                                  // which of the WK chunks of a step this slot stages
        s -= b_kk[i] * BCH;
        const bool in = b_kk[i] < WK;
        if (!BT) {
            b_row[i] = s / (KC / 4);                        // n
            b_k4[i] = (s % (KC / 4)) * 4;                   // k (4 consecutive)
            bool ok = in && (n0 + b_row[i]) < p.Co;
            b_off[i] = ok ? (unsigned)(((n0 + b_row[i]) * Kfull + b_k4[i]) * 4) : OOB;
        } else {
            b_k4[i] = s / (BN / 4);                         // k (one)
            b_row[i] = (s % (BN / 4)) * 4;                  // n (4 consecutive; Co % 4 == 0 is checked on the host)
            bool ok = in && (n0 + b_row[i]) < p.Co;
            b_off[i] = ok ? (unsigned)((b_k4[i] * p.R * p.S * p.Co + n0 + b_row[i]) * 4) : OOB;
        }
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    const float* const wbase = MODE == 0 && p.wimg > 0 ? p.w + (int64_t)(m0 / (Hp * Wp)) * p.wimg : p.w;   // (uniform)
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, p.wbytes, 0x00020000);
    __syncthreads();                                       // taptab visible

    // ---- epilogue geometry, and its operands fetched NOW ------------------------------------------------------------
    constexpr int RSL = MS::NR / WK;                       // accumulator registers a wave finishes itself
    static_assert(MS::NR % WK == 0, "epilogue slices");
    const int r_lo = WK > 1 ? wk * RSL : 0;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.y), 0, p.ybytes,
                                                                        0x00020000);
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bnb_c ? p.bnb_c : p.y), 0, p.ybytes,
                                                                        0x00020000);
    const __amdgpu_buffer_rsrc_t mkr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bnb_mask ? (const void*)p.bnb_mask : (const void*)p.y),
                                                                         0, p.ybytes >> 4, 0x00020000);
    // element offset of accumulator register r of tile (t, u) in the output (the host checks that y fits 2^31 bytes)
    int e_l31 = l31, e_lh = lh;                            // (made opaque again before the epilogue, see there)
    auto elem_off = [&](int t, int u, int r, bool& valid) -> int {
        const int col = n0 + wn * TN * 32 + u * MR + e_l31;
        int m = m0 + wm * TM * 32 + t * MR + MS::row(r, e_lh);
        valid = m < Mp && col < p.Co;
        if (!valid) m = 0;
        if (MODE == 0 || p.stride == 1)                    // (stride 1: one phase, rows are already output pixels -
            return m * p.Co + col;                         //  no two integer divisions per accumulator register)
        const int n = m / (Hp * Wp);
        const int rem = m - n * (Hp * Wp);
        const int hi_ = rem / Wp, wi_ = rem - hi_ * Wp;
        return ((n * p.Ho + rh + hi_ * p.stride) * p.Wo + rw + wi_ * p.stride) * p.Co + col;
    };
    // 16-byte view of a wave's share of tile (t, u): chunk q = 8 rows x 32 columns, lane -> (row 8q + lane / 8 of the
    // 2 * RSL rows the wave finishes, 4 consecutive columns).  The accumulator layout (one column, 16 scattered rows per
    // lane) costs 16 four-byte memory instructions per lane and operand; through a wave-private LDS transposer the same
    // bytes move as RSL / 4 sixteen-byte ones - the skeleton of this kernel is bound by vector-memory ISSUE, not bytes.
    constexpr int TQ = RSL / 4;
    constexpr int TP = Geo<TM, TN, WM, WN, KC, NW>::TP;
    const bool t128 = !SPLIT && (p.Co & 3) == 0;           // uniform: rows of the output are 16-byte aligned
    int e_lane = lane;
    auto chunk_off = [&](int t, int u, int q, bool& valid) -> int {
        const int col = n0 + wn * TN * 32 + u * MR + ((e_lane & 7) << 2);
        int m = m0 + wm * TM * 32 + t * MR + 2 * r_lo + 8 * q + (e_lane >> 3);
        valid = m < Mp && col < p.Co;
        if (!valid) m = 0;
        if (MODE == 0 || p.stride == 1) return m * p.Co + col;
        const int n = m / (Hp * Wp);
        const int rem = m - n * (Hp * Wp);
        const int hi_ = rem / Wp, wi_ = rem - hi_ * Wp;
        return ((n * p.Ho + rh + hi_ * p.stride) * p.Wo + rw + wi_ * p.stride) * p.Co + col;
    };
    // The epilogue's READ operands - the residual (forward) / the addend (input gradient), and for the BatchNorm-
    // backward epilogue the producer's c and y - are requested before the main loop and sit in registers until it ends.
    // Loaded in the epilogue they were a pure memory phase with the matrix pipe idle, every wave of the launch at once:
    // 3x3 32->32 @64x48 input gradient 23.5 us, + addend 28.0, + BatchNorm-backward sums 36.1.  Single-tile waves only
    // (16 registers per operand).
    constexpr bool PRE = RM * RN == 1 && !SPLIT;
    const bool pre_a_on = PRE && t128 && p.res != nullptr && (MODE == 1 || EPI);
    // The 64x64 four-wave tile (WN == 2: two weight staging slots per thread) does not fit both prefetched operands into
    // 128 registers: its c operand is fetched in the epilogue instead (16-byte loads through the same transposer).
    constexpr bool LATE_C = PRE && EPI && MODE == 1 && WN == 2;
    const bool pre_c_on = PRE && t128 && EPI && MODE == 1 && !LATE_C;
    const bool late_c_on = LATE_C && t128;
    f32x4 pq_a[TQ], pq_c[TQ];
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
        pq_a[q] = f32x4{0.f, 0.f, 0.f, 0.f}; pq_c[q] = pq_a[q];
        if (PRE) {
            bool valid;
            const int off = chunk_off(0, 0, q, valid);
            const unsigned boff = valid ? (unsigned)off * 4u : OOB;
            if (pre_a_on) pq_a[q] = bload(rr, boff);
            if (pre_c_on) pq_c[q] = bload(cr, boff);
        }
    }
    // The activation mask of the BatchNorm-backward epilogue: ONE 4-byte load per lane and MFMA tile, requested now.  Lane
    // (l31, lh) fetches the 16 channels [16 * lh, 16 * lh + 16) of ITS row l31 of the tile (4 mask bytes); in the epilogue
    // the lane that needs (row, column) gets that word from lane row + 32 * (column / 16) with one ds_bpermute.
    const bool mask_on = EPI && MODE == 1 && p.bnb_mask != nullptr && p.bnb_act != ADVMIX_ACT_NONE;
    unsigned mw[RM][RN];
#pragma unroll
    for (int t = 0; t < RM; ++t)
#pragma unroll
        for (int u = 0; u < RN; ++u) {
            mw[t][u] = 0u;
            if (EPI && MODE == 1 && mask_on) {
                const int col = n0 + wn * TN * 32 + u * MR + 16 * lh;
                int m = m0 + wm * TM * 32 + t * MR + l31;
                const bool valid = m < Mp && col < p.Co;
                if (!valid) m = 0;
                int pix = m;
                if (!(MODE == 0 || p.stride == 1)) {
                    const int n = m / (Hp * Wp);
                    const int rem = m - n * (Hp * Wp);
                    const int hi_ = rem / Wp, wi_ = rem - hi_ * Wp;
                    pix = (n * p.Ho + rh + hi_ * p.stride) * p.Wo + rw + wi_ * p.stride;
                }
                mw[t][u] = __builtin_amdgcn_raw_buffer_load_b32(mkr, valid ? (unsigned)(pix * (p.Co >> 2) + (col >> 2)) : OOB, 0, 0);
            }
        }

    // Two named register sets (statically indexed): one being multiplied, one in flight.  A third
    // set (two chunks of look-ahead) was measured and is NOT faster: 43.4 % vs 43.8 % on the
    // 3x3 32->32 conv and 47 % vs 60 % on the 128x64 tile (130 VGPRs -> 2 waves/SIMD), i.e. the
    // kernel is not bound by load latency per wave.
    f32x4 A0[RM][KQ], A1[RM][KQ], Br[BSL];
    MS::acc_t acc[RM][RN];
#pragma unroll
    for (int t = 0; t < RM; ++t)
#pragma unroll
        for (int u = 0; u < RN; ++u)
#pragma unroll
            for (int r = 0; r < MS::NR; ++r) acc[t][u][r] = 0.f;

    // chunk cursor (of the STEP being issued: WK consecutive chunks, wave wk multiplies chunk cur + wk)
    int cur = ch_lo;

    // (tap, first channel) cursors of the chunk each wave multiplies / each staging slot loads next; advanced by WK
    // chunks per step without divisions
    int a_tap = (ch_lo + wk) / cpt, a_c0 = ((ch_lo + wk) - a_tap * cpt) * KC;
    int s_tap[BSL], s_c0[BSL];
#pragma unroll
    for (int i = 0; i < BSL; ++i) {
        const int c = ch_lo + (WK == 1 ? 0 : b_kk[i]);
        s_tap[i] = c / cpt;
        s_c0[i] = (c - s_tap[i] * cpt) * KC;
    }
    auto advance = [&](int& tap_, int& c0_) {
        c0_ += WK * KC;
        while (c0_ >= p.Ci) { c0_ -= p.Ci; ++tap_; }
    };
    auto issue = [&](f32x4 (&A)[RM][KQ]) {                 // loads of the step starting at chunk ``cur``
        const int ck = cur + wk;                           // this wave's chunk; past the end -> zeros
        int tap = a_tap;
        const int c0 = a_c0;
        const bool live = ck < ch_hi;
        if (!live) tap = 0;
        const int4 tt = taptab[tap];
#pragma unroll
        for (int t = 0; t < RM; ++t) {
            int hi = a_h[t] + tt.x, wi = a_w[t] + tt.y;
            bool ok = live && a_ok[t] && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
            unsigned off = ok ? (unsigned)(((a_nb[t] + hi * p.Wi + wi) * p.Ci + c0 + lh * 4) * 4) : OOB;
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                A[t][q] = bload(xr, off + q * (16 * KL));
            }
        }
#pragma unroll
        for (int i = 0; i < BSL; ++i) {
            const int cb = WK == 1 ? ck : cur + b_kk[i];
            const int tb = s_tap[i], c0b = s_c0[i];
            advance(s_tap[i], s_c0[i]);
            const bool lb = cb < ch_hi && b_off[i] != OOB;
            const int4 tq = taptab[lb ? tb : 0];
            Br[i] = bload(wr, !lb ? OOB
                              : b_off[i] + (BT ? (unsigned)(((c0b * p.R * p.S + tq.w) * p.Co) * 4)
                                             : (unsigned)((tq.z + c0b) * 4)));
        }
        cur += WK;
        advance(a_tap, a_c0);
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < BSL; ++i)
            if (tid + NT * i < BCH * WK) {
                float* dst = &Bs[buf][b_kk[i] * (BN * LDB)];
                if (!BT) {
                    *reinterpret_cast<f32x4*>(&dst[b_row[i] * LDB + b_k4[i]]) = Br[i];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[(b_row[i] + e) * LDB + b_k4[i]] = Br[i][e];
                }
            }
    };
    auto compute = [&](f32x4 (&A)[RM][KQ], int buf) {
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            f32x4 b[RN];
#pragma unroll
            for (int u = 0; u < RN; ++u) {
                b[u] = *reinterpret_cast<const f32x4*>(
                    &Bs[buf][wk * (BN * LDB) + (wn * TN * 32 + u * MR + l31) * LDB + q * (4 * KL) + lh * 4]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < RM; ++t)
#pragma unroll
                    for (int u = 0; u < RN; ++u) {
                        acc[t][u] = MS::mma(A[t][q][j], b[u][j], acc[t][u]);
                    }
        }
    };

    const int nch = (ch_hi - ch_lo + WK - 1) / WK;         // steps
    if (nch > 0) {
        // Steady-state loop with NO conditional load issue and one body: the conditional "issue the next chunk
        // if there is one" of the first version merged two control-flow paths in front of the MFMAs, and the
        // compiler's conservative s_waitcnt for the merge (vmcnt(4..1)) made every chunk's MFMAs wait for the
        // loads of the NEXT chunk that had just been issued - load latency was added to, not hidden behind, the
        // multiplies, and the two unrolled halves ping-ponged the accumulator between register ranges through
        // v_accvgpr moves.  The last chunk is peeled; the in-flight set is copied into the multiply set with
        // 16 register moves per chunk.
        issue(A0);
            stage(0);
        __syncthreads();
        int buf = 0;
        for (int ci = 0; ci + 1 < nch; ++ci) {
            issue(A1);                                     // chunk ci + 1 in flight ...
            __builtin_amdgcn_sched_barrier(0);             // (the scheduler otherwise sinks the weight load below the MFMAs)
            compute(A0, buf);                              // ... while chunk ci is multiplied
            stage(buf ^ 1);
            __syncthreads();
#pragma unroll
            for (int t = 0; t < RM; ++t)
#pragma unroll
                for (int q = 0; q < KQ; ++q) A0[t][q] = A1[t][q];
            buf ^= 1;
        }
        compute(A0, buf);
    }

    // ---- epilogue -----------------------------------------------------------------------------
    // BN column sums: the WM waves that share a column meet in LDS (the weight buffer is free now), so a
    // workgroup issues ONE pair of fp64 atomics per column instead of one per wave
    float* sred = &Bs[0][0];                               // [2][WK * WM][BN]
    const bool stats = EPI && p.stats != nullptr;          // uniform over the grid
    if (WK > 1) {                                          // the WK partial tiles meet in LDS
        float* red = &Bs[0][0];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RM; ++t)
#pragma unroll
            for (int u = 0; u < RN; ++u)
#pragma unroll
                for (int r = 0; r < MS::NR; ++r)
                    red[(((wk * WM * WN + wmn) * RM * RN + t * RN + u) * MS::NR + r) * 64 + lane] = acc[t][u][r];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RM; ++t)
#pragma unroll
            for (int u = 0; u < RN; ++u)
#pragma unroll
                for (int r = 0; r < RSL; ++r) {
                    float v = 0.f;
#pragma unroll
                    for (int k = 0; k < WK; ++k)                  // the WK waves that own the same (wm, wn) tile
                        v += red[(((k * WM * WN + wmn) * RM * RN + t * RN + u) * MS::NR + r_lo + r) * 64 + lane];
                    acc[t][u][r] = v;                          // slot r now holds accumulator register r_lo + r
                }
    }
    // The offsets of the prefetched operands are NOT kept across the main loop (16 more live registers per lane): the
    // epilogue recomputes them from lane ids the compiler cannot connect to the earlier ones.
    asm volatile("" : "+v"(e_l31), "+v"(e_lh), "+v"(e_lane));
    if (stats) __syncthreads();                            // every wave is done reading Bs
    // The wave-private transposer: 2 * RSL rows x 32 columns at a 36-float pitch (16-byte aligned rows; a 4-byte access
    // of the accumulator layout - row MS::row(r, lh), column l31 - is bank-conflict free per half wave).  LDS accesses
    // of ONE wave execute in issue order, so a write followed by a read of other lanes' data needs no barrier - only
    // that the compiler keeps the order (the accesses go through one pointer it cannot prove disjoint, plus a
    // scheduling fence).
    float* const Tx = Ts0 + wid * (2 * RSL * TP);
    auto wave_fence = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    float pre_a[RSL], pre_c[RSL];
    auto to_acc_layout = [&](const f32x4 (&q)[TQ], float (&o)[RSL]) {
#pragma unroll
        for (int i = 0; i < TQ; ++i)
            *reinterpret_cast<f32x4*>(&Tx[(8 * i + (lane >> 3)) * TP + ((lane & 7) << 2)]) = q[i];
        wave_fence();
#pragma unroll
        for (int r = 0; r < RSL; ++r) o[r] = Tx[MS::row(r, lh) * TP + l31];
        wave_fence();
    };
#pragma unroll
    for (int r = 0; r < RSL; ++r) { pre_a[r] = 0.f; pre_c[r] = 0.f; }
    if (PRE) {
        if (pre_a_on) to_acc_layout(pq_a, pre_a);
        if (pre_c_on) to_acc_layout(pq_c, pre_c);
    }
    const int mshift = ((e_l31 >> 2) & 3) * 8 + (e_l31 & 3);   // this lane's column within its mask word
    // Straight-line stores: out-of-tile lanes get an out-of-range offset and the hardware drops their write
    // (reads return 0), so there is no per-element branch.  The first version branched around every store, and
    // the compiler's s_waitcnt for the bias value at each re-convergence (vmcnt(0) - which on gfx9 also counts
    // STORES) made every store wait for the previous one to be acknowledged.
#pragma unroll
    for (int u = 0; u < RN; ++u) {
        const int col = n0 + wn * TN * 32 + u * MR + l31;
        const bool cvalid = col < p.Co;
        const float bv = (cvalid && p.bias && (!SPLIT || zsl == 0)) ? p.bias[col] : 0.f;
        float bn_is = 1.f, bn_g = 1.f, bn_b = 0.f, bn_m = 0.f;
        const bool bn = EPI && p.bn_gamma != nullptr;
        if (bn && cvalid) {
            bn_is = 1.0f / sqrtf(p.bn_rv[col] + p.bn_eps);
            bn_g = p.bn_gamma[col]; bn_b = p.bn_beta[col]; bn_m = p.bn_rm[col];
        }
        const bool addend = MODE == 1 && p.res && (!SPLIT || zsl == 0);
        const bool bnb = EPI && MODE == 1;                  // BatchNorm-backward producer (see ConvD)
        float bb_mu = 0.f, bb_is = 0.f, bb_g = 0.f, bb_b = 0.f;
        if (bnb && cvalid) { bb_mu = p.bnb_mean[col]; bb_is = p.bnb_invstd[col]; }
        const bool recompute = bnb && !mask_on && p.bnb_act != ADVMIX_ACT_NONE;     // sign of y = act(BN(c)) from c itself
        if (recompute && cvalid) { bb_g = p.bnb_gamma[col]; bb_b = p.bnb_beta[col]; }
        const float bb_slope = act_neg_slope(p.bnb_act);
        // element offsets in the accumulator layout are only needed on the scalar path (Co % 4 != 0, K split across
        // the grid): everything else moves in 16-byte chunks through the transposer
        const bool op_a = addend || (EPI && MODE == 0 && p.res != nullptr);
        const bool need_off = !t128;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < RM; ++t) {
            // read operands that were not prefetched (multi-tile waves): 16-byte loads now, same transposer
            float oa[RSL], oc[RSL];
#pragma unroll
            for (int r = 0; r < RSL; ++r) { oa[r] = pre_a[r]; oc[r] = pre_c[r]; }
            if (LATE_C && late_c_on) {
                f32x4 qc[TQ];
#pragma unroll
                for (int q = 0; q < TQ; ++q) {
                    bool valid;
                    const int off = chunk_off(t, u, q, valid);
                    qc[q] = bload(cr, valid ? (unsigned)off * 4u : OOB);
                }
                to_acc_layout(qc, oc);
            }
            if (t128 && !PRE) {
                const bool la = op_a, lc = bnb;
                f32x4 qa[TQ], qc[TQ];
#pragma unroll
                for (int q = 0; q < TQ; ++q) {
                    bool valid;
                    const int off = chunk_off(t, u, q, valid);
                    const unsigned boff = valid ? (unsigned)off * 4u : OOB;
                    qa[q] = qc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (la) qa[q] = bload(rr, boff);
                    if (lc) qc[q] = bload(cr, boff);
                }
                if (la) to_acc_layout(qa, oa);
                if (lc) to_acc_layout(qc, oc);
            }
            float vo[RSL];
#pragma unroll
            for (int r = 0; r < RSL; ++r) {
                bool valid = (m0 + wm * TM * 32 + t * MR + MS::row(r_lo + r, e_lh)) < Mp && cvalid;
                unsigned boff = OOB;
                int off = 0;
                if (need_off) {
                    off = elem_off(t, u, r_lo + r, valid);
                    boff = valid ? (unsigned)off * 4u : OOB;
                }
                float v = acc[t][u][r] + bv;
                // transposed gather (input gradients): an addend, e.g. the other gradient of a tensor with two
                // consumers, rides in the epilogue instead of a separate add kernel (slice 0 only under SPLIT)
                if (addend) v += t128 ? oa[r] : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, boff, 0, 0));
                if (SPLIT) {
                    if (valid) atomicAdd(p.y + off, v);
                    continue;
                }
                if (EPI && MODE == 0) {
                    if (valid) { s1 += v; s2 += v * v; }
                    if (bn) v = (v - bn_m) * bn_is * bn_g + bn_b;
                    if (p.res) v += t128 ? oa[r] : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, boff, 0, 0));
                    v = act_fwd(v, p.act);
                }
                if (bnb) {                                  // out-of-tile lanes load 0 and contribute 0
                    const float cv = t128 ? oc[r] : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cr, boff, 0, 0));
                    const float xh = (cv - bb_mu) * bb_is;
                    if (mask_on) {                          // (wave-uniform: every lane takes part in the permute)
                        const int src = MS::row(r_lo + r, e_lh) + ((e_l31 & 16) << 1);
                        const unsigned wv = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)mw[t][u]);
                        v = ((wv >> mshift) & 1u) ? v : v * bb_slope;
                    } else if (recompute) {
                        v = __builtin_fmaf(xh, bb_g, bb_b) > 0.f ? v : v * bb_slope;
                    }
                    if (valid) { s1 += v; s2 += v * xh; }
                }
                vo[r] = v;
                if (!t128) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, boff, 0, CD_STORE_AUX);
            }
            if (!SPLIT && t128) {                           // the tile leaves as RSL / 4 sixteen-byte stores per lane
#pragma unroll
                for (int r = 0; r < RSL; ++r) Tx[MS::row(r, lh) * TP + l31] = vo[r];
                wave_fence();
#pragma unroll
                for (int q = 0; q < TQ; ++q) {
                    bool valid;
                    const int off = chunk_off(t, u, q, valid);
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(&Tx[(8 * q + (lane >> 3)) * TP + ((lane & 7) << 2)]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, w4), yr, valid ? (unsigned)off * 4u : OOB, 0,
                                                           CD_STORE_AUX);
                }
                wave_fence();
            }
        }
        if (stats) {                                        // wave-uniform branch: every lane shuffles
#pragma unroll
            for (int o = MR; o < 64; o <<= 1) {             // lanes of the same column differ in the k-lane bits
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            if (lh == 0) {
                const int cb = wn * TN * 32 + u * MR + l31;
                sred[(wk * WM + wm) * BN + cb] = s1;
                sred[((WK + wk) * WM + wm) * BN + cb] = s2;
            }
        }
    }
    if (stats) {
        __syncthreads();
        if (tid < BN && n0 + tid < p.Co) {
            double d1 = 0.0, d2 = 0.0;
#pragma unroll
            for (int k = 0; k < WK * WM; ++k) {
                d1 += (double)sred[k * BN + tid];
                d2 += (double)sred[(WK * WM + k) * BN + tid];
            }
            // the per-workgroup sums are folded onto stats_nbg slots per column (768 workgroups on the
            // dominant shape -> 12 atomics per address), which norm_finalize reduces and re-zeroes
            if (p.stats_tiles) {                            // ordered later by advmix_stats_fold: bit-reproducible
                const int64_t pb = (int64_t)bz * gridDim.x + bx;
                p.stats[(int64_t)(n0 + tid) * p.stats_nbg + pb] = d1;
                p.stats[((int64_t)p.Co + n0 + tid) * p.stats_nbg + pb] = d2;
            } else {
                // slot-major, [2][slots][Co]: the BN threads' atomics are consecutive doubles - a few cache-line requests per
                // workgroup where the channel-major layout of rounds 2-3 ([2][Co][slots]: one line per thread, and every
                // channel's slots in ONE run of lines) made 2 BN of them (round 4: -0.5 ... -1.5 us per launch, -5 at 1x1 64->256)
                const int pb = bx % p.stats_nbg;
                atomicAdd(p.stats + (int64_t)pb * p.Co + n0 + tid, d1);
                atomicAdd(p.stats + ((int64_t)p.stats_nbg + pb) * p.Co + n0 + tid, d2);
            }
        }
    }
}

template <int TM, int TN, int WM, int WN, int KC, int MODE, bool SPLIT, bool BT, bool EPI, int NW = 4>
// (left free, the register allocator spreads the epilogue operands over 254 VGPRs = one wave per SIMD.)  Four waves per SIMD
// (4-wave workgroups: 4 per CU, 8-wave: 2) for EVERY variant: a kernel's waves share the SIMDs with the other lanes' kernels,
// and the cap was worth 1.8 % of the step for the BatchNorm-backward variants alone (profiles/r03_ab_launch_bounds.log).
// Round 3 withdrew it from those variants while hunting the two-rank failure, which turned out to be the NULL-stream graph
// replay; with the y operand gone (round 4: activation mask) they fit 128 registers without scratch.
__global__ __launch_bounds__(64 * NW, NW == 4 ? 4 : 2) void conv_direct(ConvD p) {
    using G = Geo<TM, TN, WM, WN, KC, NW>;
    // K split inside the workgroup (WK > 1): the epilogue starts with a workgroup barrier (the partial tiles meet in the
    // FIRST weight buffer), so the wave-private transposer can live in the SECOND weight buffer instead of LDS of its
    // own - 42.5 -> 37.9 KB for the 32x32 tile, i.e. four workgroups per CU instead of three.
    constexpr bool ALIAS = G::WK > 1 && G::RED <= G::BSZ && G::TSZ <= G::BSZ;   // (the partial tiles must fit the first buffer)
    __shared__ __attribute__((aligned(16))) float Bs[2 * G::BSZ];
    __shared__ __attribute__((aligned(16))) float Ts[ALIAS ? 4 : G::TSZ];
    __shared__ int4 taptab[64];
    conv_body<TM, TN, WM, WN, KC, MODE, SPLIT, BT, EPI, NW>(p, blockIdx.x, blockIdx.y, blockIdx.z, Bs,
                                                            ALIAS ? Bs + G::BSZ : Ts, taptab);
}

// Tile configuration for a problem (the only place that decides it; advmix_conv_direct_config reports it).
enum Cfg { CFG_128x32 = 1, CFG_128x64 = 2, CFG_64x64 = 3, CFG_64x64_GRID_SPLIT = 4, CFG_32x32_WAVE_SPLIT = 5,
           CFG_64x32_WAVE_SPLIT2 = 6, CFG_64x32_K4_W8 = 7 };

// Several problems of one kind (same MODE / KC / BT / EPI, stride 1, whole-K tiles) in ONE launch: block b belongs to
// problem i with start[i] <= b < start[i + 1] and computes that problem's tile (b' % gx, b' / gx) with the problem's own
// tile configuration.  HRNet's branches run the same layer at 2-4 resolutions: one launch of 1,000-1,800 workgroups
// instead of four of 380-770 on four streams (kernel boundaries and ramps paid once; long and short tiles interleave).
struct ConvG {
    int n;
    int start[5];
    int cfg[4];
    int gx[4];
    ConvD p[4];
};

template <int A, int B> struct Max2 { static constexpr int v = A > B ? A : B; };

template <int MODE, int KC, bool BT, bool EPI>
// (the BatchNorm-backward group holds the largest configuration's registers: two waves per SIMD is what fits)
__global__ __launch_bounds__(256, (EPI && MODE == 1) ? 2 : 1) void conv_group(ConvG g) {
    constexpr int BSZ = Max2<Max2<Max2<Geo<1, 1, 4, 1, KC>::BSZ, Geo<1, 2, 4, 1, KC>::BSZ>::v,
                                  Max2<Geo<1, 1, 2, 2, KC>::BSZ, Geo<1, 1, 1, 1, KC>::BSZ>::v>::v,
                             Geo<1, 1, 2, 1, KC>::BSZ>::v;
    __shared__ __attribute__((aligned(16))) float Bs[2 * BSZ];
    __shared__ __attribute__((aligned(16))) float Ts[Geo<1, 1, 4, 1, KC>::TSZ];     // (WK = 1: the largest)
    __shared__ int4 taptab[64];
    const int b = blockIdx.x;
    int i = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < g.n && b >= g.start[k]) i = k;
    const int local = b - g.start[i], gx = g.gx[i];
    const int by = local / gx, bx = local - by * gx;
    switch (g.cfg[i]) {
        case CFG_128x32: conv_body<1, 1, 4, 1, KC, MODE, false, BT, EPI>(g.p[i], bx, by, 0, Bs, Ts, taptab); break;
        case CFG_128x64: conv_body<1, 2, 4, 1, KC, MODE, false, BT, EPI>(g.p[i], bx, by, 0, Bs, Ts, taptab); break;
        case CFG_64x64: conv_body<1, 1, 2, 2, KC, MODE, false, BT, EPI>(g.p[i], bx, by, 0, Bs, Ts, taptab); break;
        case CFG_32x32_WAVE_SPLIT: conv_body<1, 1, 1, 1, KC, MODE, false, BT, EPI>(g.p[i], bx, by, 0, Bs, Ts, taptab); break;
        case CFG_64x32_WAVE_SPLIT2: conv_body<1, 1, 2, 1, KC, MODE, false, BT, EPI>(g.p[i], bx, by, 0, Bs, Ts, taptab); break;
        default: break;
    }
}

// share of the chip's workgroup slots n workgroups fill when every CU takes ceil(n / 256) of them
static double fill(int64_t n) { return n <= 0 ? 0.0 : (double)n / (256.0 * (double)((n + 255) / 256)); }

static Cfg pick_cfg(int64_t Mmax, int Co, int phases, int nch, int* nsplit) {
    *nsplit = 1;
    if (Co <= 32) {                                        // (256 x 32 measured: 34.9 vs 27.5 us)
        static const int force = [] { const char* e = getenv("ADVMIX_CFG32"); return e ? atoi(e) : 0; }();
        if (force == 6 && nch >= 4) return CFG_64x32_WAVE_SPLIT2;
        if (force == 5 && nch >= 8) return CFG_32x32_WAVE_SPLIT;
        return CFG_128x32;
    }
    static const int force_all = [] { const char* e = getenv("ADVMIX_CFG"); return e ? atoi(e) : 0; }();   // measurement aid
    if (force_all == 2 || force_all == 3 || (force_all == 5 && nch >= 8) || (force_all == 6 && nch >= 4) || (force_all == 7 && nch >= 8)) return (Cfg)force_all;
    if ((int64_t)cdiv(Mmax, 128) * cdiv(Co, 64) * phases >= 512) return CFG_128x64;
    const int64_t b64 = (int64_t)cdiv(Mmax, 64) * cdiv(Co, 64) * phases;
    int ns = 1;
    if (b64 < 256 && nch >= 8) {                           // too few workgroups: split K
        ns = (int)((512 + b64 - 1) / b64);
        if (ns > nch / 4) ns = nch / 4;
        if (ns > 8) ns = 8;
        if (ns < 1) ns = 1;
    }
    if (ns <= 1) {
        // Between one and two 64 x 64 workgroups per CU the busiest CUs carry twice the work of the others (3x3 64->64
        // @32x24, B = 32: 384 workgroups on 256 CUs, 31.3 us = 0.37 of peak where B = 64 reaches 0.51).  64 x 32 tiles
        // whose two wave pairs split K double the workgroup count (768 = 3 per CU) and keep the fused epilogue.
        const int64_t b6432 = (int64_t)cdiv(Mmax, 64) * cdiv(Co, 32) * phases;
        if (advmix_opts().ksplit_wg && nch >= 4 && fill(b6432) > 1.15 * fill(b64)) return CFG_64x32_WAVE_SPLIT2;
        return CFG_64x64;
    }
    // Enough 32 x 32 tiles to give every CU a workgroup: split K between the four waves of a workgroup (no
    // zero-fill, no atomics, fused epilogue kept).  3x3 128->128 @16x12: 29.1 vs 35.8 us; 256->256 @8x6: 37.0 vs
    // 34.8 us but the separate statistics / BN pass and the memset disappear; with fewer tiles (U-Net
    // bottleneck, 4x4 512->512 @4x3: 84 vs 59 us) the grid-level split wins.
    const int64_t b32 = (int64_t)cdiv(Mmax, 32) * cdiv(Co, 32) * phases;
    // Between one and two 32 x 32 workgroups per CU (3x3 256->256 @8x6, B = 32: 384) the layer is bound by its weights -
    // 2.36 MB re-staged by each of 48 row tiles: eight-wave workgroups whose two row tiles share ONE staging of every weight
    // chunk (192 workgroups, the same 1,536 waves): forward + sums 36.2 -> 34.2 us, input gradient + BatchNorm backward
    // 40.0 -> 36.6 (profiles/r03_exp_tiles_8waves.log); with 768 tiles (128->128 @16x12) the four-wave form stays ahead.
    if (advmix_opts().ksplit_wg && b32 >= 256 && b32 < 512 && nch >= 8 && Mmax >= 64) return CFG_64x32_K4_W8;
    if (advmix_opts().ksplit_wg && b32 >= 256) return CFG_32x32_WAVE_SPLIT;
    if (advmix_opts().deterministic)                       // the grid split adds with fp32 atomics: order varies run to run
        return advmix_opts().ksplit_wg ? CFG_32x32_WAVE_SPLIT : CFG_64x64;
    *nsplit = ns;
    return CFG_64x64_GRID_SPLIT;
}

static void problem_shape(int mode, int Ci, int R, int S, int stride, int KC, int* phases, int* nch) {
    *phases = mode == 0 ? 1 : stride * stride;
    const int maxtaps = mode == 0 ? R * S : ((R + stride - 1) / stride) * ((S + stride - 1) / stride);
    *nch = maxtaps * (Ci / KC);
}

template <int MODE, int KC, bool BT, bool EPI>
int launch(ConvD& p, int64_t Mmax, hipStream_t st, int64_t stats_cap) {
    static_assert(!EPI || (MODE == 0 && !BT) || (MODE == 1 && BT), "fused epilogues: forward gather, or BN-backward on the k-major transposed gather");
    int phases, nch, ns;
    problem_shape(MODE, p.Ci, p.R, p.S, p.stride, KC, &phases, &nch);
#define LAUNCHD(TM_, TN_, WM_, WN_, SP_) LAUNCHW(TM_, TN_, WM_, WN_, SP_, 4)
#define LAUNCHW(TM_, TN_, WM_, WN_, SP_, NW_)                                                     \
    do {                                                                                          \
        dim3 g(cdiv(Mmax, 32 * TM_ * WM_), cdiv(p.Co, 32 * TN_ * WN_), phases * p.nsplit);        \
        if (p.stats_tiles) {                               /* one slot per (phase, row tile); capacity checked here */ \
            if ((int64_t)g.x * g.z > stats_cap) return -2;                                        \
            p.stats_nbg = (int)(g.x * g.z);                                                       \
        }                                                                                         \
        hipLaunchKernelGGL((conv_direct<TM_, TN_, WM_, WN_, KC, MODE, SP_, BT, (EPI && !SP_), NW_>), g, dim3(64 * NW_), 0, st, p); \
        if (advmix_opts().trace_shapes) {                                                         \
            char nm[96];                                                                          \
            snprintf(nm, sizeof nm, "conv_direct<%d, %d, %d, %d, %d, %d, %s, %s, %s, %d>", TM_, TN_, WM_, WN_, KC, MODE, \
                     SP_ ? "true" : "false", BT ? "true" : "false", (EPI && !SP_) ? "true" : "false", NW_); \
            advmix_trace_launch(nm, g, MODE == 0 ? (p.stats ? "fwd+sums" : (p.bn_gamma ? "fwd+bn_eval" : "fwd")) \
                                                 : (p.bnb_c ? "dgrad+bnb" : "dgrad"),              \
                                p.N, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co, p.R, p.S, p.stride,      \
                                2.0 * p.N * (MODE == 0 ? (double)p.Ho * p.Wo : (double)p.Hi * p.Wi) * p.Co * p.Ci * p.R * p.S); \
        }                                                                                         \
    } while (0)
    p.nsplit = 1;
    switch (pick_cfg(Mmax, p.Co, phases, nch, &ns)) {
        case CFG_128x32: LAUNCHD(1, 1, 4, 1, false); break;
        case CFG_128x64: {                                 // two tiles per wave: no registers to prefetch the BatchNorm-
            // backward operands; they are loaded in the epilogue, 16 bytes per lane and access (round 3; as scalar loads
            // they lost to the separate statistics pass).  ADVMIX_BNB128=0: A/B switch back to that pass.
            static const int bnb128 = [] { const char* e = getenv("ADVMIX_BNB128"); return e ? atoi(e) : 1; }();
            if (p.bnb_c && !bnb128) return -2;
            LAUNCHD(1, 2, 4, 1, false);
            break;
        }
        case CFG_64x64: LAUNCHD(1, 1, 2, 2, false); break;
        case CFG_32x32_WAVE_SPLIT: LAUNCHD(1, 1, 1, 1, false); break;      // four waves share K in the workgroup
        case CFG_64x32_WAVE_SPLIT2: LAUNCHD(1, 1, 2, 1, false); break;     // two wave pairs share K
        case CFG_64x32_K4_W8: LAUNCHW(1, 1, 2, 1, false, 8); break;       // eight waves: two row tiles x four-way K split, weights staged once
        case CFG_64x64_GRID_SPLIT:                                         // K across gridDim.z + atomics
            if (p.bn_gamma || (MODE == 0 && p.res) || p.act || p.stats || p.bnb_c) return -2;   // fused epilogue needs whole-K tiles
            p.nsplit = ns;
            if (hipMemsetAsync(p.y, 0, (size_t)p.N * p.Ho * p.Wo * p.Co * sizeof(float), st) != hipSuccess)
                return ADVMIX_ELAUNCH;
            LAUNCHD(1, 1, 2, 2, true);
            break;
    }
#undef LAUNCHD
#undef LAUNCHW
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

}  // namespace direct

// Checks a problem and fills its descriptor: 0, or -1 when the problem is not eligible (caller falls back to conv_igemm),
// -2 when only the fused epilogue is not available.  ``*bnb``: the epilogue is the BatchNorm-backward one.
static int prepare(int mode, const float* x, const float* w, const float* bias, float* y,
                   int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S,
                   int stride, int pad, int64_t Mmax, int bt, const ConvEpi* epi, const int* stats_nbg,
                   direct::ConvD* out, bool* bnb_out) {
    if (Ci % 16 != 0 || R * S > 64) return -1;
    if (bt && (mode != 1 || Co % 4 != 0)) return -1;
    if (epi && mode != 0 && (epi->gamma || epi->act)) return -2;                 // mode 1: addend and/or BN-backward sums
    if (epi && mode != 0 && epi->stats && !(bt && epi->bnb_c && epi->bnb_mean && epi->bnb_invstd &&
                                            (epi->bnb_act == ADVMIX_ACT_NONE || (epi->bnb_mask && Co % 16 == 0) ||
                                             (!epi->bnb_mask && epi->bnb_gamma && epi->bnb_beta)))) return -2;
    const int64_t xb = (int64_t)N * Hi * Wi * Ci * 4, wb = (int64_t)Co * R * S * Ci * 4;
    const int64_t yb = (int64_t)N * Ho * Wo * Co * 4;
    if (xb >= 0x7fffffffLL || wb >= 0x7fffffffLL || yb >= 0x7fffffffLL) return -1;
    direct::ConvD p{x, w, bias, y, N, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride, pad, (int)xb, (int)wb, (int)yb, 1,
                    nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0, nullptr, 0,
                    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    static const int xcd_remap = [] { const char* e = getenv("ADVMIX_XCD_REMAP"); return e ? atoi(e) : 1; }();
    p.xcd_remap = xcd_remap;
    // Slots per channel the workgroup sums are folded onto: many row blocks hammering few addresses serialise the
    // fp64 atomics at the memory side, while every consumer workgroup has to reduce all of them again.  (Channel-major
    // slots, rounds 2-3: 3x3 32->32 @64x48, 768 workgroups, 30.8 us with 16 slots, 23.1 with 64.  Slot-major - a workgroup's
    // atomics are a few whole cache lines - the count stopped mattering below 32.)
    // *stats_nbg < 0 (deterministic mode): -capacity - one slot per row tile, plain stores (see ConvD::stats_tiles)
    const bool tiles = epi && epi->stats && stats_nbg && *stats_nbg < 0;
    int ns = stats_nbg && *stats_nbg > 0 ? *stats_nbg : advmix_opts().stat_slots;
    if (ns <= 0) ns = 16;      // (slot-major layout, round 4: 16 everywhere 606.9 / 606.6 images/s, 8: 606.8 / 607.1, 4: 605.7 / 606.4, the
                               //  channel-major rule - 64 / 32 for the large grids, 16 for the small - 604.9 / 605.3, 64: 598.0 / 599.8)
    if (ns > 64 || (ns & (ns - 1))) ns = 16;
    p.stats_nbg = ns;
    p.stats_tiles = tiles ? 1 : 0;
    bool bnb = false;
    if (epi) {
        p.bn_gamma = epi->gamma; p.bn_beta = epi->beta; p.bn_rm = epi->rm; p.bn_rv = epi->rv; p.res = epi->res;
        p.bn_eps = epi->eps; p.act = epi->act; p.stats = epi->stats;
        if (mode != 0 && epi->stats) {
            bnb = true;
            p.bnb_mask = epi->bnb_mask; p.bnb_c = epi->bnb_c; p.bnb_mean = epi->bnb_mean; p.bnb_invstd = epi->bnb_invstd;
            p.bnb_gamma = epi->bnb_gamma; p.bnb_beta = epi->bnb_beta;
            p.bnb_act = epi->bnb_act;
        }
    }
    *out = p;
    *bnb_out = bnb;
    return 0;
}

// bt != 0 (mode 1 only): w is k-major [Ci(k)][R][S][Co(n)] instead of [Co][R][S][Ci].
int advmix_conv_direct_dispatch(int mode, const float* x, const float* w, const float* bias, float* y,
                                int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S,
                                int stride, int pad, int64_t Mmax, hipStream_t st, int bt, const ConvEpi* epi,
                                int* stats_nbg) {
    direct::ConvD p;
    bool bnb = false;
    const int64_t stats_cap = stats_nbg && *stats_nbg < 0 ? -(int64_t)*stats_nbg : 0;
    int rc = prepare(mode, x, w, bias, y, N, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride, pad, Mmax, bt, epi, stats_nbg, &p, &bnb);
    if (rc < 0) return rc;
#define LAUNCH_KC(MODE_, BT_, EPI_)                                                             \
    (Ci % 32 == 0 ? direct::launch<MODE_, 32, BT_, EPI_>(p, Mmax, st, stats_cap) : direct::launch<MODE_, 16, BT_, EPI_>(p, Mmax, st, stats_cap))
    if (bt && bnb)
        rc = LAUNCH_KC(1, true, true);
    else if (bt)
        rc = LAUNCH_KC(1, true, false);
    else if (mode == 1)
        rc = LAUNCH_KC(1, false, false);
    else if (epi)
        rc = LAUNCH_KC(0, false, true);
    else
        rc = LAUNCH_KC(0, false, false);
#undef LAUNCH_KC
    if (stats_nbg) *stats_nbg = p.stats_nbg;
    return rc;
}

// conv_wino4.hip's batched GEMM: c[b] = a[b] . w[b]^T for b < nb, a[b] = a + b * rows * K ([rows][K]), w[b] = w + b * Nc * K ([Nc][K]),
// c[b] = c + b * rows * Nc - a 1x1 convolution over nb images of ``rows`` pixels in which every image has its own filters
// (ConvD::wimg).  rows % 128 == 0 (no row tile straddles two images), K % 32 == 0, Nc % 4 == 0.  -1: not served.  a == NULL: dry run
// (0 = this shape would be served; nothing is launched).
int advmix_conv_direct_gemm_batched(const float* a, const float* w, float* c, int nb, int rows, int K, int Nc, hipStream_t st) {
    if (nb <= 0 || rows <= 0 || rows % 128 != 0 || K % 32 != 0 || Nc % 4 != 0) return -1;
    direct::ConvD p;
    bool bnb = false;
    const int64_t Mmax = (int64_t)nb * rows;
    int rc = prepare(0, a, w, nullptr, c, nb, rows / 128, 128, K, rows / 128, 128, Nc, 1, 1, 1, 0, Mmax, 0, nullptr, nullptr, &p, &bnb);
    if (rc < 0) return -1;
    if (!a) return 0;                                       // a == NULL: a dry run - "would be served", nothing launched (ADVICE r5:
    p.wimg = Nc * K;                                        // conv_wino4's entry points ask BEFORE they launch their input transform)
    return direct::launch<0, 32, false, false>(p, Mmax, st, 0);
}

// 2-4 problems in one launch (direct::conv_group).  -1: the group cannot be served as one launch (different kernel
// kinds, a stride, a problem that needs the grid K split or the first-generation kernel, no room for the BatchNorm-
// backward operands): nothing was launched.  Forward problems (mode 0) all carry an epilogue descriptor or none does;
// transposed-gather problems (mode 1, weights in their own layout) are all BatchNorm-backward ones or none is.
int advmix_conv_direct_group(int mode, int bt, int n, ConvProb* pr, hipStream_t st) {
    if (n < 2 || n > 4) return -1;
    direct::ConvG g;
    g.n = n;
    bool bnb0 = false, epi0 = pr[0].epi != nullptr;
    const int KC = pr[0].Ci % 32 == 0 ? 32 : 16;
    int order[4] = {0, 1, 2, 3};
    int work[4], gy[4];
    for (int i = 0; i < n; ++i) {
        ConvProb& q = pr[i];
        if (q.stride != 1 || (q.Ci % 32 == 0 ? 32 : 16) != KC || (mode == 0 && (q.epi != nullptr) != epi0)) return -1;
        bool bnb = false;
        direct::ConvD d;
        if (prepare(mode, q.x, q.w, q.bias, q.y, q.N, q.Hi, q.Wi, q.Ci, q.Ho, q.Wo, q.Co, q.R, q.S, q.stride, q.pad, q.Mmax,
                    bt, q.epi, &q.stats_nbg, &d, &bnb) < 0)
            return -1;
        if (i == 0) bnb0 = bnb;
        if (bnb != bnb0) return -1;
        int phases, nch, ns;
        direct::problem_shape(mode, q.Ci, q.R, q.S, q.stride, KC, &phases, &nch);
        direct::Cfg c = direct::pick_cfg(q.Mmax, q.Co, phases, nch, &ns);
        if (c == direct::CFG_64x32_K4_W8) c = direct::CFG_32x32_WAVE_SPLIT;   // (a group's workgroups have four waves: the same
                                                                               //  K split per row tile, bit-identical outputs)
        if (phases != 1 || c == direct::CFG_64x64_GRID_SPLIT) return -1;
        int bm, bn, wk;
        switch (c) {
            case direct::CFG_128x32: bm = 128; bn = 32; wk = 1; break;
            case direct::CFG_128x64: bm = 128; bn = 64; wk = 1; break;
            case direct::CFG_64x64: bm = 64; bn = 64; wk = 1; break;
            case direct::CFG_32x32_WAVE_SPLIT: bm = 32; bn = 32; wk = 4; break;
            default: bm = 64; bn = 32; wk = 2; break;      // CFG_64x32_WAVE_SPLIT2
        }
        d.nsplit = 1;
        if (d.stats_tiles) return -1;                      // (deterministic per-tile slots: single launches only)
        d.xcd_remap = 0;                                  // (block ranges per problem: the launch-order argument does not hold)
        g.p[i] = d;
        g.cfg[i] = (int)c;
        g.gx[i] = cdiv(q.Mmax, bm);
        gy[i] = cdiv(q.Co, bn);
        work[i] = nch * (c == direct::CFG_128x64 ? 2 : 1) / wk;     // MFMA chunks per wave of a tile: longest first
        q.stats_nbg = d.stats_nbg;
    }
    for (int a = 0; a < n; ++a)                            // order: longest-running tiles first
        for (int b = a + 1; b < n; ++b)
            if (work[order[b]] > work[order[a]]) { int t = order[a]; order[a] = order[b]; order[b] = t; }
    direct::ConvG h = g;
    int start = 0;
    for (int k = 0; k < n; ++k) {
        const int i = order[k];
        h.p[k] = g.p[i]; h.cfg[k] = g.cfg[i]; h.gx[k] = g.gx[i];
        h.start[k] = start;
        start += g.gx[i] * gy[i];
    }
    for (int k = n; k < 5; ++k) h.start[k] = start;
    dim3 grid(start);
#define LAUNCHG(MODE_, BT_, EPI_)                                                                         \
    do {                                                                                                  \
        if (KC == 32) hipLaunchKernelGGL((direct::conv_group<MODE_, 32, BT_, EPI_>), grid, dim3(256), 0, st, h); \
        else hipLaunchKernelGGL((direct::conv_group<MODE_, 16, BT_, EPI_>), grid, dim3(256), 0, st, h);   \
    } while (0)
    if (mode == 0 && !bt) {
        if (epi0) LAUNCHG(0, false, true); else LAUNCHG(0, false, false);
    } else if (mode == 1 && bt) {
        if (bnb0) LAUNCHG(1, true, true); else LAUNCHG(1, true, false);
    } else {
        return -1;
    }
#undef LAUNCHG
    if (advmix_opts().trace_shapes) {
        double fl = 0;
        for (int i = 0; i < n; ++i)
            fl += 2.0 * pr[i].N * (mode == 0 ? (double)pr[i].Ho * pr[i].Wo : (double)pr[i].Hi * pr[i].Wi) * pr[i].Co * pr[i].Ci *
                  pr[i].R * pr[i].S;
        char nm[64];
        snprintf(nm, sizeof nm, "conv_group<%d, %d, %s, %s>", mode, KC, bt ? "true" : "false",
                 (mode == 0 ? epi0 : bnb0) ? "true" : "false");
        advmix_trace_launch(nm, grid, mode == 0 ? "fwd group" : "dgrad group", pr[0].N, pr[0].Hi, pr[0].Wi, pr[0].Ci, pr[0].Ho,
                            pr[0].Wo, pr[0].Co, pr[0].R, pr[0].S, 1, fl);
    }
    return hipGetLastError() == hipSuccess ? ADVMIX_OK : ADVMIX_ELAUNCH;
}

// Bit mask of the measurement switches this library was compiled with - THE place a measurement build is detected: 0 for the
// shipped library (__graft_entry__.build(), advmix_amd/_lib.py and bench.py check it).  1 = CD_DBG (parts of the kernel
// compiled out), 2 = CD_PRELOAD, 4 = CD_CLK, 8 = CD_NO_PRE - all four exist only in tools/variants/conv_direct_dbg.patch -
// 16 = a store cache policy other than sc1, 32 = WL_DBG (wgrad_lds.hip, tools/variants/wgrad_lds_dbg.patch).
extern "C" int advmix_build_flags(void) {
    int f = 0;
#ifdef CD_DBG
    f |= 1;
#endif
#ifdef CD_PRELOAD
    f |= 2;
#endif
#ifdef CD_CLK
    f |= 4;
#endif
#ifdef CD_NO_PRE
    f |= 8;
#endif
    if (CD_STORE_AUX != 16) f |= 16;
    f |= advmix_wgrad_lds_build_flags();
    return f;
}

// Which conv_direct tile configuration a problem gets (tests assert that the shapes meant to exercise a
// kernel variant really reach it): 1 = 128x32, 2 = 128x64, 3 = 64x64, 4 = 64x64 + grid K split,
// 5 = 32x32 + K split between the four waves of a workgroup, 6 = 64x32 + K split between two wave pairs;
// -1 = not served by conv_direct.
// mode 0: forward (Mmax = N*Ho*Wo output pixels); mode 1: transposed gather (per-phase rows of the LARGER side).
extern "C" int advmix_conv_direct_config(int mode, int N, int Ho, int Wo, int Ci, int Co, int R, int S, int stride) {
    if (Ci % 16 != 0 || R * S > 64 || N <= 0 || stride < 1) return -1;
    const int KC = Ci % 32 == 0 ? 32 : 16;
    int phases, nch, ns;
    direct::problem_shape(mode, Ci, R, S, stride, KC, &phases, &nch);
    const int64_t Mmax = mode == 0 ? (int64_t)N * Ho * Wo : (int64_t)N * cdiv(Ho, stride) * cdiv(Wo, stride);
    return (int)direct::pick_cfg(Mmax, Co, phases, nch, &ns);
}
