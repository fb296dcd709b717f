// 3x3 / stride 1 / pad 1 convolution of SMALL maps with 256 input channels - one workgroup per image (round 5).
//
// HRNet's lowest-resolution branch (pose_hrnet.py:22-57 at 256x192: 256 -> 256 @8x6, 24 convs per pass) is the slowest
// member of stage 4 on every kernel so far: 1,536 pixels x 256 channels are 384 output tiles of 32 x 32 - 192 eight-wave
// workgroups of conv_direct on 256 CUs, two K-sharing waves per SIMD, each row tile re-staging all 2.36 MB of filters
// through LDS: 33 us for 11.5 us of MFMA work (0.35 of peak), while the other branches' convs take 14-16.  A Winograd form
// needs a 32-tile block to span 2 2/3 images and streams 16/9 as many filter bytes.  Here instead:
//   * a workgroup = ONE image (48 pixels) x 32 output channels, eight waves: grid 8 column tiles x 32 images = 256
//     workgroups, one per CU, every SIMD two waves with exactly 432 v_mfma_f32_16x16x4_f32 each - the balanced split of
//     the 442 k MFMAs of the layer (11.5 us at peak);
//   * the image with its zero halo ((H + 2) x (W + 2) pixels x 256 channels = 83 KB) is staged in LDS ONCE; all nine taps
//     are 16-byte LDS reads of it (pixel pitch 260 floats: a 16-lane group falls on 16 different slots);
//   * wave w multiplies channels [32 w, 32 w + 32) of every tap - K is split over the waves, not over workgroups - with
//     filters pre-laid in MFMA B-fragment order by smap_weights (one launch per forward pass for all such convs, like the
//     Winograd images): a wave's 36 loads are 36 consecutive KB, four iterations in flight, never staged in LDS.  XCD k
//     works on two column tiles and every second image: filters and images are each fetched by a few of the 8 L2s only;
//   * the eight partial tiles meet in LDS ([wave][pixel][36] floats); thread (pixel, 4 channels) adds them with 16-byte
//     reads and runs the fused epilogues of conv_direct / conv_wino in the natural layout: BatchNorm column sums (fp64
//     slots), eval-mode BatchNorm + residual + activation, or - input-gradient role - addend, activation slope from the
//     bit mask / from c, BatchNorm-backward sums; 16-byte loads and stores, no transposer.
#include "common.h"
#include <stdio.h>

namespace smap {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;      // >= any buffer size accepted -> loads return 0
constexpr int STORE_AUX = 16;              // sc1 (write-through), as conv_direct's epilogue
constexpr int C = 256;                     // input channels (= 8 waves x 32)
constexpr int NT = 512;
constexpr int PP = C + 4;                  // floats per staged pixel
constexpr int MAXPOS = 80, MAXPX = 48;     // (H + 2) (W + 2) <= 80 staged pixels, H W <= 48 output pixels (3 MFMA row tiles)
constexpr int RP = 36;                     // pitch of a pixel row in the reduction image (floats)
constexpr int NIT = 18;                    // 9 taps x 2 groups of 16 channels per wave
constexpr int RING = 4;                    // iterations of filter loads in flight

struct SP {
    const float* x;
    const float* u;           // filters in fragment order (see smap_weights)
    float* y;
    int N, H, W, Co;          // 3x3, stride 1, pad 1: input (256 channels) and output are both H x W
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
};

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}

// Workgroups are dealt to the 8 XCDs (and their 8 non-coherent L2s) round-robin in launch order.  With eight column tiles
// XCD k = (column-tile pair k % 4, image parity k / 4): an image is fetched by 4 L2s and a filter slice by 2 (6.3 + 4.7 MB
// from HBM for 256 -> 256 @8x6 at B = 32, against 12.6 + 2.4 with XCD = column tile and 1.6 + 18.9 with XCD = image group).
__device__ __forceinline__ void block_map(int& nt, int& img) {
    nt = blockIdx.x; img = blockIdx.y;
    if (gridDim.x == 8 && (gridDim.y & 1) == 0) {
        const int lin = (int)(blockIdx.y * 8 + blockIdx.x), xcd = lin & 7, j = lin >> 3;
        nt = 2 * (xcd & 3) + (j & 1);
        img = 2 * (j >> 1) + (xcd >> 2);
    }
}

// The epilogue of thread (pixel m = tid / 8, channels col .. col + 3, col = n0 + 4 (tid % 8)) - both kernels of this file end
// with their result in this natural layout.  Its read operands (residual / addend, the producer's c, its activation mask) are
// requested early (epi_request) and consumed after the workgroup's reduction (epi_finish).
struct EpiOps {
    f32x4 oa, oc;
    unsigned mbits, yo;
    int col, pix;
    bool live;
};

template <int ROLE>
__device__ __forceinline__ EpiOps epi_request(const SP& p, int tid, int img, int PX, int n0) {
    EpiOps o;
    const int m = tid >> 3, cq = tid & 7;
    o.live = m < PX;
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.y), 0, p.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bnb_c ? p.bnb_c : p.y), 0, p.ybytes, 0x00020000);
    o.col = n0 + 4 * cq;
    o.pix = img * PX + (o.live ? m : 0);
    o.yo = o.live ? (unsigned)((o.pix * p.Co + o.col) * 4) : OOB;
    const bool bnb = ROLE == 1 && p.bnb_c != nullptr;
    const bool mask_on = bnb && p.bnb_mask != nullptr && p.bnb_act != ADVMIX_ACT_NONE;
    o.oa = f32x4{0.f, 0.f, 0.f, 0.f};
    o.oc = o.oa;
    if (p.res != nullptr) o.oa = bload(rr, o.yo);
    if (ROLE == 1 && bnb) o.oc = bload(cr, o.yo);
    o.mbits = 0u;
    if (ROLE == 1 && mask_on && o.live) o.mbits = p.bnb_mask[o.yo >> 4];    // byte (pixel * Co + column) / 4; bit e: channel col + e
    return o;
}

// conv_direct.hip's epilogue arithmetic in the natural layout: BatchNorm column sums (fp64 slots), eval-mode BatchNorm +
// residual + activation, or - input-gradient role - addend, activation slope from the bit mask / from c, BatchNorm-backward sums.
template <int ROLE>
__device__ __forceinline__ void epi_finish(const SP& p, f32x4 v, const EpiOps& o, float* sred, int tid, int img, int n0) {
    const int lane = tid & 63, wv = tid >> 6;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.ybytes, 0x00020000);
    const bool stats = p.stats != nullptr;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const bool bnf = ROLE == 0 && p.bn_gamma != nullptr;
    const bool bnb = ROLE == 1 && p.bnb_c != nullptr;
    const bool mask_on = bnb && p.bnb_mask != nullptr && p.bnb_act != ADVMIX_ACT_NONE;
    const bool recompute = ROLE == 1 && bnb && !mask_on && p.bnb_act != ADVMIX_ACT_NONE;
    const float bb_slope = act_neg_slope(p.bnb_act);
    const int col = o.col;
    auto ld4 = [&](const float* q) { return *reinterpret_cast<const f32x4*>(q + col); };    // (col % 4 == 0: one 16-byte load)
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bn_is = z4, bn_g = z4, bn_b = z4, bn_m = z4, bb_mu = z4, bb_is = z4, bb_g = z4, bb_b = z4;
    if (bnf) {
        const f32x4 rv = ld4(p.bn_rv);
#pragma unroll
        for (int e = 0; e < 4; ++e) bn_is[e] = 1.0f / sqrtf(rv[e] + p.bn_eps);
        bn_g = ld4(p.bn_gamma); bn_b = ld4(p.bn_beta); bn_m = ld4(p.bn_rm);
    }
    if (ROLE == 1 && bnb) {
        bb_mu = ld4(p.bnb_mean); bb_is = ld4(p.bnb_invstd);
        if (recompute) { bb_g = ld4(p.bnb_gamma); bb_b = ld4(p.bnb_beta); }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e];
        if (ROLE == 0) {
            if (o.live) { s1[e] = x; s2[e] = x * x; }
            if (bnf) x = (x - bn_m[e]) * bn_is[e] * bn_g[e] + bn_b[e];
            x += o.oa[e];
            x = act_fwd(x, p.act);
        } else {
            x += o.oa[e];
            if (bnb) {
                const float xh = (o.oc[e] - bb_mu[e]) * bb_is[e];
                if (mask_on) x = ((o.mbits >> e) & 1u) ? x : x * bb_slope;
                else if (recompute) x = __builtin_fmaf(xh, bb_g[e], bb_b[e]) > 0.f ? x : x * bb_slope;
                if (o.live) { s1[e] = x; s2[e] = x * xh; }
            }
        }
        v[e] = x;
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, o.yo, 0, STORE_AUX);
    if (stats) {                                            // uniform over the grid
        // a wave = 8 pixels x 8 channel quads (lane = 8 (m % 8) + cq): the pixels add up by shuffles, the waves in LDS
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
                sred[(8 + wv) * 32 + 4 * lane + e] = s2[e];
            }
        }
        __syncthreads();
        if (tid < 32) {
            double d1 = 0.0, d2 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                d1 += (double)sred[k * 32 + tid];
                d2 += (double)sred[(8 + k) * 32 + tid];
            }
            const int sl = img % p.stats_nbg;              // slot-major [2][slots][Co]: consecutive doubles per workgroup
            atomicAdd(p.stats + (int64_t)sl * p.Co + n0 + tid, d1);
            atomicAdd(p.stats + ((int64_t)p.stats_nbg + sl) * p.Co + n0 + tid, d2);
        }
    }
}

// The padded image -> LDS (an out-of-image pixel is an out-of-range offset: the load returns 0).
__device__ __forceinline__ void stage_image(const SP& p, float* L, int tid, int img, int PW, int NPOS) {
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    constexpr int SIT = (MAXPOS * (C / 4) + NT - 1) / NT;
    f32x4 stg[SIT];
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
        const int s = tid + NT * it;
        const int pos = s >> 6, cs = s & 63;
        const int ph = pos / PW, pw = pos - ph * PW;
        const int h = ph - 1, w = pw - 1;
        const bool ok = pos < NPOS && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        stg[it] = bload(xr, ok ? (unsigned)((((img * p.H + h) * p.W + w) * C + cs * 4) * 4) : OOB);
    }
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
        const int s = tid + NT * it;
        const int pos = s >> 6, cs = s & 63;
        if (pos < NPOS) *reinterpret_cast<f32x4*>(&L[pos * PP + cs * 4]) = stg[it];
    }
}

template <int ROLE>
__global__ __launch_bounds__(NT, 1) void conv_smap(const SP p) {
    // one region, two lives (separated by workgroup barriers): the padded image, then the eight waves' partial tiles
    __shared__ __attribute__((aligned(16))) float L[MAXPOS * PP];
    __shared__ float sred[2 * 8 * 32];
    static_assert(8 * MAXPX * RP <= MAXPOS * PP, "the reduction image fits the patch region");

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;               // MFMA 16x16x4: row / column l15, k-lane q
    int nt, img;
    block_map(nt, img);
    const int n0 = nt * 32;
    const int PX = p.H * p.W, PW = p.W + 2, NPOS = (p.H + 2) * PW;

    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.u, 0, p.ubytes, 0x00020000);
    // this wave's filters: 36 consecutive fragments of 1 KB - iteration it = (tap, 16-channel group) needs 2 it and 2 it + 1
    const unsigned bo = (unsigned)((((nt * 8 + wv) * (2 * NIT)) * 64 + lane) * 16);
    f32x4 bq[RING][2];
    auto issue_b = [&](int it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) bq[it % RING][u] = bload(ur, bo + (unsigned)((2 * it + u) * 1024));
    };
#pragma unroll
    for (int it = 0; it < RING; ++it) issue_b(it);

    stage_image(p, L, tid, img, PW, NPOS);                  // ---- the image with its zero halo -> LDS ----

    // ---- this lane's three pixels (one per MFMA row tile; rows past the image repeat its last pixel: their results are
    //      never used - a row of A only reaches the same row of D) ---------------------------------------------------------
    const float* la[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        int m = 16 * t + l15;
        if (m >= PX) m = PX - 1;
        const int h = m / p.W, w = m - h * p.W;
        la[t] = L + (h * PW + w) * PP + 32 * wv + 4 * q;   // tap (0, 0) of the pixel's 3 x 3 window, this wave's channels
    }
    f32x4 acc[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();                                        // the image is complete
    auto toff = [&](int it) { const int tap = it >> 1; return ((tap / 3) * PW + (tap % 3)) * PP + 16 * (it & 1); };
    f32x4 an[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) an[t] = *reinterpret_cast<const f32x4*>(la[t] + toff(0));
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        f32x4 bc[2], a[3];
        __builtin_amdgcn_sched_barrier(0);                  // (nothing of iteration it + 1 is hoisted above this one's MFMAs)
#pragma unroll
        for (int u = 0; u < 2; ++u) bc[u] = bq[it % RING][u];
#pragma unroll
        for (int t = 0; t < 3; ++t) a[t] = an[t];
        if (it + RING < NIT) issue_b(it + RING);             // four iterations of filters in flight,
        if (it + 1 < NIT) {                                  // the next iteration's pixels read under this one's 24 MFMAs
#pragma unroll
            for (int t = 0; t < 3; ++t) an[t] = *reinterpret_cast<const f32x4*>(la[t] + toff(it + 1));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j], bc[u][j], acc[t][u], 0, 0, 0);
    }

    const EpiOps eo = epi_request<ROLE>(p, tid, img, PX, n0);   // (requested now, they arrive under the reduction)
    const int m = tid >> 3, cq = tid & 7;

    __syncthreads();                                        // every wave is done with the image: the region becomes the reduction image
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)                      // D[row 4 q + r][column l15]
                L[(wv * MAXPX + 16 * t + 4 * q + r) * RP + 16 * u + l15] = acc[t][u][r];
    __syncthreads();
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (m < MAXPX) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(&L[(k * MAXPX + m) * RP + 4 * cq]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += x[e];
        }
    }
    epi_finish<ROLE>(p, v, eo, sred, tid, img, n0);
}

// ---- the same workgroup shape with Winograd F(2x2, 3x3) inside ------------------------------------------------------------
// An 8 x 6 map is 4 x 3 = 12 output tiles of 2 x 2: one MFMA row tile (16 rows, 12 used) per position xi of the transformed
// 4 x 4 patch.  Wave w multiplies xi = 2 w and 2 w + 1 over ALL 256 channels (no K split, no partial tiles to add): 256
// v_mfma_f32_16x16x4_f32 per wave instead of 432.  The A operand is formed on the way from LDS to the MFMA - (B^T d B)[i][j]
// is a signed sum of FOUR pixels of the staged image (each row of B^T has two non-zeros) - so nothing transformed is ever
// stored; filters come as U = G g G^T in fragment order (smapw_weights: [column tile][wave][16-channel group][xi of the
// wave][column half][lane][4]: 64 consecutive KB per wave).  The sixteen 16 x 32 products meet in LDS; thread (pixel, 4
// channels) applies A^T . A (nine 16-byte reads) and runs the shared epilogue.
constexpr int WIT = 16;                    // 16-channel groups of the 256 channels
template <int ROLE>
__global__ __launch_bounds__(NT, 1) void conv_smapw(const SP p) {
    __shared__ __attribute__((aligned(16))) float L[MAXPOS * PP];
    __shared__ float sred[2 * 8 * 32];
    static_assert(16 * 16 * RP <= MAXPOS * PP, "the product image fits the patch region");

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    int nt, img;
    block_map(nt, img);
    const int n0 = nt * 32;
    const int PX = p.H * p.W, PW = p.W + 2, NPOS = (p.H + 2) * PW;
    const int Wt = p.W >> 1, NTIL = (p.H >> 1) * Wt;

    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.u, 0, p.ubytes, 0x00020000);
    // this wave's filters: iteration g (16 channels) needs fragments 4 g .. 4 g + 3 = (xi of the wave, column half)
    const unsigned bo = (unsigned)((((nt * 8 + wv) * (4 * WIT)) * 64 + lane) * 16);
    f32x4 bq[RING][4];
    auto issue_b = [&](int it) {
#pragma unroll
        for (int f = 0; f < 4; ++f) bq[it % RING][f] = bload(ur, bo + (unsigned)((4 * it + f) * 1024));
    };
#pragma unroll
    for (int it = 0; it < RING; ++it) issue_b(it);

    stage_image(p, L, tid, img, PW, NPOS);

    // xi = (i, j): i = wv / 2 for both of the wave's positions, j = 2 (wv % 2) + xl.  Row i of B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0;
    // 0 1 0 -1] picks pixels (ra, rb) with sign sg:  i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3.
    auto pick = [](int i, int& a, int& b, float& sg) {
        a = i == 0 ? 0 : (i == 2 ? 2 : 1);
        b = i == 3 ? 3 : (i == 2 ? 1 : 2);
        sg = i == 1 ? 1.f : -1.f;
    };
    int ra, rb, ca[2], cb[2];
    float sgi, sgj[2];
    pick(wv >> 1, ra, rb, sgi);
    pick(2 * (wv & 1), ca[0], cb[0], sgj[0]);
    pick(2 * (wv & 1) + 1, ca[1], cb[1], sgj[1]);
    int t = l15;
    if (t >= NTIL) t = NTIL - 1;                            // (rows past the last tile repeat it: never used)
    const int ty = t / Wt, tx = t - ty * Wt;
    const float* const lt = L + ((2 * ty) * PW + 2 * tx) * PP + 4 * q;     // pixel (0, 0) of the tile's 4 x 4 patch
    int off[2][4];
#pragma unroll
    for (int xl = 0; xl < 2; ++xl) {
        off[xl][0] = (ra * PW + ca[xl]) * PP; off[xl][1] = (ra * PW + cb[xl]) * PP;
        off[xl][2] = (rb * PW + ca[xl]) * PP; off[xl][3] = (rb * PW + cb[xl]) * PP;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int xl = 0; xl < 2; ++xl)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[xl][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();                                        // the image is complete
    auto load_a = [&](int it, f32x4 (&d)[2][4]) {
#pragma unroll
        for (int xl = 0; xl < 2; ++xl)
#pragma unroll
            for (int k = 0; k < 4; ++k) d[xl][k] = *reinterpret_cast<const f32x4*>(lt + off[xl][k] + 16 * it);
    };
    f32x4 dn[2][4];
    load_a(0, dn);
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        f32x4 bc[4], a[2];
        __builtin_amdgcn_sched_barrier(0);                  // (nothing of iteration it + 1 is hoisted above this one's MFMAs)
#pragma unroll
        for (int f = 0; f < 4; ++f) bc[f] = bq[it % RING][f];
#pragma unroll
        for (int xl = 0; xl < 2; ++xl)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                a[xl][e] = __builtin_fmaf(sgi, __builtin_fmaf(sgj[xl], dn[xl][3][e], dn[xl][2][e]), __builtin_fmaf(sgj[xl], dn[xl][1][e], dn[xl][0][e]));
        if (it + RING < WIT) issue_b(it + RING);
        if (it + 1 < WIT) load_a(it + 1, dn);               // the next group's pixels read under this group's 16 MFMAs
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int xl = 0; xl < 2; ++xl)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[xl][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[xl][j], bc[2 * xl + u][j], acc[xl][u], 0, 0, 0);
    }

    const EpiOps eo = epi_request<ROLE>(p, tid, img, PX, n0);
    __syncthreads();                                        // every wave is done with the image: the region becomes the product image
#pragma unroll
    for (int xl = 0; xl < 2; ++xl)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)                      // M[xi][tile 4 q + r][column 16 u + l15]
                L[((2 * wv + xl) * 16 + 4 * q + r) * RP + 16 * u + l15] = acc[xl][u][r];
    __syncthreads();
    // ---- inverse transform of thread (pixel m, 4 channels):  Y = A^T M A,  A^T = [1 1 1 0; 0 1 -1 -1] ------------------------
    const int m = tid >> 3, cq = tid & 7;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (eo.live) {
        const int h = m / p.W, w = m - h * p.W;
        const int tl = (h >> 1) * Wt + (w >> 1), oa = h & 1, ob = w & 1;
        f32x4 rowv[3];
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const float* const mp = &L[(((oa + di) * 4 + ob) * 16 + tl) * RP + 4 * cq];
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(mp);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(mp + 16 * RP);
            const f32x4 x2 = *reinterpret_cast<const f32x4*>(mp + 32 * RP);
#pragma unroll
            for (int e = 0; e < 4; ++e) rowv[di][e] = ob ? (x0[e] - x1[e]) - x2[e] : (x0[e] + x1[e]) + x2[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = oa ? (rowv[0][e] - rowv[1][e]) - rowv[2][e] : (rowv[0][e] + rowv[1][e]) + rowv[2][e];
    }
    epi_finish<ROLE>(p, v, eo, sred, tid, img, n0);
}

// ---- filter re-layout ---------------------------------------------------------------------------------------------------
// One block of 256 threads = one 1 KB fragment: (column tile nt of 32, wave w = 32 input channels, tap, 16-channel group g,
// 16-column half u); thread t = 4 lane + j holds B[k-lane lane / 16, MFMA j][column lane % 16] =
// filter(n = 32 nt + 16 u + lane % 16, tap, c = 32 w + 16 g + 4 (lane / 16) + j).  role 0: w[n][r][s][c] (forward; n = Cout,
// c = Cin); role 1: w[c][2 - r][2 - s][n] (input gradient; n = Cin, c = Cout).  Same record as wino::WinoEnt.
struct SEnt {
    const float* w;
    float* u;
    int Cn, Ck, role, blk0;
};

__global__ __launch_bounds__(256) void smap_weights(const SEnt* __restrict__ ents, const int* __restrict__ blk_ent) {
    const SEnt e = ents[blk_ent[blockIdx.x]];
    const int lb = (int)blockIdx.x - e.blk0;
    const int u = lb & 1, g = (lb >> 1) & 1;
    const int rest = lb >> 2;
    const int tap = rest % 9, r2 = rest / 9;
    const int nw = e.Ck >> 5;
    const int w = r2 % nw, nt = r2 / nw;
    const int t = threadIdx.x, lane = t >> 2, j = t & 3;
    const int n = 32 * nt + 16 * u + (lane & 15), c = 32 * w + 16 * g + 4 * (lane >> 4) + j;
    const int r = tap / 3, s = tap % 3;
    const float val = e.role == 0 ? e.w[((int64_t)(n * 3 + r) * 3 + s) * e.Ck + c]
                                  : e.w[((int64_t)(c * 3 + (2 - r)) * 3 + (2 - s)) * e.Cn + n];
    e.u[(int64_t)lb * 256 + t] = val;
}

// U = G g G^T of the same filters for conv_smapw.  One block of 256 threads = (column tile nt, 16-channel group g, column half
// u): thread t = 4 lane + j loads the 3 x 3 filter of (n = 32 nt + 16 u + lane % 16, c = 16 g + 4 (lane / 16) + j) once and writes
// its 16 values U[xi] into the 16 fragments (((nt * 8 + xi / 2) * 16 + g) * 2 + xi % 2) * 2 + u - 1 KB per store instruction.
// G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1] (the arithmetic of wino::wino_weights).
__global__ __launch_bounds__(256) void smapw_weights(const SEnt* __restrict__ ents, const int* __restrict__ blk_ent) {
    const SEnt e = ents[blk_ent[blockIdx.x]];
    const int lb = (int)blockIdx.x - e.blk0;
    const int u = lb & 1, g = (lb >> 1) & 15, nt = lb >> 5;
    const int t = threadIdx.x, lane = t >> 2, j = t & 3;
    const int n = 32 * nt + 16 * u + (lane & 15), c = 16 * g + 4 * (lane >> 4) + j;
    float f[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s)
            f[r][s] = e.role == 0 ? e.w[((int64_t)(n * 3 + r) * 3 + s) * e.Ck + c]
                                  : e.w[((int64_t)(c * 3 + (2 - r)) * 3 + (2 - s)) * e.Cn + n];
    float a[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        a[0][s] = f[0][s];
        a[1][s] = 0.5f * ((f[0][s] + f[2][s]) + f[1][s]);
        a[2][s] = 0.5f * ((f[0][s] + f[2][s]) - f[1][s]);
        a[3][s] = f[2][s];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float u4[4] = {a[i][0], 0.5f * ((a[i][0] + a[i][2]) + a[i][1]), 0.5f * ((a[i][0] + a[i][2]) - a[i][1]), a[i][2]};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int xi = 4 * i + jj;
            e.u[(int64_t)((((nt * 8 + (xi >> 1)) * 16 + g) * 2 + (xi & 1)) * 2 + u) * 256 + t] = u4[jj];
        }
    }
}

}  // namespace smap

// Which problems the kernel serves (3x3 / stride 1 / pad 1 is implied by the entry points).
static bool smap_shape_ok(int N, int H, int W, int Ci, int Co) {
    if (N <= 0 || H < 1 || W < 1 || Ci != smap::C || Co % 32 != 0 || Co <= 0 || Co > 4096) return false;
    if (H * W > smap::MAXPX || (H + 2) * (W + 2) > smap::MAXPOS) return false;
    if ((int64_t)N * H * W * (Ci > Co ? Ci : Co) * 4 >= 0x7fffffffLL) return false;
    return true;
}

// 0: not served; otherwise the number of workgroups of the launch (images x column tiles of 32)
extern "C" int advmix_conv_smap_config(int N, int H, int W, int Ci, int Co) {
    if (!smap_shape_ok(N, H, W, Ci, Co)) return 0;
    const int64_t wgs = (int64_t)N * (Co / 32);
    return wgs > 0x7fffffff ? 0x7fffffff : (int)wgs;
}

// floats of one re-laid image (forward or input gradient) of a 3x3 Cn x Ck filter bank
extern "C" int64_t advmix_smap_u_floats(int Co, int Ci) { return (int64_t)9 * Co * Ci; }

// Re-lay the filters of n convs in one launch.  ``ents`` (device): records {w, u, Cn, Ck, role, first block}, ``blk_ent``
// (device): the record index of each of the ``blocks`` workgroups (a record owns (Cn / 32) * (Ck / 32) * 36 consecutive ones).
extern "C" int advmix_smap_weights(const void* ents, const int* blk_ent, int blocks, void* stream) {
    if (!ents || !blk_ent || blocks <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(smap::smap_weights, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const smap::SEnt*)ents, blk_ent);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

static int smap_fill(smap::SP& p, const float* x, const float* u, float* y, int N, int H, int W, int Ci, int Co) {
    if (!x || !u || !y || !smap_shape_ok(N, H, W, Ci, Co)) return ADVMIX_EINVAL;
    p = smap::SP{};
    p.x = x; p.u = u; p.y = y;
    p.N = N; p.H = H; p.W = W; p.Co = Co;
    p.xbytes = (int)((int64_t)N * H * W * Ci * 4);
    p.ybytes = (int)((int64_t)N * H * W * Co * 4);
    p.ubytes = (int)((int64_t)9 * Co * Ci * 4);
    return ADVMIX_OK;
}

static int smap_slots(const int* stats_ns) {
    int ns = stats_ns && *stats_ns > 0 ? *stats_ns : advmix_opts().stat_slots;
    if (ns <= 0 || ns > ADVMIX_STAT_SLOTS_MAX || (ns & (ns - 1))) ns = 16;
    return ns;
}

static int smap_launch(int role, const smap::SP& p, hipStream_t st) {
    const dim3 g(p.Co / 32, p.N);
    if (role) hipLaunchKernelGGL(smap::conv_smap<1>, g, dim3(smap::NT), 0, st, p);
    else hipLaunchKernelGGL(smap::conv_smap<0>, g, dim3(smap::NT), 0, st, p);
    if (advmix_opts().trace_shapes) {
        char nm[32];
        snprintf(nm, sizeof nm, "conv_smap<%d>", role);
        advmix_trace_launch(nm, g, role == 0 ? (p.stats ? "fwd+sums" : (p.bn_gamma ? "fwd+bn_eval" : "fwd")) : (p.bnb_c ? "dgrad+bnb" : "dgrad"),
                            p.N, p.H, p.W, smap::C, p.H, p.W, p.Co, 3, 3, 1, 2.0 * p.N * (double)p.H * p.W * p.Co * smap::C * 9);
    }
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// advmix_conv3x3_wino_fwd's arguments and semantics for the small-map kernel: ``u`` is the role 0 image of advmix_smap_weights.
// Returns ADVMIX_EINVAL (nothing launched) for shapes the kernel does not serve (advmix_conv_smap_config == 0).
// Semantics: lib/models/pose_hrnet.py:22-57 (conv3x3 + BatchNorm2d (+ residual) + ReLU of a BasicBlock).
extern "C" int advmix_conv3x3_smap_fwd(const float* x, const float* u, float* y, int N, int H, int W, int Ci, int Co,
                                       const float* bn_gamma, const float* bn_beta, const float* bn_rm, const float* bn_rv,
                                       float bn_eps, const float* residual, int act, double* stats, int* stats_ns, void* stream) {
    if ((bn_gamma != nullptr) != (bn_beta && bn_rm && bn_rv)) return ADVMIX_EINVAL;
    if (stats && !stats_ns) return ADVMIX_EINVAL;
    if (advmix_opts().deterministic && stats) return ADVMIX_EINVAL;        // fp64 atomics: the ordered form is conv_direct's
    smap::SP p;
    int rc = smap_fill(p, x, u, y, N, H, W, Ci, Co);
    if (rc) return rc;
    p.bn_gamma = bn_gamma; p.bn_beta = bn_beta; p.bn_rm = bn_rm; p.bn_rv = bn_rv; p.bn_eps = bn_eps;
    p.res = residual; p.act = act; p.stats = stats;
    p.stats_nbg = smap_slots(stats_ns);
    rc = smap_launch(0, p, (hipStream_t)stream);
    if (rc == ADVMIX_OK && stats_ns) *stats_ns = p.stats_nbg;
    return rc;
}

// advmix_conv3x3_wino_dgrad's arguments and semantics: dx = conv(dy, rotated transposed filters) + addend from the role 1
// image ``u`` (n = Cin, k = Cout = 256); with ``bn_c`` the BatchNorm-backward epilogue of advmix_conv_tr_w_bnb.
extern "C" int advmix_conv3x3_smap_dgrad(const float* dy, const float* u, const float* addend, float* dx, int N, int H, int W,
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
    smap::SP p;
    int rc = smap_fill(p, dy, u, dx, N, H, W, Co, Ci);      // the gradient conv reads Co channels and writes Ci
    if (rc) return rc;
    p.res = addend;
    if (bn_c) {
        p.stats = stats; p.stats_nbg = smap_slots(stats_ns);
        p.bnb_mask = act_mask; p.bnb_c = bn_c; p.bnb_mean = bn_mean; p.bnb_invstd = bn_invstd;
        p.bnb_gamma = bn_gamma; p.bnb_beta = bn_beta; p.bnb_act = act;
    }
    rc = smap_launch(1, p, (hipStream_t)stream);
    if (rc == ADVMIX_OK && bn_c) *stats_ns = p.stats_nbg;
    return rc;
}

// ---- Winograd variant (conv_smapw): even H and W only -------------------------------------------------------------------
static bool smapw_shape_ok(int N, int H, int W, int Ci, int Co) {
    return smap_shape_ok(N, H, W, Ci, Co) && !(H & 1) && !(W & 1) && (H / 2) * (W / 2) <= 16;
}

extern "C" int advmix_conv_smapw_config(int N, int H, int W, int Ci, int Co) {
    if (!smapw_shape_ok(N, H, W, Ci, Co)) return 0;
    const int64_t wgs = (int64_t)N * (Co / 32);
    return wgs > 0x7fffffff ? 0x7fffffff : (int)wgs;
}

extern "C" int64_t advmix_smapw_u_floats(int Co, int Ci) { return (int64_t)16 * Co * Ci; }

// records as advmix_smap_weights'; a record owns (Cn / 32) * 32 workgroups and needs Ck == 256
extern "C" int advmix_smapw_weights(const void* ents, const int* blk_ent, int blocks, void* stream) {
    if (!ents || !blk_ent || blocks <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(smap::smapw_weights, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const smap::SEnt*)ents, blk_ent);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

static int smapw_launch(int role, const smap::SP& p, hipStream_t st) {
    const dim3 g(p.Co / 32, p.N);
    if (role) hipLaunchKernelGGL(smap::conv_smapw<1>, g, dim3(smap::NT), 0, st, p);
    else hipLaunchKernelGGL(smap::conv_smapw<0>, g, dim3(smap::NT), 0, st, p);
    if (advmix_opts().trace_shapes) {
        char nm[32];
        snprintf(nm, sizeof nm, "conv_smapw<%d>", role);
        advmix_trace_launch(nm, g, role == 0 ? (p.stats ? "fwd+sums" : (p.bn_gamma ? "fwd+bn_eval" : "fwd")) : (p.bnb_c ? "dgrad+bnb" : "dgrad"),
                            p.N, p.H, p.W, smap::C, p.H, p.W, p.Co, 3, 3, 1, 2.0 * p.N * (double)p.H * p.W * p.Co * smap::C * 9);
    }
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_conv3x3_smapw_fwd(const float* x, const float* u, float* y, int N, int H, int W, int Ci, int Co,
                                        const float* bn_gamma, const float* bn_beta, const float* bn_rm, const float* bn_rv,
                                        float bn_eps, const float* residual, int act, double* stats, int* stats_ns, void* stream) {
    if ((bn_gamma != nullptr) != (bn_beta && bn_rm && bn_rv)) return ADVMIX_EINVAL;
    if (stats && !stats_ns) return ADVMIX_EINVAL;
    if (advmix_opts().deterministic && stats) return ADVMIX_EINVAL;
    if (!smapw_shape_ok(N, H, W, Ci, Co)) return ADVMIX_EINVAL;
    smap::SP p;
    int rc = smap_fill(p, x, u, y, N, H, W, Ci, Co);
    if (rc) return rc;
    p.ubytes = (int)((int64_t)16 * Co * Ci * 4);
    p.bn_gamma = bn_gamma; p.bn_beta = bn_beta; p.bn_rm = bn_rm; p.bn_rv = bn_rv; p.bn_eps = bn_eps;
    p.res = residual; p.act = act; p.stats = stats;
    p.stats_nbg = smap_slots(stats_ns);
    rc = smapw_launch(0, p, (hipStream_t)stream);
    if (rc == ADVMIX_OK && stats_ns) *stats_ns = p.stats_nbg;
    return rc;
}

extern "C" int advmix_conv3x3_smapw_dgrad(const float* dy, const float* u, const float* addend, float* dx, int N, int H, int W,
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
    if (!smapw_shape_ok(N, H, W, Co, Ci)) return ADVMIX_EINVAL;
    smap::SP p;
    int rc = smap_fill(p, dy, u, dx, N, H, W, Co, Ci);
    if (rc) return rc;
    p.ubytes = (int)((int64_t)16 * Co * Ci * 4);
    p.res = addend;
    if (bn_c) {
        p.stats = stats; p.stats_nbg = smap_slots(stats_ns);
        p.bnb_mask = act_mask; p.bnb_c = bn_c; p.bnb_mean = bn_mean; p.bnb_invstd = bn_invstd;
        p.bnb_gamma = bn_gamma; p.bnb_beta = bn_beta; p.bnb_act = act;
    }
    rc = smapw_launch(1, p, (hipStream_t)stream);
    if (rc == ADVMIX_OK && bn_c) *stats_ns = p.stats_nbg;
    return rc;
}
