// Winograd weight gradient for the 3x3 / stride 1 / pad 1 branch convs (round 5) - F(3x3, 2x2): per 2x2 tile of dy and its
// 4x4 input patch the 3x3 gradient needs 16 multiplies per channel pair instead of 36,
//     dW = A'^T [ sum_tiles (G' dY G'^T)^T-outer-(B^T d B) ] A'      A'^T = [1 1 1 0; 0 1 -1 0; 0 1 1 -1],
//     G' = [1 0; 1/2 1/2; 1/2 -1/2; 0 1],  B^T as in the forward kernel (conv_wino.hip) - checked against autograd in numpy,
// i.e. 16 independent [Co x tiles] . [tiles x Ci] products, one per position xi of the transformed patch: 2.25x fewer MFMAs
// than conv_wgrad / wgrad3x3_c32, which sit at 0.57-0.66 of the fp32 matrix peak and are NOT hidden by the other launch
// lanes (knock-out of every weight gradient: 43.9 -> 35.8 ms per step, profiles/r05l).
//
//   * A unit of work = (block of 32 tiles, 32 output channels, 32 input channels); a workgroup walks a run of blocks for ONE
//     (problem, channel pair), accumulates dU in registers and merges ONCE with fp32 atomics (as the grouped kernels do).
//   * v_mfma_f32_16x16x4_f32: k = four tiles.  Wave (qa, qb) owns the 16 x 16 quadrant (co in 16 qa .., ci in 16 qb ..) for
//     ALL 16 xi (64 accumulator registers), so the inverse transform A'^T dU A' needs no exchange between waves.  Lane
//     (c = lane % 16, kl = lane / 16) handles tile 4 j + kl: the 4x4 patch of input channel 16 qb + c and the 2x2 of output
//     channel 16 qa + c, both transformed in registers (44 additions / halvings), 16 MFMAs per four tiles.
//   * Operands come from LDS laid out CHANNEL-major (a lane reads two adjacent pixels of ITS channel as one ds_read_b64;
//     plane pitches with an odd number of bank pairs); the staging transposes: two 16-byte global loads (4 channels of two
//     adjacent pixels) become four ds_write_b64.
#include "common.h"
#include <stdio.h>

namespace wgw {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned OOB = 0x80000000u;
constexpr int MAXP = 8;

struct WP {
    const float* x[MAXP];
    const float* dy[MAXP];
    float* dw[MAXP];
    int n;                       // problems of ONE geometry in this launch
    int N, H, W, Ci, Co;
    int Ht, Wt, nbw, nblk;       // tiles per column / row; blocks per row / per image
    int nblocks;                 // N * nblk
    int runs;                    // runs of blocks per (problem, channel pair): workgroups = n * pairs * runs
    int bpr;                     // blocks per run
    int xbytes, ybytes;
    // round 6: problem i's x operand may be the RAW output c of the conv that precedes it, its train-mode BatchNorm + ReLU not
    // applied (conv_wino's INBN forward never wrote the activation): bn_mean[i] != NULL -> relu(fma((c - mean) * invstd, gamma,
    // beta)) - norm_apply_slots' very expression - is applied to every x element while it is staged; NULL: x as it is
    const float* bn_mean[MAXP];
    const float* bn_invstd[MAXP];
    const float* bn_gamma[MAXP];
    const float* bn_beta[MAXP];
};

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}

template <int LBW>
__global__ __launch_bounds__(256, 3) void wgrad_wino(const WP p) {
    constexpr int BW = 1 << LBW, BH = 32 >> LBW;
    constexpr int PH = 2 * BH + 2, PW = 2 * BW + 2;        // input patch of a block (pixels)
    constexpr int DH = 2 * BH, DW = 2 * BW;                // its output pixels
    constexpr int PLX = PH * PW + 2, PLY = DH * DW + 2;    // plane pitches (floats): 182 / 130 - an odd number of bank pairs
    static_assert((PLX / 2) % 2 == 1 && (PLY / 2) % 2 == 1 && PW % 2 == 0, "channel planes: conflict-free b64 reads");
    __shared__ __attribute__((aligned(16))) float XL[32 * PLX];
    __shared__ __attribute__((aligned(16))) float DL[32 * PLY];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c16 = lane & 15, kl = lane >> 4;
    const int qa = wid >> 1, qb = wid & 1;
    const int pairs_ci = p.Ci >> 5, pairs = (p.Co >> 5) * pairs_ci;
    // workgroup -> (problem, channel pair, run)
    int g = blockIdx.x;
    const int run = g % p.runs; g /= p.runs;
    const int pair = g % pairs, prob = g / pairs;
    const int cot = pair / pairs_ci, cit = pair - cot * pairs_ci;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x[prob], 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy[prob], 0, p.ybytes, 0x00020000);

    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the staging thread's channel quad is fixed (256 threads, 8 quads): its BatchNorm parameters, once per workgroup
    const bool xbn = p.bn_mean[prob] != nullptr;            // (uniform)
    f32x4 bmu = {0.f, 0.f, 0.f, 0.f}, bis = bmu, bga = bmu, bbe = bmu;
    if (xbn) {
        const int ch = cit * 32 + (tid & 7) * 4;
        bmu = *reinterpret_cast<const f32x4*>(p.bn_mean[prob] + ch);
        bis = *reinterpret_cast<const f32x4*>(p.bn_invstd[prob] + ch);
        bga = *reinterpret_cast<const f32x4*>(p.bn_gamma[prob] + ch);
        bbe = *reinterpret_cast<const f32x4*>(p.bn_beta[prob] + ch);
    }

    const float* const xl = XL + (16 * qb + c16) * PLX;     // this lane's input-channel plane
    const float* const dl = DL + (16 * qa + c16) * PLY;     // ... and output-channel plane
    const int b0 = run * p.bpr, b1 = min(p.nblocks, b0 + p.bpr);
    for (int b = b0; b < b1; ++b) {
        const int img = b / p.nblk, rblk = b - img * p.nblk;
        const int bby = rblk / p.nbw, bbx = rblk - bby * p.nbw;
        // ---- stage the block: x patch (32 channels of the pair's input slice) and dy (32 of its output slice), transposed ----
        {
            constexpr int NX = PH * (PW / 2) * 8, NXI = (NX + 255) / 256;
            constexpr int NY = DH * (DW / 2) * 8, NYI = (NY + 255) / 256;
            const int hb = 2 * BH * bby - 1, wb = 2 * BW * bbx - 1;
            f32x4 sx[NXI][2], sy[NYI][2];
            unsigned okx = 0u;                              // which staged x pieces lie inside the image (the ring pads the ACTIVATION)
            static_assert(2 * NXI <= 32, "one bit per staged piece");
#pragma unroll
            for (int it = 0; it < NXI; ++it) {
                const int s = tid + 256 * it;
                const int pp = s >> 3, cs = s & 7;
                const int pr = pp / (PW / 2), pc = 2 * (pp % (PW / 2));
                const int h = hb + pr;
                const bool okh = s < NX && (unsigned)h < (unsigned)p.H;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int w = wb + pc + k;
                    const bool ok = okh && (unsigned)w < (unsigned)p.W;
                    sx[it][k] = bload(xr, ok ? (unsigned)((((img * p.H + h) * p.W + w) * p.Ci + cit * 32 + cs * 4) * 4) : OOB);
                    okx |= ok ? (1u << (2 * it + k)) : 0u;
                }
            }
#pragma unroll
            for (int it = 0; it < NYI; ++it) {
                const int s = tid + 256 * it;
                const int pp = s >> 3, cs = s & 7;
                const int pr = pp / (DW / 2), pc = 2 * (pp % (DW / 2));
                const int h = 2 * BH * bby + pr;
                const bool okh = s < NY && h < p.H;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int w = 2 * BW * bbx + pc + k;
                    const bool ok = okh && w < p.W;
                    sy[it][k] = bload(yr, ok ? (unsigned)((((img * p.H + h) * p.W + w) * p.Co + cot * 32 + cs * 4) * 4) : OOB);
                }
            }
            if (xbn) {
#pragma unroll
                for (int it = 0; it < NXI; ++it)
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = fmaxf(__builtin_fmaf((sx[it][k][e] - bmu[e]) * bis[e], bga[e], bbe[e]), 0.f);
                            sx[it][k][e] = ((okx >> (2 * it + k)) & 1u) ? t : 0.f;
                        }
            }
            __syncthreads();                                // the previous block's operands have been read
#pragma unroll
            for (int it = 0; it < NXI; ++it) {
                const int s = tid + 256 * it;
                const int pp = s >> 3, cs = s & 7;
                const int pr = pp / (PW / 2), pc = 2 * (pp % (PW / 2));
                if (s < NX) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        *reinterpret_cast<f32x2*>(&XL[(4 * cs + e) * PLX + pr * PW + pc]) = f32x2{sx[it][0][e], sx[it][1][e]};
                }
            }
#pragma unroll
            for (int it = 0; it < NYI; ++it) {
                const int s = tid + 256 * it;
                const int pp = s >> 3, cs = s & 7;
                const int pr = pp / (DW / 2), pc = 2 * (pp % (DW / 2));
                if (s < NY) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        *reinterpret_cast<f32x2*>(&DL[(4 * cs + e) * PLY + pr * DW + pc]) = f32x2{sy[it][0][e], sy[it][1][e]};
                }
            }
            __syncthreads();
        }
        // ---- 8 steps of four tiles -----------------------------------------------------------------------------------
#pragma unroll 2
        for (int j = 0; j < 8; ++j) {
            const int t = 4 * j + kl;
            const int bxl = t & (BW - 1), byl = t >> LBW;
            const float* xp = xl + (2 * byl) * PW + 2 * bxl;
            const float* dp = dl + (2 * byl) * DW + 2 * bxl;
            float d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x2 a = *reinterpret_cast<const f32x2*>(xp + r * PW);
                const f32x2 b2 = *reinterpret_cast<const f32x2*>(xp + r * PW + 2);
                d[r][0] = a[0]; d[r][1] = a[1]; d[r][2] = b2[0]; d[r][3] = b2[1];
            }
            const f32x2 y0 = *reinterpret_cast<const f32x2*>(dp), y1 = *reinterpret_cast<const f32x2*>(dp + DW);
            // V = B^T d B
            float rc[4][4], v[4][4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                rc[0][cc] = d[0][cc] - d[2][cc];
                rc[1][cc] = d[1][cc] + d[2][cc];
                rc[2][cc] = d[2][cc] - d[1][cc];
                rc[3][cc] = d[1][cc] - d[3][cc];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i][0] = rc[i][0] - rc[i][2];
                v[i][1] = rc[i][1] + rc[i][2];
                v[i][2] = rc[i][2] - rc[i][1];
                v[i][3] = rc[i][1] - rc[i][3];
            }
            // M = G' dY G'^T
            float m[4][2], u[4][4];
            m[0][0] = y0[0]; m[0][1] = y0[1];
            m[1][0] = 0.5f * (y0[0] + y1[0]); m[1][1] = 0.5f * (y0[1] + y1[1]);
            m[2][0] = 0.5f * (y0[0] - y1[0]); m[2][1] = 0.5f * (y0[1] - y1[1]);
            m[3][0] = y1[0]; m[3][1] = y1[1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u[i][0] = m[i][0];
                u[i][1] = 0.5f * (m[i][0] + m[i][1]);
                u[i][2] = 0.5f * (m[i][0] - m[i][1]);
                u[i][3] = m[i][1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[4 * i + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[i][jj], v[i][jj], acc[4 * i + jj], 0, 0, 0);
        }
    }
    // ---- dW = A'^T dU A'  (in registers), merged with atomics -----------------------------------------------------------
    float* const dw = p.dw[prob];
    const int ci = cit * 32 + 16 * qb + c16;
    const int co0 = cot * 32 + 16 * qa + 4 * kl;            // D layout of 16x16x4: lane -> column lane % 16, rows 4 (lane / 16) + reg
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        float t3[4][3];                                     // column half: dU A'
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a0 = acc[4 * i + 0][reg], a1 = acc[4 * i + 1][reg], a2 = acc[4 * i + 2][reg], a3 = acc[4 * i + 3][reg];
            t3[i][0] = (a0 + a1) + a2;
            t3[i][1] = a1 - a2;
            t3[i][2] = (a1 + a2) - a3;
        }
        float* const row = dw + (int64_t)(co0 + reg) * 9 * p.Ci + ci;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            atomicAdd(row + (0 * 3 + s) * p.Ci, (t3[0][s] + t3[1][s]) + t3[2][s]);
            atomicAdd(row + (1 * 3 + s) * p.Ci, t3[1][s] - t3[2][s]);
            atomicAdd(row + (2 * 3 + s) * p.Ci, (t3[1][s] + t3[2][s]) - t3[3][s]);
        }
    }
}

}  // namespace wgw

static bool wgw_shape_ok(int N, int H, int W, int Ci, int Co) {
    if (N <= 0 || H < 4 || W < 4 || (H & 1) || (W & 1)) return false;
    if (Ci % 32 != 0 || Co % 32 != 0 || Ci > 256 || Co > 256) return false;
    if ((int64_t)N * H * W * (Ci > Co ? Ci : Co) * 4 >= 0x7fffffffLL) return false;
    return true;
}

// 0 = not served, else the number of (block, channel pair) units of ONE problem - what ops.py decides on (enough work to
// fill the chip and to amortise the merge).
extern "C" int advmix_wgrad_wino_config(int N, int H, int W, int Ci, int Co) {
    if (!wgw_shape_ok(N, H, W, Ci, Co)) return 0;
    const int Ht = H / 2, Wt = W / 2;
    const int64_t w3 = (int64_t)cdiv(Wt, 8) * cdiv(Ht, 4), w2 = (int64_t)cdiv(Wt, 4) * cdiv(Ht, 8);
    const int64_t nblk = w3 <= w2 ? w3 : w2;
    if ((int64_t)Ht * Wt * 10 < nblk * 32 * 6) return 0;   // blocks less than 60 % full (8 x 6 maps: 12 of 32 tiles): the MFMAs saved are wasted again
    const int64_t units = (int64_t)N * nblk * (Ci / 32) * (Co / 32);
    return units > 0x7fffffff ? 0x7fffffff : (int)units;
}

// Weight gradients of n (1-8) 3x3 / stride 1 / pad 1 convs of ONE geometry in one launch, ACCUMULATED into dw[i]
// ([Co][3][3][Ci], fp32 atomics): dw[i] += sum_pixels dy[i] (x) x[i].  dy[i]: [N,H,W,Co], x[i]: [N,H,W,Ci].  Replaces
// advmix_conv_wgrad_group for the branch convs of HRNet (pose_hrnet.py:22-57 backward).  ADVMIX_EINVAL (nothing launched):
// odd sizes, channel counts that are not multiples of 32, deterministic mode (atomics), n out of range.
//
// advmix_conv3x3_wgrad_wino_group_bn: the same with a per-problem BatchNorm on the x operand - bn_mean / bn_invstd / bn_gamma /
// bn_beta are arrays of n pointers (the arrays or single entries may be NULL); where entry i is given, x[i] is the RAW output c
// of the conv that precedes problem i and relu(BN(c)) (saved batch statistics) is applied while x is staged: the activation
// advmix_conv3x3_wino_fwd_inbn never wrote.
extern "C" int advmix_conv3x3_wgrad_wino_group_bn(int n, const float* const* dy, const float* const* x, float* const* dw,
                                                  const float* const* bn_mean, const float* const* bn_invstd,
                                                  const float* const* bn_gamma, const float* const* bn_beta, int N, int H, int W,
                                                  int Co, int Ci, void* stream) {
    if (n < 1 || n > wgw::MAXP || !dy || !x || !dw || !wgw_shape_ok(N, H, W, Ci, Co)) return ADVMIX_EINVAL;
    if (advmix_opts().deterministic) return ADVMIX_EINVAL;
    wgw::WP p{};
    for (int i = 0; i < n; ++i) {
        if (!dy[i] || !x[i] || !dw[i]) return ADVMIX_EINVAL;
        p.x[i] = x[i]; p.dy[i] = dy[i]; p.dw[i] = dw[i];
        if (bn_mean && bn_mean[i]) {
            if (!bn_invstd || !bn_gamma || !bn_beta || !bn_invstd[i] || !bn_gamma[i] || !bn_beta[i]) return ADVMIX_EINVAL;
            p.bn_mean[i] = bn_mean[i]; p.bn_invstd[i] = bn_invstd[i]; p.bn_gamma[i] = bn_gamma[i]; p.bn_beta[i] = bn_beta[i];
        }
    }
    p.n = n; p.N = N; p.H = H; p.W = W; p.Ci = Ci; p.Co = Co;
    p.Ht = H / 2; p.Wt = W / 2;
    const int64_t w3 = (int64_t)cdiv(p.Wt, 8) * cdiv(p.Ht, 4), w2 = (int64_t)cdiv(p.Wt, 4) * cdiv(p.Ht, 8);
    const int lbw = w3 <= w2 ? 3 : 2;
    p.nbw = cdiv(p.Wt, 1 << lbw);
    p.nblk = p.nbw * cdiv(p.Ht, 32 >> lbw);
    p.nblocks = N * p.nblk;
    p.xbytes = (int)((int64_t)N * H * W * Ci * 4);
    p.ybytes = (int)((int64_t)N * H * W * Co * 4);
    const int pairs = (Ci / 32) * (Co / 32);
    // runs per (problem, pair): about 768 workgroups in all (three per CU), at least 2 blocks per run
    static const int target = advmix_env_int("ADVMIX_WGW_WGS", 768);
    int runs = target / (n * pairs);
    if (runs < 1) runs = 1;
    if (runs > p.nblocks / 2) runs = p.nblocks / 2 > 0 ? p.nblocks / 2 : 1;
    p.bpr = cdiv(p.nblocks, runs);
    p.runs = cdiv(p.nblocks, p.bpr);
    dim3 g(n * pairs * p.runs);
    if (lbw == 3) hipLaunchKernelGGL(wgw::wgrad_wino<3>, g, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wgw::wgrad_wino<2>, g, dim3(256), 0, (hipStream_t)stream, p);
    if (advmix_opts().trace_shapes) {
        char kd[24];
        snprintf(kd, sizeof kd, "wgrad x%d", n);
        advmix_trace_launch(lbw == 3 ? "wgrad_wino<3>" : "wgrad_wino<2>", g, kd, N, H, W, Ci, H, W, Co, 3, 3, 1,
                            2.0 * n * N * (double)H * W * Co * Ci * 9);
    }
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_conv3x3_wgrad_wino_group(int n, const float* const* dy, const float* const* x, float* const* dw, int N,
                                               int H, int W, int Co, int Ci, void* stream) {
    return advmix_conv3x3_wgrad_wino_group_bn(n, dy, x, dw, nullptr, nullptr, nullptr, nullptr, N, H, W, Co, Ci, stream);
}
