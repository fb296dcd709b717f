// Winograd F(3x3, 2x2) per input phase for 4x4 / stride 2 / pad 1 convolutions (gfx950) - round 5.
//
// The U-Net generator's down convs and the input gradients of its transposed convs (lib/models/Unet_generator.py:60-112:
// Conv2d(k 4, s 2, p 1) / ConvTranspose2d(k 4, s 2, p 1)) are 20 % of the AdvMix step on the direct kernel at 0.65-0.77 of the
// fp32 matrix peak with the MFMA pipe 74 % busy: what is left is the multiply count.  A 4x4 / stride-2 conv is the sum of
// four 2x2 / stride-1 convs, one per PHASE (p, q) of the input (rows of parity p, columns of parity q):
//     out[oy][ox] = sum_{p,q} sum_{r,s in {0,1}} X_pq[oy + r][ox + s] . w[2 r + p][2 s + q],   X_pq[i][j] = x[2 i + p - 1][2 j + q - 1]
// and F(3x3, 2x2) computes a 3x3 output tile of a 2x2 conv from a 4x4 patch with 16 multiplies instead of 36:
//     Y = A^T [ sum_{p,q} (G g_pq G^T) .* (B^T X_pq B) ] A        (2.25x fewer multiplies; the phases add up BEFORE the inverse)
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0; 1/2 1/2; 1/2 -1/2; 0 1]   A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 -1]
// i.e. 16 independent GEMMs [tiles x 4 Cin] . [4 Cin x Cout], one per position xi of the transformed patch.  With Cout and
// 4 Cin in the hundreds to thousands the transforms are small beside the GEMMs, so this first form is NOT fused:
//     wino4_input   x -> V[xi][tile][(p, q), ci]                 (HBM-bound; V = 16/9 of the input's bytes)
//     conv_direct   M[xi] = V[xi] . U[xi]^T, all 16 in one launch (advmix_conv_direct_gemm_batched: a 1x1 conv whose 16
//                   "images" carry their own filters; XCD k works on xi = 2 k, 2 k + 1, so a U slice lives in one L2)
//     wino4_output  M -> y (+ bias)                              (HBM-bound)
// The filters are transformed once per forward pass by wino4_weights (ops.py: WinoBank, kind 'w4').
// Measured against the direct kernel (tools/microbench_wino4.py): profiles/EXPERIMENTS.md section H.
#include "common.h"
#include <stdio.h>

namespace w4 {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}

// ---- filters: U[xi][co][(p, q), ci] = (G g_pq G^T)[xi],  g_pq[r][s] = w[co][2 r + p][2 s + q][ci] -------------------------
// One thread per (co, phase, ci); a record owns Co * 4 * Ci / 256 consecutive blocks.  Same record layout as conv_wino.hip's
// WinoEnt (ops.py builds one table format for every image kind).
struct W4Ent {
    const float* w;
    float* u;
    int Cn, Ck, role, blk0;
};

__global__ __launch_bounds__(256) void wino4_weights(const W4Ent* __restrict__ ents, const int* __restrict__ blk_ent) {
    const W4Ent e = ents[blk_ent[blockIdx.x]];
    const int Co = e.Cn, Ci = e.Ck;
    const int idx = ((int)blockIdx.x - e.blk0) * 256 + (int)threadIdx.x;      // (co * 4 + ph) * Ci + ci
    const int ci = idx % Ci, cp = idx / Ci;
    const int ph = cp & 3, co = cp >> 2;
    if (co >= Co) return;
    const int p = ph >> 1, q = ph & 1;
    float g[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) g[r][s] = e.w[((int64_t)(co * 4 + 2 * r + p) * 4 + 2 * s + q) * Ci + ci];
    float a[4][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        a[0][s] = g[0][s];
        a[1][s] = 0.5f * (g[0][s] + g[1][s]);
        a[2][s] = 0.5f * (g[0][s] - g[1][s]);
        a[3][s] = g[1][s];
    }
    const int64_t plane = (int64_t)Co * 4 * Ci;
    float* const out = e.u + (int64_t)co * 4 * Ci + ph * Ci + ci;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out[(i * 4 + 0) * plane] = a[i][0];
        out[(i * 4 + 1) * plane] = 0.5f * (a[i][0] + a[i][1]);
        out[(i * 4 + 2) * plane] = 0.5f * (a[i][0] - a[i][1]);
        out[(i * 4 + 3) * plane] = a[i][1];
    }
}

struct W4P {
    const float* x;
    float* v;
    const float* m;
    const float* bias;
    float* y;
    int N, H, W, Ci, Co;      // input H x W (even), output H / 2 x W / 2
    int Th, Tw, tiles, rows;  // 3x3 output tiles per column / row, N * Th * Tw, rows = tiles rounded up to 128
    int xbytes;
};

// ---- input transform: thread = (tile, phase, 4 channels); 16 sixteen-byte loads (out-of-image pixels read 0), 16 stores ----
__global__ __launch_bounds__(256) void wino4_input(const W4P p) {
    const int gid = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int Q = p.Ci >> 2;                               // channel quads
    const int tile = gid / p.Ci, rem = gid - tile * p.Ci;  // (4 phases x Q quads = Ci threads per tile)
    if (tile >= p.rows) return;
    const int ph = rem / Q, cq = rem - ph * Q;
    if (tile >= p.tiles) {                                 // rows that pad the tile count to a multiple of 128: zeros (the weight
        float* const o = p.v + (int64_t)tile * (4 * p.Ci) + ph * p.Ci + 4 * cq;     // gradient's GEMMs reduce over ALL rows)
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) *reinterpret_cast<f32x4*>(o + xi * (int64_t)p.rows * (4 * p.Ci)) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int pp = ph >> 1, qq = ph & 1;
    const int n = tile / (p.Th * p.Tw), tr = tile - n * (p.Th * p.Tw);
    const int ty = tr / p.Tw, tx = tr - ty * p.Tw;
    const int h0 = 6 * ty - 1 + pp, w0 = 6 * tx - 1 + qq;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    f32x4 X[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int h = h0 + 2 * a, w = w0 + 2 * b;
            const bool ok = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            X[a][b] = bload(xr, ok ? (unsigned)((((n * p.H + h) * p.W + w) * p.Ci + 4 * cq) * 4) : OOB);
        }
    f32x4 t[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {                          // B^T X
        t[0][b] = X[0][b] - X[2][b];
        t[1][b] = X[1][b] + X[2][b];
        t[2][b] = X[2][b] - X[1][b];
        t[3][b] = X[1][b] - X[3][b];
    }
    const int K = 4 * p.Ci;
    float* const out = p.v + (int64_t)tile * K + ph * p.Ci + 4 * cq;
    const int64_t plane = (int64_t)p.rows * K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                          // (B^T X) B
        *reinterpret_cast<f32x4*>(out + (i * 4 + 0) * plane) = t[i][0] - t[i][2];
        *reinterpret_cast<f32x4*>(out + (i * 4 + 1) * plane) = t[i][1] + t[i][2];
        *reinterpret_cast<f32x4*>(out + (i * 4 + 2) * plane) = t[i][2] - t[i][1];
        *reinterpret_cast<f32x4*>(out + (i * 4 + 3) * plane) = t[i][1] - t[i][3];
    }
}

// ---- output transform: thread = (tile, 4 channels); 16 sixteen-byte loads, up to 9 stores --------------------------------
__global__ __launch_bounds__(256) void wino4_output(const W4P p) {
    const int gid = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int Q = p.Co >> 2;
    const int tile = gid / Q, cq = gid - tile * Q;
    if (tile >= p.tiles) return;
    const int64_t plane = (int64_t)p.rows * p.Co;
    const float* const in = p.m + (int64_t)tile * p.Co + 4 * cq;
    f32x4 M[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) M[i][j] = *reinterpret_cast<const f32x4*>(in + (i * 4 + j) * plane);
    f32x4 s[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                          // A^T M
        s[0][j] = (M[0][j] + M[1][j]) + M[2][j];
        s[1][j] = M[1][j] - M[2][j];
        s[2][j] = (M[1][j] + M[2][j]) - M[3][j];
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + 4 * cq);
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const int n = tile / (p.Th * p.Tw), tr = tile - n * (p.Th * p.Tw);
    const int ty = tr / p.Tw, tx = tr - ty * p.Tw;
#pragma unroll
    for (int a = 0; a < 3; ++a) {                          // (A^T M) A
        const int oy = 3 * ty + a;
        if (oy >= Ho) break;
        f32x4 o[3];
        o[0] = (s[a][0] + s[a][1]) + s[a][2] + bv;
        o[1] = (s[a][1] - s[a][2]) + bv;
        o[2] = ((s[a][1] + s[a][2]) - s[a][3]) + bv;
        float* const yrow = p.y + ((int64_t)(n * Ho + oy) * Wo + 3 * tx) * p.Co + 4 * cq;
#pragma unroll
        for (int b = 0; b < 3; ++b)
            if (3 * tx + b < Wo) *reinterpret_cast<f32x4*>(yrow + (int64_t)b * p.Co) = o[b];
    }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------
// dL/dg_pq = G^T [ sum_tiles (A dy A^T) .* (B^T X_pq B) ] G: the adjoint of the output transform applied to the low-resolution
// operand (3x3 tile -> 4x4), the SAME input transform of the high-resolution one, 16 weight-gradient GEMMs
// dU[xi] = dM[xi]^T . V[xi] (one grouped launch of conv_mfma.hip's kernel), and the adjoint of the filter transform.

// dM[xi][tile][c] = (A d A^T)[xi], d = the tile's 3x3 pixels of ``lo`` (outside the map: 0); A = [1 0 0; 1 1 1; 1 -1 1; 0 0 -1].
// thread = (tile, 4 channels); rows that pad the tile count are written as zeros.
__global__ __launch_bounds__(256) void wino4_output_adj(const float* __restrict__ lo, float* __restrict__ dm, int N, int Ho, int Wo, int C,
                                                        int Th, int Tw, int tiles, int rows) {
    const int gid = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int Q = C >> 2;
    const int tile = gid / Q, cq = gid - tile * Q;
    if (tile >= rows) return;
    const int64_t plane = (int64_t)rows * C;
    float* const out = dm + (int64_t)tile * C + 4 * cq;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (tile >= tiles) {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) *reinterpret_cast<f32x4*>(out + xi * plane) = z;
        return;
    }
    const int n = tile / (Th * Tw), tr = tile - n * (Th * Tw);
    const int ty = tr / Tw, tx = tr - ty * Tw;
    f32x4 d[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int oy = 3 * ty + a, ox = 3 * tx + b;
            d[a][b] = (oy < Ho && ox < Wo) ? *reinterpret_cast<const f32x4*>(lo + ((int64_t)(n * Ho + oy) * Wo + ox) * C + 4 * cq) : z;
        }
    f32x4 s[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {                          // A d
        s[0][b] = d[0][b];
        s[1][b] = (d[0][b] + d[2][b]) + d[1][b];
        s[2][b] = (d[0][b] + d[2][b]) - d[1][b];
        s[3][b] = -d[2][b];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                          // (A d) A^T
        *reinterpret_cast<f32x4*>(out + (i * 4 + 0) * plane) = s[i][0];
        *reinterpret_cast<f32x4*>(out + (i * 4 + 1) * plane) = (s[i][0] + s[i][2]) + s[i][1];
        *reinterpret_cast<f32x4*>(out + (i * 4 + 2) * plane) = (s[i][0] + s[i][2]) - s[i][1];
        *reinterpret_cast<f32x4*>(out + (i * 4 + 3) * plane) = -s[i][2];
    }
}

// dw[cl][2 r + p][2 s + q][ch] += (G^T dU_pq G)[r][s], dU[xi][cl][(p, q), ch]; thread = (cl, phase, 4 channels).
__global__ __launch_bounds__(256) void wino4_wgrad_out(const float* __restrict__ du, float* __restrict__ dw, int Cl, int Ch) {
    const int gid = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int Q = Ch >> 2;
    const int cq = gid % Q, cp = gid / Q;
    const int ph = cp & 3, cl = cp >> 2;
    if (cl >= Cl) return;
    const int p = ph >> 1, q = ph & 1;
    const int64_t plane = (int64_t)Cl * 4 * Ch;
    const float* const in = du + (int64_t)cl * 4 * Ch + ph * Ch + 4 * cq;
    f32x4 t[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                          // G^T dU
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(in + (0 * 4 + j) * plane), u1 = *reinterpret_cast<const f32x4*>(in + (1 * 4 + j) * plane);
        const f32x4 u2 = *reinterpret_cast<const f32x4*>(in + (2 * 4 + j) * plane), u3 = *reinterpret_cast<const f32x4*>(in + (3 * 4 + j) * plane);
        t[0][j] = u0 + 0.5f * (u1 + u2);
        t[1][j] = 0.5f * (u1 - u2) + u3;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {                          // (G^T dU) G
        f32x4* const d0 = reinterpret_cast<f32x4*>(dw + ((int64_t)(cl * 4 + 2 * r + p) * 4 + q) * Ch + 4 * cq);
        f32x4* const d1 = reinterpret_cast<f32x4*>(dw + ((int64_t)(cl * 4 + 2 * r + p) * 4 + 2 + q) * Ch + 4 * cq);
        *d0 = *d0 + (t[r][0] + 0.5f * (t[r][1] + t[r][2]));
        *d1 = *d1 + (0.5f * (t[r][1] - t[r][2]) + t[r][3]);
    }
}

// ---- the transposed form: ConvTranspose2d(k 4, s 2, p 1) forward / the input gradient of a Conv2d(k 4, s 2, p 1) ---------------
// y[2 i - 1 + ky] += x[i] w[ky]: output rows of parity 0 are y[2 m] = x[m] w[1] + x[m - 1] w[3], of parity 1 y[2 m + 1] = x[m + 1] w[0] +
// x[m] w[2] - each a 2-tap correlation over the LOW-resolution map.  With the patch d[a] = x[3 t - 1 + a], a = 0..3, BOTH parities are
// F(3, 2) on the same patch: parity 0 with taps (w[3], w[1]) gives rows 6 t + 2 mu, parity 1 with taps (w[2], w[0]) gives rows
// 6 t - 1 + 2 mu (mu = 0..2) - together the six rows 6 t - 1 ... 6 t + 4, so tile t of floor(H / 3) + 1 per axis owns them and ONE input
// transform per tile serves the four output phases: 16 GEMMs [tiles x Cl] . [Cl x 4 Ch], the phase is part of the column index.
struct W4T {
    const float* x;           // [N][Hl][Wl][Cl]
    float* v;                 // [16][rows][Cl]
    const float* m;           // [16][rows][4 Ch]
    const float* bias;
    const float* add;         // [N][2 Hl][2 Wl][Ch] or null
    float* y;                 // [N][2 Hl][2 Wl][Ch]
    int N, Hl, Wl, Cl, Ch;
    int Th, Tw, tiles, rows;
    int xbytes;
};

// filters: U'[xi][(P, Q), ch][cl] = (G g G^T)[xi], g[r][s] = w[cl][ky(P, r)][kx(Q, s)][ch], ky(0, .) = (3, 1), ky(1, .) = (2, 0).
// w is ch-contiguous, U' is cl-contiguous: a workgroup = 32 cl x 32 ch of one phase, transposed through LDS (both sides coalesced).
// A record (W4Ent: Cn = Cl, Ck = Ch, role 1) owns (Cl / 32) * (Ch / 32) * 4 consecutive blocks.
__global__ __launch_bounds__(256) void wino4t_weights(const W4Ent* __restrict__ ents, const int* __restrict__ blk_ent) {
    __shared__ float S[8][32][33];
    const W4Ent e = ents[blk_ent[blockIdx.x]];
    const int Cl = e.Cn, Ch = e.Ck;
    const int lb = (int)blockIdx.x - e.blk0;
    const int ph = lb & 3, t = lb >> 2;
    const int nct = Ch >> 5;
    const int cl0 = (t / nct) * 32, ch0 = (t % nct) * 32;
    const int P = ph >> 1, Q = ph & 1;
    const int tid = threadIdx.x, j = tid & 31, i0 = tid >> 5;
    float U[4][16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cl = cl0 + i0 + 8 * k;
        float g[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
                g[r][s_] = e.w[((int64_t)(cl * 4 + (3 - P) - 2 * r) * 4 + (3 - Q) - 2 * s_) * Ch + ch0 + j];
        float a[4][2];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            a[0][s_] = g[0][s_];
            a[1][s_] = 0.5f * (g[0][s_] + g[1][s_]);
            a[2][s_] = 0.5f * (g[0][s_] - g[1][s_]);
            a[3][s_] = g[1][s_];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            U[k][i * 4 + 0] = a[i][0];
            U[k][i * 4 + 1] = 0.5f * (a[i][0] + a[i][1]);
            U[k][i * 4 + 2] = 0.5f * (a[i][0] - a[i][1]);
            U[k][i * 4 + 3] = a[i][1];
        }
    }
    const int64_t plane = (int64_t)4 * Ch * Cl;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int x = 0; x < 8; ++x) S[x][j][i0 + 8 * k] = U[k][half * 8 + x];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int chl = i0 + 8 * k;
            float* const out = e.u + (int64_t)(ph * Ch + ch0 + chl) * Cl + cl0 + j;
#pragma unroll
            for (int x = 0; x < 8; ++x) out[(half * 8 + x) * plane] = S[x][chl][j];
        }
        __syncthreads();
    }
}

// input transform: thread = (tile, 4 channels); the tile's 4x4 low-resolution patch starts at (3 ty - 1, 3 tx - 1)
__global__ __launch_bounds__(256) void wino4t_input(const W4T p) {
    const int gid = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int Q = p.Cl >> 2;
    const int tile = gid / Q, cq = gid - tile * Q;
    if (tile >= p.rows) return;
    const int64_t plane = (int64_t)p.rows * p.Cl;
    float* const out = p.v + (int64_t)tile * p.Cl + 4 * cq;
    if (tile >= p.tiles) {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) *reinterpret_cast<f32x4*>(out + xi * plane) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int n = tile / (p.Th * p.Tw), tr = tile - n * (p.Th * p.Tw);
    const int ty = tr / p.Tw, tx = tr - ty * p.Tw;
    const int h0 = 3 * ty - 1, w0 = 3 * tx - 1;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    f32x4 X[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int h = h0 + a, w = w0 + b;
            const bool ok = (unsigned)h < (unsigned)p.Hl && (unsigned)w < (unsigned)p.Wl;
            X[a][b] = bload(xr, ok ? (unsigned)((((n * p.Hl + h) * p.Wl + w) * p.Cl + 4 * cq) * 4) : OOB);
        }
    f32x4 t[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        t[0][b] = X[0][b] - X[2][b];
        t[1][b] = X[1][b] + X[2][b];
        t[2][b] = X[2][b] - X[1][b];
        t[3][b] = X[1][b] - X[3][b];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4*>(out + (i * 4 + 0) * plane) = t[i][0] - t[i][2];
        *reinterpret_cast<f32x4*>(out + (i * 4 + 1) * plane) = t[i][1] + t[i][2];
        *reinterpret_cast<f32x4*>(out + (i * 4 + 2) * plane) = t[i][2] - t[i][1];
        *reinterpret_cast<f32x4*>(out + (i * 4 + 3) * plane) = t[i][1] - t[i][3];
    }
}

// output transform: thread = (tile, output phase, 4 channels); its 3x3 results go to rows 6 ty + 2 mu (P = 0) / 6 ty - 1 + 2 mu (P = 1)
__global__ __launch_bounds__(256) void wino4t_output(const W4T p) {
    const int gid = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int Q = p.Ch >> 2;
    const int tile = gid / p.Ch, rem = gid - tile * p.Ch;
    if (tile >= p.tiles) return;
    const int ph = rem / Q, cq = rem - ph * Q;
    const int P = ph >> 1, Qp = ph & 1;
    const int N4 = 4 * p.Ch;
    const int64_t plane = (int64_t)p.rows * N4;
    const float* const in = p.m + (int64_t)tile * N4 + ph * p.Ch + 4 * cq;
    f32x4 M[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) M[i][j] = *reinterpret_cast<const f32x4*>(in + (i * 4 + j) * plane);
    f32x4 s[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[0][j] = (M[0][j] + M[1][j]) + M[2][j];
        s[1][j] = M[1][j] - M[2][j];
        s[2][j] = (M[1][j] + M[2][j]) - M[3][j];
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + 4 * cq);
    const int Hh = 2 * p.Hl, Wh = 2 * p.Wl;
    const int n = tile / (p.Th * p.Tw), tr = tile - n * (p.Th * p.Tw);
    const int ty = tr / p.Tw, tx = tr - ty * p.Tw;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int Y = 6 * ty - P + 2 * a;
        if ((unsigned)Y >= (unsigned)Hh) continue;
        f32x4 o[3];
        o[0] = (s[a][0] + s[a][1]) + s[a][2] + bv;
        o[1] = (s[a][1] - s[a][2]) + bv;
        o[2] = ((s[a][1] + s[a][2]) - s[a][3]) + bv;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int X = 6 * tx - Qp + 2 * b;
            if ((unsigned)X >= (unsigned)Wh) continue;
            const int64_t off = ((int64_t)(n * Hh + Y) * Wh + X) * p.Ch + 4 * cq;
            f32x4 r = o[b];
            if (p.add) r += *reinterpret_cast<const f32x4*>(p.add + off);
            *reinterpret_cast<f32x4*>(p.y + off) = r;
        }
    }
}

}  // namespace w4

static bool w4_shape_ok(int N, int H, int W, int Ci, int Co) {
    if (N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || Ci % 8 != 0 || Co % 4 != 0) return false;
    const int64_t tiles = (int64_t)N * cdiv(H / 2, 3) * cdiv(W / 2, 3);
    const int64_t rows = (tiles + 127) / 128 * 128;
    if ((int64_t)N * H * W * Ci * 4 >= 0x7fffffffLL || 16 * rows * 4 * Ci * 4 >= 0x7fffffffLL || 16 * rows * Co * 4 >= 0x7fffffffLL ||
        (int64_t)16 * Co * 4 * Ci * 4 >= 0x7fffffffLL)
        return false;
    return true;
}

// floats of the transformed filters of a [Co][4][4][Ci] bank
extern "C" int64_t advmix_wino4_u_floats(int Co, int Ci) { return (int64_t)16 * Co * 4 * Ci; }

// floats of scratch (V then M) advmix_conv4x4s2_wino_fwd needs; 0 = shape not served
extern "C" int64_t advmix_conv4x4s2_wino_ws_floats(int N, int H, int W, int Ci, int Co) {
    if (!w4_shape_ok(N, H, W, Ci, Co)) return 0;
    const int64_t tiles = (int64_t)N * cdiv(H / 2, 3) * cdiv(W / 2, 3);
    const int64_t rows = (tiles + 127) / 128 * 128;
    return 16 * rows * (4 * (int64_t)Ci + Co);
}

// Transform the filters of n convs in one launch (records / block owners as advmix_wino_weights; a record owns
// Cn * 4 * Ck / 256 blocks).  Replaces nothing in the reference - cuDNN prepares its own algorithm behind nn.Conv2d.
extern "C" int advmix_w4_weights(const void* ents, const int* blk_ent, int blocks, void* stream) {
    if (!ents || !blk_ent || blocks <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(w4::wino4_weights, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const w4::W4Ent*)ents, blk_ent);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// advmix_conv_fwd for a 4x4 / stride 2 / pad 1 conv whose filters w[Co][4][4][Ci] were transformed by advmix_w4_weights into
// ``u``: y[N][H/2][W/2][Co] = conv(x[N][H][W][Ci]) + bias.  ``ws``: advmix_conv4x4s2_wino_ws_floats() floats of scratch, free again
// when the launches have run.  ADVMIX_EINVAL (nothing launched) for shapes not served: the caller runs advmix_conv_fwd.
// Semantics: lib/models/Unet_generator.py:60-62 (downconv) and the input gradient of :63-65 / :74-76 / :84-86 (upconv).
extern "C" int advmix_conv4x4s2_wino_fwd(const float* x, const float* u, const float* bias, float* y, float* ws, int64_t ws_floats,
                                         int N, int H, int W, int Ci, int Co, void* stream) {
    if (!x || !u || !y || !ws || !w4_shape_ok(N, H, W, Ci, Co)) return ADVMIX_EINVAL;
    if (ws_floats < advmix_conv4x4s2_wino_ws_floats(N, H, W, Ci, Co)) return ADVMIX_EINVAL;
    w4::W4P p{};
    p.x = x; p.bias = bias; p.y = y;
    p.N = N; p.H = H; p.W = W; p.Ci = Ci; p.Co = Co;
    p.Th = cdiv(H / 2, 3); p.Tw = cdiv(W / 2, 3);
    p.tiles = N * p.Th * p.Tw;
    p.rows = (p.tiles + 127) / 128 * 128;
    p.xbytes = (int)((int64_t)N * H * W * Ci * 4);
    p.v = ws;
    float* const m = ws + (int64_t)16 * p.rows * 4 * Ci;
    p.m = m;
    hipStream_t st = (hipStream_t)stream;
    if (advmix_conv_direct_gemm_batched(nullptr, nullptr, nullptr, 16, p.rows, 4 * Ci, Co, st) != 0)
        return ADVMIX_EINVAL;                               // (asked BEFORE the input transform goes out: "nothing launched" holds)
    hipLaunchKernelGGL(w4::wino4_input, dim3(cdiv((int64_t)p.rows * Ci, 256)), dim3(256), 0, st, p);
    ADVMIX_CHECK_LAUNCH();
    int rc = advmix_conv_direct_gemm_batched(p.v, u, m, 16, p.rows, 4 * Ci, Co, st);
    if (rc != ADVMIX_OK) return rc < 0 ? ADVMIX_EINVAL : rc;
    hipLaunchKernelGGL(w4::wino4_output, dim3(cdiv((int64_t)p.tiles * (Co / 4), 256)), dim3(256), 0, st, p);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// floats of scratch advmix_conv4x4s2_wino_wgrad needs (V unless handed in, dM, dU); 0 = shape not served (as the forward's, and
// Cl a multiple of 64: the grouped weight-gradient kernel's row tile)
extern "C" int64_t advmix_conv4x4s2_wino_wgrad_ws_floats(int N, int H, int W, int Ch, int Cl, int have_v) {
    if (!w4_shape_ok(N, H, W, Ch, Cl) || Cl % 64 != 0 || (4 * Ch) % 128 != 0) return 0;
    const int64_t tiles = (int64_t)N * cdiv(H / 2, 3) * cdiv(W / 2, 3);
    const int64_t rows = (tiles + 127) / 128 * 128;
    return 16 * rows * ((have_v ? 0 : 4 * (int64_t)Ch) + Cl) + (int64_t)16 * Cl * 4 * Ch;
}

// Weight gradient of a 4x4 / stride 2 / pad 1 conv with filters [Cl][4][4][Ch], ACCUMULATED into dw:
// dw[cl][ky][kx][ch] += sum lo[n][oy][ox][cl] * hi[n][2 oy + ky - 1][2 ox + kx - 1][ch]; hi [N][H][W][Ch] is the conv's input (a
// transposed conv's output gradient), lo [N][H/2][W/2][Cl] its output gradient (a transposed conv's input).  ``v``: the input
// transform of ``hi`` if a preceding advmix_conv4x4s2_wino_fwd(hi, ...) left it at the start of ITS scratch (same stream), else
// null (made here).  ADVMIX_EINVAL (nothing launched) for unserved shapes and in deterministic mode.
// Replaces autograd's convolution_backward weight path for lib/models/Unet_generator.py:60-65,74-76,84-86.
extern "C" int advmix_conv4x4s2_wino_wgrad(const float* hi, const float* lo, float* dw, const float* v, float* ws, int64_t ws_floats,
                                           int N, int H, int W, int Ch, int Cl, void* stream) {
    if (!hi || !lo || !dw || !ws || advmix_opts().deterministic) return ADVMIX_EINVAL;
    const int64_t need = advmix_conv4x4s2_wino_wgrad_ws_floats(N, H, W, Ch, Cl, v != nullptr);
    if (need <= 0 || ws_floats < need) return ADVMIX_EINVAL;
    w4::W4P p{};
    p.x = hi;
    p.N = N; p.H = H; p.W = W; p.Ci = Ch; p.Co = Cl;
    p.Th = cdiv(H / 2, 3); p.Tw = cdiv(W / 2, 3);
    p.tiles = N * p.Th * p.Tw;
    p.rows = (p.tiles + 127) / 128 * 128;
    p.xbytes = (int)((int64_t)N * H * W * Ch * 4);
    hipStream_t st = (hipStream_t)stream;
    const int K = 4 * Ch;
    float* cur = ws;
    if (!v) {
        p.v = cur;
        cur += (int64_t)16 * p.rows * K;
        hipLaunchKernelGGL(w4::wino4_input, dim3(cdiv((int64_t)p.rows * Ch, 256)), dim3(256), 0, st, p);
        ADVMIX_CHECK_LAUNCH();
        v = p.v;
    }
    float* const dm = cur;
    float* const du = dm + (int64_t)16 * p.rows * Cl;
    hipLaunchKernelGGL(w4::wino4_output_adj, dim3(cdiv((int64_t)p.rows * (Cl / 4), 256)), dim3(256), 0, st, lo, dm, N, H / 2, W / 2, Cl,
                       p.Th, p.Tw, p.tiles, p.rows);
    ADVMIX_CHECK_LAUNCH();
    if (hipMemsetAsync(du, 0, (size_t)16 * Cl * K * sizeof(float), st) != hipSuccess) return ADVMIX_ELAUNCH;
    const float* a[16];
    const float* b[16];
    float* d[16];
    for (int xi = 0; xi < 16; ++xi) {
        a[xi] = dm + (int64_t)xi * p.rows * Cl;
        b[xi] = v + (int64_t)xi * p.rows * K;
        d[xi] = du + (int64_t)xi * Cl * K;
    }
    int rc = advmix_conv_wgrad_group(16, a, b, d, p.rows / 128, 8, 16, Cl, 8, 16, K, 1, 1, 1, 0, stream);
    if (rc != ADVMIX_OK) return rc;
    hipLaunchKernelGGL(w4::wino4_wgrad_out, dim3(cdiv((int64_t)Cl * Ch, 256)), dim3(256), 0, st, du, dw, Cl, Ch);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

static bool w4t_shape_ok(int N, int Hl, int Wl, int Cl, int Ch) {
    if (N <= 0 || Hl < 1 || Wl < 1 || Cl % 32 != 0 || Ch % 32 != 0) return false;
    const int64_t tiles = (int64_t)N * (Hl / 3 + 1) * (Wl / 3 + 1);
    const int64_t rows = (tiles + 127) / 128 * 128;
    if ((int64_t)N * Hl * Wl * Cl * 4 >= 0x7fffffffLL || (int64_t)N * Hl * Wl * Ch * 16 >= 0x7fffffffLL || 16 * rows * Cl * 4 >= 0x7fffffffLL ||
        16 * rows * 4 * Ch * 4 >= 0x7fffffffLL || (int64_t)16 * Cl * 4 * Ch * 4 >= 0x7fffffffLL)
        return false;
    return true;
}

// floats of scratch advmix_deconv4x4s2_wino_fwd needs; 0 = shape not served (served: Cl, Ch multiples of 32, every buffer below 2 GiB)
extern "C" int64_t advmix_deconv4x4s2_wino_ws_floats(int N, int Hl, int Wl, int Cl, int Ch) {
    if (!w4t_shape_ok(N, Hl, Wl, Cl, Ch)) return 0;
    const int64_t tiles = (int64_t)N * (Hl / 3 + 1) * (Wl / 3 + 1);
    const int64_t rows = (tiles + 127) / 128 * 128;
    return 16 * rows * ((int64_t)Cl + 4 * Ch);
}

// Transposed-form filter images of n banks [Cl][4][4][Ch] in one launch (records as advmix_w4_weights with role 1; a record owns
// (Cl / 32) * (Ch / 32) * 4 workgroups); u'[xi][(P, Q), ch][cl], 16 * 4 * Ch * Cl floats (advmix_wino4_u_floats(Cl, Ch)).
extern "C" int advmix_w4t_weights(const void* ents, const int* blk_ent, int blocks, void* stream) {
    if (!ents || !blk_ent || blocks <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(w4::wino4t_weights, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const w4::W4Ent*)ents, blk_ent);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// advmix_conv_tr_w / advmix_conv_tr_w_add for k 4 / stride 2 / pad 1 with filters w[Cl][4][4][Ch] transformed by advmix_w4t_weights
// into ``u``: y[N][2 Hl][2 Wl][Ch] = conv_transpose(x[N][Hl][Wl][Cl]) + bias + addend.  ADVMIX_EINVAL (nothing launched) for unserved
// shapes.  Semantics: lib/models/Unet_generator.py:63-65,74-76,84-86 (upconv) and the input gradient of :60-62 (downconv).
extern "C" int advmix_deconv4x4s2_wino_fwd(const float* x, const float* u, const float* bias, const float* addend, float* y, float* ws,
                                           int64_t ws_floats, int N, int Hl, int Wl, int Cl, int Ch, void* stream) {
    if (!x || !u || !y || !ws) return ADVMIX_EINVAL;
    const int64_t need = advmix_deconv4x4s2_wino_ws_floats(N, Hl, Wl, Cl, Ch);
    if (need <= 0 || ws_floats < need) return ADVMIX_EINVAL;
    w4::W4T p{};
    p.x = x; p.bias = bias; p.add = addend; p.y = y;
    p.N = N; p.Hl = Hl; p.Wl = Wl; p.Cl = Cl; p.Ch = Ch;
    p.Th = Hl / 3 + 1; p.Tw = Wl / 3 + 1;
    p.tiles = N * p.Th * p.Tw;
    p.rows = (p.tiles + 127) / 128 * 128;
    p.xbytes = (int)((int64_t)N * Hl * Wl * Cl * 4);
    p.v = ws;
    float* const m = ws + (int64_t)16 * p.rows * Cl;
    p.m = m;
    hipStream_t st = (hipStream_t)stream;
    if (advmix_conv_direct_gemm_batched(nullptr, nullptr, nullptr, 16, p.rows, Cl, 4 * Ch, st) != 0)
        return ADVMIX_EINVAL;                               // (asked BEFORE the input transform goes out: "nothing launched" holds)
    hipLaunchKernelGGL(w4::wino4t_input, dim3(cdiv((int64_t)p.rows * (Cl / 4), 256)), dim3(256), 0, st, p);
    ADVMIX_CHECK_LAUNCH();
    int rc = advmix_conv_direct_gemm_batched(p.v, u, m, 16, p.rows, Cl, 4 * Ch, st);
    if (rc != ADVMIX_OK) return rc < 0 ? ADVMIX_EINVAL : rc;
    hipLaunchKernelGGL(w4::wino4t_output, dim3(cdiv((int64_t)p.tiles * Ch, 256)), dim3(256), 0, st, p);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}
