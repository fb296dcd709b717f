// Convolution family for gfx950 as implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, bit-for-bit an fmaf chain).
//
//   conv_igemm<MODE=0>  forward gather   : Conv2d fwd, ConvTranspose2d dgrad
//   conv_igemm<MODE=1>  transposed gather: Conv2d dgrad, ConvTranspose2d fwd
//                       (phase-decomposed by output parity: only real taps are multiplied)
//   conv_wgrad          weight gradient for both, pixels are the reduction axis
//
// GEMM view (fwd): rows = output pixels, cols = output channels, K = taps x input channels.
// Both operands are K-contiguous in HBM (NHWC activations, [Cout][R][S][Cin] weights), so a
// lane's 16-byte load is 4 consecutive k of one row; tiles are staged through LDS in
// [row][16+4] layout (80-byte rows: conflict-free ds_read_b128 per the 16-lane groups) and
// each ds_read_b128 feeds four MFMAs (k-pairs {j, j+4} of an 8-wide k group).
//
// Replaces (reference): nn.Conv2d / nn.ConvTranspose2d in lib/models/pose_hrnet.py,
// lib/models/pose_resnet.py, lib/models/Unet_generator.py (see include/advmix_hip.h).
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int BK = 16;      // k per LDS tile
constexpr int LDT = 20;     // LDS row pitch in floats (16 + 4 pad)

struct ConvP {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    int N, Hi, Wi, Ci;      // gathered tensor
    int Ho, Wo, Co;         // produced tensor
    int R, S, stride, pad;
};

template <int BM, int BN, int MODE, bool VEC>
__global__ __launch_bounds__(256) void conv_igemm(ConvP p) {
    constexpr int WAVES_N = BN / 32;
    constexpr int WAVES_M = 4 / WAVES_N;
    constexpr int TM = BM / (32 * WAVES_M);
    constexpr int AROWS = BM / 64;                    // A rows staged per thread
    constexpr int BROWS = (BN + 63) / 64;             // B rows staged per thread
    static_assert(TM >= 1 && AROWS >= 1, "tile");

    __shared__ __attribute__((aligned(16))) float As[BM * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];
    __shared__ int4 taptab[64];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;

    // ---- phase geometry -------------------------------------------------------------
    int Hp, Wp, Th, Tw, rh = 0, rw = 0, phh = 0, phw = 0;
    if (MODE == 0) {
        Hp = p.Ho; Wp = p.Wo; Th = p.R; Tw = p.S;
    } else {
        rh = blockIdx.z / p.stride; rw = blockIdx.z % p.stride;
        Hp = p.Ho > rh ? (p.Ho - rh + p.stride - 1) / p.stride : 0;
        Wp = p.Wo > rw ? (p.Wo - rw + p.stride - 1) / p.stride : 0;
        phh = (rh + p.pad) % p.stride; phw = (rw + p.pad) % p.stride;
        Th = phh < p.R ? (p.R - phh + p.stride - 1) / p.stride : 0;
        Tw = phw < p.S ? (p.S - phw + p.stride - 1) / p.stride : 0;
    }
    const int Mp = p.N * Hp * Wp;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    if (m0 >= Mp) return;
    const int ntaps = Th * Tw;
    const int Ktot = ntaps * p.Ci;
    const int nkt = (Ktot + BK - 1) / BK;
    const int Kfull = p.R * p.S * p.Ci;

    if (tid < ntaps) {
        int4 t;
        if (MODE == 0) {
            t.x = tid / p.S; t.y = tid % p.S; t.z = tid * p.Ci;
        } else {
            int th = tid / Tw, tw = tid % Tw;
            int r = phh + p.stride * th, s = phw + p.stride * tw;
            t.x = -th; t.y = -tw; t.z = (r * p.S + s) * p.Ci;
        }
        t.w = 0;
        taptab[tid] = t;
    }

    // ---- per-thread staging rows -----------------------------------------------------
    const int kc = tid & 3;                            // which float4 of the 16-wide k tile
    const int srow = tid >> 2;                         // 0..63
    int a_nb[AROWS], a_h[AROWS], a_w[AROWS];
    bool a_ok[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        int m = m0 + srow + 64 * i;
        a_ok[i] = m < Mp;
        int mm = a_ok[i] ? m : 0;
        int n = mm / (Hp * Wp);
        int rem = mm - n * (Hp * Wp);
        int hi_ = rem / Wp, wi_ = rem - hi_ * Wp;
        a_nb[i] = n * p.Hi * p.Wi;
        if (MODE == 0) {
            a_h[i] = hi_ * p.stride - p.pad;
            a_w[i] = wi_ * p.stride - p.pad;
        } else {
            a_h[i] = (rh + hi_ * p.stride + p.pad - phh) / p.stride;
            a_w[i] = (rw + wi_ * p.stride + p.pad - phw) / p.stride;
        }
    }
    bool b_ok[BROWS];
    const float* b_ptr[BROWS];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
        int n = n0 + srow + 64 * i;
        b_ok[i] = (srow + 64 * i < BN) && n < p.Co;
        b_ptr[i] = p.w + (int64_t)(b_ok[i] ? n : 0) * Kfull;
    }
    __syncthreads();                                   // taptab visible

    // k tracker for this thread's float4 slot (VEC path)
    int kt_tap = (kc * 4) / p.Ci;
    int kt_c = (kc * 4) - kt_tap * p.Ci;

    f32x4 ra[AROWS], rb[BROWS];
    auto load_tile = [&](int kt) {
        if (VEC) {
            const bool tv = kt_tap < ntaps;
            int4 tt = taptab[tv ? kt_tap : 0];
#pragma unroll
            for (int i = 0; i < AROWS; ++i) {
                int hi = a_h[i] + tt.x, wi = a_w[i] + tt.y;
                bool ok = tv && a_ok[i] && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
                ra[i] = ok ? *reinterpret_cast<const f32x4*>(
                                 p.x + (int64_t)(a_nb[i] + hi * p.Wi + wi) * p.Ci + kt_c)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < BROWS; ++i)
                rb[i] = (tv && b_ok[i]) ? *reinterpret_cast<const f32x4*>(b_ptr[i] + tt.z + kt_c)
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int kf = kt * BK + kc * 4 + e;
                int tap = kf / p.Ci;
                int c = kf - tap * p.Ci;
                const bool tv = tap < ntaps;
                int4 tt = taptab[tv ? tap : 0];
#pragma unroll
                for (int i = 0; i < AROWS; ++i) {
                    int hi = a_h[i] + tt.x, wi = a_w[i] + tt.y;
                    bool ok = tv && a_ok[i] && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
                    ra[i][e] = ok ? p.x[(int64_t)(a_nb[i] + hi * p.Wi + wi) * p.Ci + c] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < BROWS; ++i) rb[i][e] = (tv && b_ok[i]) ? b_ptr[i][tt.z + c] : 0.f;
            }
        }
    };
    auto advance = [&]() {
        kt_c += BK;
        while (kt_c >= p.Ci) { kt_c -= p.Ci; ++kt_tap; }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < AROWS; ++i)
            *reinterpret_cast<f32x4*>(&As[(srow + 64 * i) * LDT + kc * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BROWS; ++i)
            if (srow + 64 * i < BN) *reinterpret_cast<f32x4*>(&Bs[(srow + 64 * i) * LDT + kc * 4]) = rb[i];
    };

    f32x16 acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    load_tile(0);
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) { advance(); load_tile(kt + 1); }
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
            f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[(wn * 32 + l31) * LDT + kq * 8 + lh * 4]);
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                f32x4 a = *reinterpret_cast<const f32x4*>(
                    &As[((wm * TM + t) * 32 + l31) * LDT + kq * 8 + lh * 4]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) { store_tile(); __syncthreads(); }
    }

    // ---- epilogue: lane holds column l31, rows (r&3)+8*(r>>2)+4*lh of each 32x32 tile ------
    const int col = n0 + wn * 32 + l31;
    if (col >= p.Co) return;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = m0 + (wm * TM + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= Mp) continue;
            int64_t off;
            if (MODE == 0) {
                off = (int64_t)m * p.Co + col;
            } else {
                int n = m / (Hp * Wp);
                int rem = m - n * (Hp * Wp);
                int hi_ = rem / Wp, wi_ = rem - hi_ * Wp;
                off = ((int64_t)(n * p.Ho + rh + hi_ * p.stride) * p.Wo + rw + wi_ * p.stride) * p.Co + col;
            }
            p.y[off] = acc[t][r] + bv;
        }
    }
}

// -------------------------------------------------------------------------------------------
// weight gradient: dw[Ca][R*S*Cb] += A^T[Ca][P] * Bg[P][R*S*Cb],  P = N*Ha*Wa pixels
// -------------------------------------------------------------------------------------------
struct WgP {
    const float* a;
    const float* b;
    float* dw;
    int N, Ha, Wa, Ca, Hb, Wb, Cb;
    int R, S, stride, pad;
    int chunk;               // pixels per z-slice (multiple of 32)
    float* part;             // deterministic mode: slice z STORES its partial tile at part[z * Ca * Ntot + ...] instead of
                             // adding it to dw with fp32 atomics; reduce_slices_kernel sums the slices in order
    int excl;                // one slice per tile (a large group): the workgroup owns its outputs, dw += with plain accesses
    int xcd;                 // 1: workgroup order remapped so that the tiles of one pixel slice run on ONE XCD (xcd_order)
};

// Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest), so the column / row tiles of one pixel slice -
// which all read the SAME dy and x pixels - land on 8 different L2s and each fetches them from the fabric again (3x3
// 128 -> 128: 9 column tiles = 9 reads of every operand byte; 3.2 TB/s at the rate the kernel runs).  Remapped, XCD k works
// through the contiguous range [k T / 8, (k + 1) T / 8) of (slice, row tile, column tile) in order: the tiles of a slice are
// neighbours in time on one L2.  Bijective for any grid (the first T % 8 XCDs hold one workgroup more).
__device__ __forceinline__ void xcd_order(int& bx, int& by, int& bz) {
    const unsigned gx = gridDim.x, gy = gridDim.y, T = gx * gy * gridDim.z;
    const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned q = T >> 3, rem = T & 7, xcd = L & 7, k = L >> 3;
    const unsigned v = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
    bx = (int)(v % gx);
    by = (int)((v / gx) % gy);
    bz = (int)(v / (gx * gy));
}

// dw[i] += the slices' partials in a FIXED order: the deterministic tail of the weight / bias gradients.  Scalar form (any
// total): one thread per output walks the slices in order.
__global__ __launch_bounds__(256) void reduce_slices_kernel(const float* __restrict__ part, int nslices, int64_t total,
                                                            float* __restrict__ dw) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nslices; ++k) s += part[(int64_t)k * total + i];
        dw[i] += s;
    }
}

// 16-byte form (total % 4 == 0): a workgroup = G waves on the SAME 64 float4 columns; wave g adds slices g, g + G, ... in
// that order (eight independent 16-byte loads in flight per lane), then the G partial sums meet in LDS and are added in
// wave order - a fixed association for a given (nslices, G), so still bit-reproducible, and 4 G times as many loads in
// flight per output as the scalar form (256 slabs of a 9,216-element gradient: 36 threads-blocks walking 256 dependent
// steps took ~10 us; round 3).
template <int G>
__global__ __launch_bounds__(64 * G) void reduce_slices4_kernel(const float* __restrict__ part, int nslices, int64_t total4,
                                                                float* __restrict__ dw) {
    __shared__ f32x4 red[G][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t i4 = blockIdx.x * 64LL + lane;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i4 < total4) {
        const f32x4* src = reinterpret_cast<const f32x4*>(part) + i4;
        int k = g;
        for (; k + 7 * G < nslices; k += 8 * G) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + u * G) * total4];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nslices; k += G) s += src[(int64_t)k * total4];
    }
    red[g][lane] = s;
    __syncthreads();
    if (g == 0 && i4 < total4) {
#pragma unroll
        for (int j = 1; j < G; ++j) s += red[j][lane];
        f32x4* d = reinterpret_cast<f32x4*>(dw) + i4;
        *d = *d + s;
    }
}

static void launch_reduce_slices(const float* part, int nslices, int64_t total, float* dw, hipStream_t st) {
    if (total % 4 == 0 && nslices >= 4) {
        const int64_t t4 = total / 4;
        const dim3 g((unsigned)((t4 + 63) / 64));
        // enough waves for the chip (>= ~1,024) without leaving a wave fewer than two slices
        const int64_t want = 1024 / (int64_t)g.x;
        if (want >= 16 && nslices >= 32) hipLaunchKernelGGL(reduce_slices4_kernel<16>, g, dim3(1024), 0, st, part, nslices, t4, dw);
        else if (want >= 8 && nslices >= 16) hipLaunchKernelGGL(reduce_slices4_kernel<8>, g, dim3(512), 0, st, part, nslices, t4, dw);
        else if (want >= 4 && nslices >= 8) hipLaunchKernelGGL(reduce_slices4_kernel<4>, g, dim3(256), 0, st, part, nslices, t4, dw);
        else if (want >= 2) hipLaunchKernelGGL(reduce_slices4_kernel<2>, g, dim3(128), 0, st, part, nslices, t4, dw);
        else hipLaunchKernelGGL(reduce_slices4_kernel<1>, g, dim3(64), 0, st, part, nslices, t4, dw);
        return;
    }
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(reduce_slices_kernel, dim3(blocks > 2048 ? 2048 : blocks), dim3(256), 0, st, part, nslices, total, dw);
}

// TM x TN 32x32 tiles per wave (default 1 x 1): the U-Net's weight gradients (hundreds of output channels x thousands of
// tap-channels) take 128 x 128 per workgroup - one LDS read per MFMA instead of two, half the staging per FLOP.
// Several weight gradients of ONE geometry in one launch (round 4): weight gradients have no consumer before the optimizer
// step, so the eight 3x3 C -> C convs of an HRNet branch (four residual blocks, one launch chain) hand their (dY, X, dW)
// triples to a single launch at the end of the chain's backward - eight times the work per launch means an eighth of the
// pixel slices per problem (fewer partial tiles to merge with atomics, longer main loops per workgroup) and room for the
// 128 x 128 tile (one LDS read per MFMA instead of two).  blockIdx.z = problem * slices + slice.
constexpr int WG_MAXG = 64;
struct WgGroup {
    int n, slices;
    const float* a[WG_MAXG];
    const float* b[WG_MAXG];
    float* dw[WG_MAXG];
};

template <int WM, int WN, bool VEC, int TM = 1, int TN = 1>
__device__ __forceinline__ void wgrad_body(const WgP p, const int bx, const int by, const int zslice, float* const As0,
                                           float* const Bs0) {
    constexpr int BMw = 32 * WM * TM, BNw = 32 * WN * TN;
    constexpr int KS = 32;                             // pixels per step (16 MFMAs per wave between barriers)
    constexpr int ASL = (KS * BMw / 4 + 255) / 256;    // float4 slots per thread
    constexpr int BSL = (KS * BNw / 4 + 255) / 256;
    float (*const As)[KS * BMw] = reinterpret_cast<float (*)[KS * BMw]>(As0);   // double-buffered: one barrier per step
    float (*const Bs)[KS * BNw] = reinterpret_cast<float (*)[KS * BNw]>(Bs0);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;
    const int co0 = bx * BMw, j0 = by * BNw;
    const int P = p.N * p.Ha * p.Wa;
    const int Ntot = p.R * p.S * p.Cb;
    const int p_lo = zslice * p.chunk;
    const int p_hi = min(P, p_lo + p.chunk);
    if (p_lo >= P) return;

    // fixed per-thread slot geometry
    int a_k[ASL], a_c[ASL];
    bool a_in[ASL];
#pragma unroll
    for (int i = 0; i < ASL; ++i) {
        int s = tid + 256 * i;
        a_in[i] = s < KS * BMw / 4;
        a_k[i] = s / (BMw / 4);
        a_c[i] = (s % (BMw / 4)) * 4;
    }
    int b_k[BSL], b_c[BSL], b_dh[BSL][VEC ? 1 : 4], b_dw[BSL][VEC ? 1 : 4], b_cb[BSL][VEC ? 1 : 4];
    bool b_in[BSL], b_jv[BSL][VEC ? 1 : 4];
#pragma unroll
    for (int i = 0; i < BSL; ++i) {
        int s = tid + 256 * i;
        b_in[i] = s < KS * BNw / 4;
        b_k[i] = s / (BNw / 4);
        b_c[i] = (s % (BNw / 4)) * 4;
#pragma unroll
        for (int e = 0; e < (VEC ? 1 : 4); ++e) {
            int j = j0 + b_c[i] + e;
            b_jv[i][e] = j < Ntot;
            int jj = b_jv[i][e] ? j : 0;
            int tap = jj / p.Cb;
            b_cb[i][e] = jj - tap * p.Cb;
            int r = tap / p.S;
            b_dh[i][e] = r - p.pad;
            b_dw[i][e] = tap - r * p.S - p.pad;
        }
    }

    // Each B slot's pixel advances by KS per step: (n, ha, wa) are carried along instead of being divided out of the pixel
    // index on every step (two integer divisions per slot and step were ~600 VALU cycles per 1024 MFMA cycles and wave).
    const int HWa = p.Ha * p.Wa;
    const int step_n = KS / HWa;                                         // KS pixels = step_n images + step_h rows +
    const int step_h = (KS - step_n * HWa) / p.Wa;                       // step_w columns (one carry each: branch-free)
    const int step_w = KS - step_n * HWa - step_h * p.Wa;
    int b_n[BSL], b_ha[BSL], b_wa[BSL];
#pragma unroll
    for (int i = 0; i < BSL; ++i) {
        const int pix = p_lo + b_k[i];
        b_n[i] = pix / HWa;
        const int rem = pix - b_n[i] * HWa;
        b_ha[i] = rem / p.Wa;
        b_wa[i] = rem - b_ha[i] * p.Wa;
    }
    f32x4 ra[ASL], rb[BSL];
    // VEC (every channel count a multiple of 4): branch-free buffer loads.  The pointer form below costs ~1,100 cycles of
    // address arithmetic per step and wave (64-bit multiply-adds, eight exec-masked branches) in basic blocks of its own, which
    // the scheduler cannot move under the step's 4,096 cycles of MFMA - and two workgroups per CU fall into lockstep, so
    // the matrix pipe idles through both address sections.  Here a slot carries a 32-bit byte offset (relative to the slice's first
    // pixel / first image) from step to step - three adds and two selects - and an out-of-range slot loads from offset 2^31,
    // past the resource's end, which returns zeros.
    constexpr unsigned WOOB = 0x80000000u;
    __amdgpu_buffer_rsrc_t a_rs, b_rs;
    unsigned a_off[ASL], b_off[BSL];
    int b_tap[BSL], b_dh0[BSL], b_dw0[BSL];
    bool a_cv[ASL], b_ok[BSL];
    int bD0 = 0, bDW = 0, bDH = 0;
    if constexpr (VEC) {
        const int n0 = p_lo / HWa;
        const int64_t a_left = ((int64_t)P - p_lo) * p.Ca * 4;
        const int64_t b_left = ((int64_t)p.N - n0) * p.Hb * p.Wb * p.Cb * 4;
        a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a + (int64_t)p_lo * p.Ca), 0,
                                                 (int)(a_left < 0x7fffffffLL ? a_left : 0x7fffffffLL), 0x00020000);
        b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.b + (int64_t)n0 * p.Hb * p.Wb * p.Cb), 0,
                                                 (int)(b_left < 0x7fffffffLL ? b_left : 0x7fffffffLL), 0x00020000);
#pragma unroll
        for (int i = 0; i < ASL; ++i) {
            a_off[i] = (unsigned)((a_k[i] * p.Ca + co0 + a_c[i]) * 4);
            a_cv[i] = a_in[i] && co0 + a_c[i] < p.Ca;
        }
#pragma unroll
        for (int i = 0; i < BSL; ++i) {
            b_off[i] = (unsigned)((((b_n[i] - n0) * p.Hb + b_ha[i] * p.stride) * p.Wb + b_wa[i] * p.stride) * p.Cb * 4);
            b_tap[i] = ((b_dh[i][0] * p.Wb + b_dw[i][0]) * p.Cb + b_cb[i][0]) * 4;
            b_dh0[i] = b_dh[i][0];
            b_dw0[i] = b_dw[i][0];
            b_ok[i] = b_in[i] && b_jv[i][0];
        }
        bD0 = ((step_n * p.Hb + step_h * p.stride) * p.Wb + step_w * p.stride) * p.Cb * 4;
        bDW = (p.stride * p.Wb - p.Wa * p.stride) * p.Cb * 4;        // the column wrapped: wa -= Wa, ha += 1
        bDH = (p.Hb * p.Wb - p.Ha * p.stride * p.Wb) * p.Cb * 4;     // the row wrapped: ha -= Ha, n += 1
    }
    // The offsets of the tile after the one being loaded are formed under the MFMAs of the current step (advance_v after the
    // loads have been issued): the top of a step is eight buffer loads and nothing else.
    unsigned a_vo[ASL], b_vo[BSL];
    auto advance_v = [&](int pt, bool first) {            // -> a_vo / b_vo of tile pt (first: the state IS tile pt's)
#pragma unroll
        for (int i = 0; i < ASL; ++i) {
            if (!first) a_off[i] += (unsigned)(KS * p.Ca * 4);
            const bool ok = a_cv[i] & (pt + a_k[i] < p_hi);
            a_vo[i] = ok ? a_off[i] : WOOB;
        }
#pragma unroll
        for (int i = 0; i < BSL; ++i) {
            if (!first) {
                const int w2 = b_wa[i] + step_w, cw = w2 >= p.Wa;
                const int h2 = b_ha[i] + step_h + cw, ch = h2 >= p.Ha;
                b_wa[i] = w2 - (cw ? p.Wa : 0);
                b_ha[i] = h2 - (ch ? p.Ha : 0);
                b_off[i] += (unsigned)(bD0 + (cw ? bDW : 0) + (ch ? bDH : 0));
            }
            const int hb = __mul24(b_ha[i], p.stride) + b_dh0[i], wb = __mul24(b_wa[i], p.stride) + b_dw0[i];
            const bool v = b_ok[i] & (pt + b_k[i] < p_hi) & ((unsigned)hb < (unsigned)p.Hb) & ((unsigned)wb < (unsigned)p.Wb);   // (no short circuit: no branch)
            b_vo[i] = v ? b_off[i] + (unsigned)b_tap[i] : WOOB;
        }
    };
    auto load_tile_v = [&]() {
#pragma unroll
        for (int i = 0; i < ASL; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo[i], 0, 0));
#pragma unroll
        for (int i = 0; i < BSL; ++i)
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_vo[i], 0, 0));
    };
    auto load_tile_p = [&](int pt) {
#pragma unroll
        for (int i = 0; i < ASL; ++i) {
            int pix = pt + a_k[i];
            bool ok = a_in[i] && pix < p_hi;
            const float* src = p.a + (int64_t)(ok ? pix : 0) * p.Ca + co0 + a_c[i];
            if (VEC) {
                ra[i] = (ok && co0 + a_c[i] < p.Ca) ? *reinterpret_cast<const f32x4*>(src)
                                                   : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[i][e] = (ok && co0 + a_c[i] + e < p.Ca) ? src[e] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < BSL; ++i) {
            int pix = pt + b_k[i];
            bool ok = b_in[i] && pix < p_hi;
            const int n = b_n[i], ha = b_ha[i], wa = b_wa[i];           // of pixel pt + b_k[i] (load_tile is called with
            int hb0 = ha * p.stride, wb0 = wa * p.stride;              // pt = p_lo, p_lo + KS, ... in order)
            const int w2 = wa + step_w, cw = w2 >= p.Wa;
            const int h2 = ha + step_h + cw, ch = h2 >= p.Ha;
            b_wa[i] = w2 - (cw ? p.Wa : 0);
            b_ha[i] = h2 - (ch ? p.Ha : 0);
            b_n[i] = n + step_n + ch;
            if (VEC) {
                int hb = hb0 + b_dh[i][0], wb = wb0 + b_dw[i][0];
                bool v = ok && b_jv[i][0] && (unsigned)hb < (unsigned)p.Hb && (unsigned)wb < (unsigned)p.Wb;
                rb[i] = v ? *reinterpret_cast<const f32x4*>(
                                p.b + ((int64_t)(n * p.Hb + hb) * p.Wb + wb) * p.Cb + b_cb[i][0])
                          : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int hb = hb0 + b_dh[i][e], wb = wb0 + b_dw[i][e];
                    bool v = ok && b_jv[i][e] && (unsigned)hb < (unsigned)p.Hb && (unsigned)wb < (unsigned)p.Wb;
                    rb[i][e] = v ? p.b[((int64_t)(n * p.Hb + hb) * p.Wb + wb) * p.Cb + b_cb[i][e]] : 0.f;
                }
            }
        }
    };
    constexpr bool A_FULL = (KS * BMw / 4) % 256 == 0, B_FULL = (KS * BNw / 4) % 256 == 0;    // every thread owns ASL / BSL slots
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ASL; ++i)
            if (A_FULL || a_in[i]) *reinterpret_cast<f32x4*>(&As[buf][a_k[i] * BMw + a_c[i]]) = ra[i];
#pragma unroll
        for (int i = 0; i < BSL; ++i)
            if (B_FULL || b_in[i]) *reinterpret_cast<f32x4*>(&Bs[buf][b_k[i] * BNw + b_c[i]]) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    // Branch-free steady state (a tile past the slice loads zeros, so the prefetch is unconditional): the first
    // version's "if (more) load" made the compiler shuttle the accumulator AGPR -> VGPR -> AGPR around the branch
    // on every 8-MFMA step, and it paid two barriers per 16 pixels.
    // Register staging with the LDS write AFTER the barrier: tile t + 1 (loaded during step t - 1) is written into the other
    // buffer at the START of step t - its last readers left it before the barrier - and the same registers take the loads of
    // tile t + 2.  A load has a whole step to land before anything waits for it, and the write has one before the barrier that
    // publishes it; written at the END of the step (until round 4) both latencies sat in front of the barrier: 10 % of the
    // kernel (profiles/r04_wgrad_knockouts_x56.log).
    if constexpr (VEC) {
        advance_v(p_lo, true);
        load_tile_v();
        advance_v(p_lo + KS, false);
    } else {
        load_tile_p(p_lo);
    }
    store_tile(0);
    if constexpr (VEC) {
        load_tile_v();
        advance_v(p_lo + 2 * KS, false);
    } else {
        load_tile_p(p_lo + KS);
    }
    __syncthreads();
    int buf = 0;
    for (int pt = p_lo; pt < p_hi; pt += KS) {
        store_tile(buf ^ 1);                               // tile pt + KS
        if constexpr (VEC) load_tile_v();                  // tile pt + 2 KS
        else load_tile_p(pt + 2 * KS);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (VEC) advance_v(pt + 3 * KS, false);
        float av[KS / 2][TM], bv[KS / 2][TN];
#pragma unroll
        for (int kk = 0; kk < KS / 2; ++kk) {
#pragma unroll
            for (int t = 0; t < TM; ++t) av[kk][t] = As[buf][(kk * 2 + lh) * BMw + (wm * TM + t) * 32 + l31];
#pragma unroll
            for (int u = 0; u < TN; ++u) bv[kk][u] = Bs[buf][(kk * 2 + lh) * BNw + (wn * TN + u) * 32 + l31];
        }
#pragma unroll
        for (int kk = 0; kk < KS / 2; ++kk)
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < TN; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][t], bv[kk][u], acc[t][u], 0, 0, 0);
        // The operand reads run TWO k-pairs ahead of the MFMAs that use them.  Left alone the compiler reuses one register set:
        // read, wait, 4 MFMAs, read, wait ... - every k-pair's LDS latency (~100 cycles per 256 of MFMA) in the open.
        // (Asking for the address VALU work to be placed as well - a VALU group per k-pair - makes the solver give up on the whole
        // pattern and fall back to the serial schedule; left alone, the default heuristics interleave it with the MFMAs.)
        {
            constexpr int DSK = (TM == 2 ? 1 : TM) + (TN == 2 ? 1 : TN);     // LDS instructions per k-pair (pairs fuse to read2)
            constexpr int AHEAD = 2;
            __builtin_amdgcn_sched_group_barrier(0x100, DSK * AHEAD, 0);
#pragma unroll
            for (int kk = 0; kk < KS / 2 - AHEAD; ++kk) {
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, DSK, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN * AHEAD, 0);
        }
        __syncthreads();
        buf ^= 1;
    }

#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            const int j = j0 + (wn * TN + u) * 32 + l31;
            if (j >= Ntot) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = co0 + (wm * TM + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co >= p.Ca) continue;
                if (p.part) p.part[((int64_t)zslice * p.Ca + co) * Ntot + j] = acc[t][u][r];
                else if (p.excl) p.dw[(int64_t)co * Ntot + j] += acc[t][u][r];
                else atomicAdd(p.dw + (int64_t)co * Ntot + j, acc[t][u][r]);
            }
        }
}

template <int WM, int WN, bool VEC, int TM = 1, int TN = 1>
__global__ __launch_bounds__(256) void conv_wgrad(WgP p) {
    __shared__ __attribute__((aligned(16))) float As[2 * 32 * 32 * WM * TM];
    __shared__ __attribute__((aligned(16))) float Bs[2 * 32 * 32 * WN * TN];
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd) xcd_order(bx, by, bz);
    wgrad_body<WM, WN, VEC, TM, TN>(p, bx, by, bz, As, Bs);
}

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_wgrad_group(WgP p, WgGroup g) {
    __shared__ __attribute__((aligned(16))) float As[2 * 32 * 32 * WM * TM];
    __shared__ __attribute__((aligned(16))) float Bs[2 * 32 * 32 * WN * TN];
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd) xcd_order(bx, by, bz);
    const int prob = bz / g.slices;
    p.a = g.a[prob];
    p.b = g.b[prob];
    p.dw = g.dw[prob];
    wgrad_body<WM, WN, true, TM, TN>(p, bx, by, bz - prob * g.slices, As, Bs);
}

// Up to WG_MULTI weight gradients of DIFFERENT geometries in one launch (round 5): the ~70 small convs of the HRNet fuse
// layers and transitions (3x3 s2 C -> C', 1x1 C -> C': 7-23 us each at 0.02-0.28 of peak - launch latency, a few workgroups)
// have no consumer before the optimizer step either; block b belongs to problem i with start[i] <= b < start[i + 1] and
// computes tile (bx, by) of pixel slice bz of THAT problem with the problem's own tile shape (cfg 0: 64 x 64, 1: 32 x 128
// for Ca <= 32).  Every tile is merged with fp32 atomics (same sums as the single launches, another order).
constexpr int WG_MULTI = 16;
struct WgMulti {
    int n;
    int start[WG_MULTI + 1];
    int gx[WG_MULTI], gy[WG_MULTI], cfg[WG_MULTI];
    WgP p[WG_MULTI];
};

__global__ __launch_bounds__(256) void conv_wgrad_multi(WgMulti g) {
    __shared__ __attribute__((aligned(16))) float As[2 * 32 * 32 * 2];
    __shared__ __attribute__((aligned(16))) float Bs[2 * 32 * 32 * 4];
    const int b = blockIdx.x;
    int i = 0;
#pragma unroll
    for (int k = 1; k < WG_MULTI; ++k)
        if (k < g.n && b >= g.start[k]) i = k;
    const int local = b - g.start[i], gx = g.gx[i], gxy = gx * g.gy[i];
    const int bz = local / gxy, rem = local - bz * gxy;
    const int by = rem / gx, bx = rem - by * gx;
    if (g.cfg[i] == 0) wgrad_body<2, 2, true, 1, 1>(g.p[i], bx, by, bz, As, Bs);
    else wgrad_body<1, 4, true, 1, 1>(g.p[i], bx, by, bz, As, Bs);
}

__global__ void transpose_w_kernel(const float* __restrict__ in, float* __restrict__ out, int A, int T, int B) {
    int64_t total = (int64_t)A * T * B;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int a = (int)(i % A);
        int64_t q = i / A;
        int t = (int)(q % T);
        int b = (int)(q / T);
        out[i] = in[((int64_t)a * T + t) * B + b];
    }
}

// db[c] += sum over rows of dy[rows, C]: 256 threads = RP row-lanes x Cp columns, LDS tree over the
// row-lanes, one atomic per column per block (coalesced 4-byte lanes along c).
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db,
                                                        int64_t rows, int C, int64_t rows_per_block, float* part) {
    __shared__ float red[256];
    const int Cp = C < 256 ? C : 256;
    const int RP = 256 / Cp;
    const int tid = threadIdx.x, rr = tid / Cp, cc = tid - rr * Cp;
    const int64_t r0 = blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    for (int cbase = 0; cbase < C; cbase += Cp) {
        const int c = cbase + cc;
        float s = 0.f;
        if (rr < RP && c < C)
            for (int64_t r = r0 + rr; r < r1; r += RP) s += dy[r * C + c];
        red[tid] = s;
        __syncthreads();
        for (int step = 1; step < RP; step <<= 1) {
            if (rr < RP && (rr % (2 * step)) == 0 && rr + step < RP) red[tid] += red[tid + step * Cp];
            __syncthreads();
        }
        if (rr == 0 && c < C) {
            if (part) part[(int64_t)blockIdx.x * C + c] = red[tid];
            else atomicAdd(db + c, red[tid]);
        }
        __syncthreads();
    }
}

// The same for C % 4 == 0: 16-byte loads, eight rows in flight per thread, and FEWER, larger blocks - the scalar form above is
// bound by its atomics, not by reading dy: 1,024 blocks adding to the same one or two cache lines serialise at the memory side
// (~50 ns per block: 52 us for the U-Net's 101 MB activation, 1.9 TB/s; tools/microbench_unet_aux.py).  The block's sums leave
// through LDS so that ONE wave instruction carries 64 consecutive channels.
__global__ __launch_bounds__(256) void bias_grad_kernel_v4(const float* __restrict__ dy, float* __restrict__ db, int64_t rows, int C,
                                                           int64_t rows_per_block, float* part) {
    __shared__ f32x4 red[256];
    const int CV = C >> 2;
    const int CVp = CV < 256 ? CV : 256;
    const int RP = 256 / CVp;
    const int tid = threadIdx.x, rr = tid / CVp, cc = tid - rr * CVp;
    const int64_t r0 = blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    for (int cbase = 0; cbase < CV; cbase += CVp) {
        const int cv = cbase + cc;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (rr < RP && cv < CV) {
            const float* src = dy + (int64_t)cv * 4;
            const int64_t st = (int64_t)RP * C;
            int64_t r = r0 + rr;
            for (; r + 7 * (int64_t)RP < r1; r += 8 * (int64_t)RP) {
                const float* q = src + r * C;
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(q + k * st);
                s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            for (; r < r1; r += RP) s += *reinterpret_cast<const f32x4*>(src + r * C);
        }
        red[tid] = s;
        __syncthreads();
        for (int step = 1; step < RP; step <<= 1) {
            if (rr < RP && (rr % (2 * step)) == 0 && rr + step < RP) red[tid] += red[tid + step * CVp];
            __syncthreads();
        }
        const float* flat = reinterpret_cast<const float*>(red);       // red[0 .. CVp) = this pass's 4 CVp channel sums, in order
        for (int c = tid; c < 4 * CVp && cbase * 4 + c < C; c += 256) {
            if (part) part[(int64_t)blockIdx.x * C + cbase * 4 + c] = flat[c];
            else atomicAdd(db + cbase * 4 + c, flat[c]);
        }
        __syncthreads();
    }
}

template <int MODE>
int launch_igemm(const ConvP& p, int64_t Mmax, hipStream_t st) {
    const bool vec = (p.Ci % 4 == 0);
    const int phases = MODE == 0 ? 1 : p.stride * p.stride;
#define LAUNCH(BM_, BN_, V_)                                                              \
    do {                                                                                  \
        dim3 g(cdiv(Mmax, BM_), cdiv(p.Co, BN_), phases);                                 \
        hipLaunchKernelGGL((conv_igemm<BM_, BN_, MODE, V_>), g, dim3(256), 0, st, p);     \
    } while (0)
    if (p.Co <= 32) {
        if (vec) LAUNCH(128, 32, true); else LAUNCH(128, 32, false);
    } else {
        int64_t blocks128 = (int64_t)cdiv(Mmax, 128) * cdiv(p.Co, 64) * phases;
        if (blocks128 >= 512) {
            if (vec) LAUNCH(128, 64, true); else LAUNCH(128, 64, false);
        } else {
            if (vec) LAUNCH(64, 64, true); else LAUNCH(64, 64, false);
        }
    }
#undef LAUNCH
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

static bool use_direct() { return advmix_opts().direct != 0; }

}  // namespace

extern "C" int advmix_conv_fwd(const float* x, const float* w, const float* bias, float* y,
                               int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                               int R, int S, int stride, int pad, void* stream) {
    if (!x || !w || !y || N <= 0 || Ci <= 0 || Co <= 0 || R * S > 64 || stride < 1) return ADVMIX_EINVAL;
    if (Ho != (Hi + 2 * pad - R) / stride + 1 || Wo != (Wi + 2 * pad - S) / stride + 1) return ADVMIX_EINVAL;
    if (use_direct()) {
        int rc = advmix_conv_direct_dispatch(0, x, w, bias, y, N, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride, pad,
                                             (int64_t)N * Ho * Wo, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    ConvP p{x, w, bias, y, N, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride, pad};
    return launch_igemm<0>(p, (int64_t)N * Ho * Wo, (hipStream_t)stream);
}

extern "C" int advmix_conv_group(int kind, int n, advmix_conv_problem* pr, void* stream) {
    if (!pr || n < 2 || n > 4 || (kind != 0 && kind != 1) || !use_direct()) return ADVMIX_EINVAL;
    ConvProb q[4];
    ConvEpi e[4];
    for (int i = 0; i < n; ++i) {
        const advmix_conv_problem& a = pr[i];
        if (!a.x || !a.w || !a.y || a.N <= 0 || a.Cx <= 0 || a.Cy <= 0 || a.stride != 1) return ADVMIX_EINVAL;
        if (a.stats && a.stats_ns < 0) return ADVMIX_EINVAL;
        if (kind == 0) {
            if (a.Hy != a.Hx + 2 * a.pad - a.R + 1 || a.Wy != a.Wx + 2 * a.pad - a.S + 1) return ADVMIX_EINVAL;
            if ((a.bn_gamma != nullptr) != (a.bn_beta && a.bn_rm && a.bn_rv)) return ADVMIX_EINVAL;
            const bool has = a.bn_gamma || a.residual || a.act || a.stats;
            e[i] = ConvEpi{a.bn_gamma, a.bn_beta, a.bn_rm, a.bn_rv, a.residual, a.bn_eps, a.act, a.stats,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
            q[i] = ConvProb{a.x, a.w, a.bias, a.y, a.N, a.Hx, a.Wx, a.Cx, a.Hy, a.Wy, a.Cy, a.R, a.S, 1, a.pad,
                            (int64_t)a.N * a.Hy * a.Wy, has ? &e[i] : nullptr, a.stats_ns};
        } else {
            if (a.Hx != a.Hy + 2 * a.pad - a.R + 1 || a.Wx != a.Wy + 2 * a.pad - a.S + 1) return ADVMIX_EINVAL;
            const bool bnb = a.stats != nullptr;
            if (bnb && (!a.bnb_c || !a.bnb_mean || !a.bnb_invstd ||
                        (a.bnb_act != ADVMIX_ACT_NONE && !a.bnb_mask && !(a.bnb_gamma && a.bnb_beta))))
                return ADVMIX_EINVAL;
            e[i] = ConvEpi{nullptr, nullptr, nullptr, nullptr, a.residual, 0.f, 0, bnb ? a.stats : nullptr,
                           bnb ? a.bnb_mask : nullptr, bnb ? a.bnb_c : nullptr, bnb ? a.bnb_mean : nullptr,
                           bnb ? a.bnb_invstd : nullptr, bnb ? a.bnb_gamma : nullptr, bnb ? a.bnb_beta : nullptr,
                           bnb ? a.bnb_act : 0};
            q[i] = ConvProb{a.x, a.w, nullptr, a.y, a.N, a.Hx, a.Wx, a.Cx, a.Hy, a.Wy, a.Cy, a.R, a.S, 1, a.pad,
                            (int64_t)a.N * a.Hy * a.Wy, (bnb || a.residual) ? &e[i] : nullptr, a.stats_ns};
        }
    }
    if (kind == 1) {                                       // addend-only and plain problems share a kernel (EPI = false)
        bool any_bnb = false, all_bnb = true;
        for (int i = 0; i < n; ++i) { const bool b = pr[i].stats != nullptr; any_bnb |= b; all_bnb &= b; }
        if (any_bnb && !all_bnb) return ADVMIX_EINVAL;
    }
    int rc = advmix_conv_direct_group(kind, kind, n, q, (hipStream_t)stream);
    if (rc < 0) return ADVMIX_EINVAL;
    for (int i = 0; i < n; ++i) pr[i].stats_ns = q[i].stats_nbg;
    return rc;
}

extern "C" int advmix_conv_tr(const float* x, const float* wt, const float* bias, float* y,
                              int N, int Hs, int Ws, int Ck, int Hb, int Wb, int Cn,
                              int R, int S, int stride, int pad, void* stream) {
    if (!x || !wt || !y || N <= 0 || Ck <= 0 || Cn <= 0 || R * S > 64 || stride < 1 || stride > 8)
        return ADVMIX_EINVAL;
    // (Hb, Wb) must be a valid input size for a conv producing (Hs, Ws)
    if (Hs != (Hb + 2 * pad - R) / stride + 1 || Ws != (Wb + 2 * pad - S) / stride + 1) return ADVMIX_EINVAL;
    int64_t Mmax = (int64_t)N * cdiv(Hb, stride) * cdiv(Wb, stride);
    if (use_direct()) {
        int rc = advmix_conv_direct_dispatch(1, x, wt, bias, y, N, Hs, Ws, Ck, Hb, Wb, Cn, R, S, stride, pad, Mmax,
                                             (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    ConvP p{x, wt, bias, y, N, Hs, Ws, Ck, Hb, Wb, Cn, R, S, stride, pad};
    return launch_igemm<1>(p, Mmax, (hipStream_t)stream);
}

// workgroups a wgrad launch aims for (output tiles x pixel slices); every slice adds its partial tile with
// fp32 atomics, so this trades parallelism against atomic traffic (ADVMIX_WGRAD_BLOCKS to experiment)
static int wgrad_xcd_order() {          // ADVMIX_WGRAD_XCD=0: launch order as dealt (A/B of xcd_order)
    static const int v = [] { const char* e = getenv("ADVMIX_WGRAD_XCD"); return e ? atoi(e) : 1; }();
    return v;
}

// The vector form of wgrad_body addresses a workgroup's pixel slice with 32-bit byte offsets from the slice's first pixel (a)
// and first image (b); a slice whose span does not fit (a > 2 GB image pair) takes the pointer form.
static bool wgrad_spans_ok(int64_t chunk, int Ha, int Wa, int Ca, int Hb, int Wb, int Cb) {
    const int64_t span_a = (chunk + 64) * Ca * 4;
    const int64_t span_b = (chunk / ((int64_t)Ha * Wa) + 3) * Hb * Wb * Cb * 4;
    return span_a < 0x7fff0000LL && span_b < 0x7fff0000LL;
}

static int wgrad_target_blocks() {
    static int v = [] { const char* e = getenv("ADVMIX_WGRAD_BLOCKS"); int t = e ? atoi(e) : 1024; return t > 0 ? t : 1024; }();
    return v;
}

static bool wgrad_big() {
    static int v = [] { const char* e = getenv("ADVMIX_WGRAD_BIG"); return e ? atoi(e) : 1; }();
    return v != 0;
}

static int wgrad_impl(const float* a, const float* b, float* dw, int N, int Ha, int Wa, int Ca, int Hb, int Wb, int Cb,
                      int R, int S, int stride, int pad, float* part, int64_t part_floats, void* stream) {
    if (!a || !b || !dw || N <= 0 || Ca <= 0 || Cb <= 0 || stride < 1) return ADVMIX_EINVAL;
    if (Ha != (Hb + 2 * pad - R) / stride + 1 || Wa != (Wb + 2 * pad - S) / stride + 1) return ADVMIX_EINVAL;
    {
        int ns = 0;
        int rc = advmix_wgrad_lds_dispatch(a, b, dw, N, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad, part, part_floats, &ns,
                                           (hipStream_t)stream);
        if (rc >= 0) {
            if (rc == ADVMIX_OK && ns > 0) {
                const int64_t total = (int64_t)Ca * R * S * Cb;
                launch_reduce_slices(part, ns, total, dw, (hipStream_t)stream);
                ADVMIX_CHECK_LAUNCH();
            }
            return rc;
        }
    }
    if (!part) {
        int rc = advmix_wgrad_direct_dispatch(a, b, dw, N, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    WgP p{a, b, dw, N, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad, 0, part};
    p.xcd = wgrad_xcd_order();
    const int64_t P = (int64_t)N * Ha * Wa;
    const int Ntot = R * S * Cb;
    const bool vec = (Ca % 4 == 0) && (Cb % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    int nslices = 1;
#define LAUNCHW(WM_, WN_, V_) LAUNCHW2(WM_, WN_, V_, 1, 1)
#define LAUNCHW2(WM_, WN_, V_, TM_, TN_)                                                   \
    do {                                                                                  \
        int tiles = cdiv(Ca, 32 * WM_ * TM_) * cdiv(Ntot, 32 * WN_ * TN_);                \
        int64_t ns = wgrad_target_blocks() / tiles;                                       \
        if (ns < 1) ns = 1;                                                               \
        int64_t maxs = (P + advmix_wgrad_min_pix() - 1) / advmix_wgrad_min_pix();         \
        if (ns > maxs) ns = maxs;                                                         \
        int64_t chunk = ((P + ns - 1) / ns + 31) / 32 * 32;                               \
        p.chunk = (int)chunk;                                                             \
        dim3 g(cdiv(Ca, 32 * WM_ * TM_), cdiv(Ntot, 32 * WN_ * TN_), cdiv(P, chunk));     \
        nslices = (int)g.z;                                                               \
        if (part) {                                                                       \
            if ((int64_t)nslices * Ca * Ntot > part_floats) return ADVMIX_EINVAL;         \
        }                                                                                 \
        const bool v_ = V_ && wgrad_spans_ok(chunk, Ha, Wa, Ca, Hb, Wb, Cb);              \
        if (v_) hipLaunchKernelGGL((conv_wgrad<WM_, WN_, V_, TM_, TN_>), g, dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((conv_wgrad<WM_, WN_, false, TM_, TN_>), g, dim3(256), 0, st, p); \
        if (advmix_opts().trace_shapes) {                                                 \
            char nm[64];                                                                  \
            snprintf(nm, sizeof nm, "conv_wgrad<%d, %d, %s, %d, %d>", WM_, WN_, v_ ? "true" : "false", TM_, TN_); \
            advmix_trace_launch(nm, g, "wgrad", N, Hb, Wb, Cb, Ha, Wa, Ca, R, S, stride,  \
                                2.0 * N * (double)Ha * Wa * Ca * Cb * R * S);             \
        }                                                                                 \
    } while (0)
    if (Ca <= 32) {
        if (vec) LAUNCHW(1, 4, true); else LAUNCHW(1, 4, false);
    } else if (Ntot <= 32) {
        if (vec) LAUNCHW(4, 1, true); else LAUNCHW(4, 1, false);
    } else if (vec && !part && wgrad_big() && Ca >= 128 && Ntot >= 512 && (double)P * Ca * Ntot >= 1.0e10) {
        LAUNCHW2(2, 2, true, 2, 2);                        // 128 x 128 per workgroup (the U-Net's 4x4 convs)
    } else {
        if (vec) LAUNCHW(2, 2, true); else LAUNCHW(2, 2, false);
    }
#undef LAUNCHW
#undef LAUNCHW2
    if (part) {
        const int64_t total = (int64_t)Ca * Ntot;
        launch_reduce_slices(part, nslices, total, dw, st);
    }
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_conv_wgrad(const float* a, const float* b, float* dw,
                                 int N, int Ha, int Wa, int Ca, int Hb, int Wb, int Cb,
                                 int R, int S, int stride, int pad, void* stream) {
    return wgrad_impl(a, b, dw, N, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad, nullptr, 0, stream);
}

// 2-8 weight gradients of one geometry as ONE launch (see WgGroup).  a / b / dw: host arrays of n device pointers.  Served:
// Ca a multiple of 64 with Cb a multiple of 4 (the 3x3 C -> C convs of the pose nets' branches, C >= 64; the 1x1 and 3x3
// convs of the bottlenecks), and 3x3 32 -> 32 (wgrad_lds.hip: every workgroup walks several
// slabs of its problem before it merges); ADVMIX_EINVAL (nothing launched) otherwise - the caller launches the problems one by one.
// Pixel slices per problem of a grouped launch.  A launch runs in rounds of 256 CUs x (workgroups that fit a CU's LDS): what
// matters is how full its LAST round is, and how many partial tiles are merged with atomics (microbenchmark, eight 3x3 C -> C
// problems at B = 32, us per problem: 64 -> 64 with 480 / 760 / 1000 / 1520 workgroups 21.7 / 19.8 / 24.0 / 20.8 at 768 slots per
// round; 128 -> 128 with 504 / 720 / 1008 / 1512: 18.8 / 23.5 / 20.2 / 22.5 at 512; 256 -> 256 with 288 / 576 / 864 / 1440: 30.2 /
// 24.0 / 21.9 / 20.6 - profiles/r04_microbench_wgrad_group.log).  The slice count with the fullest rounds, fewer rounds first.
// ADVMIX_WGRAD_GROUP_BLOCKS=<n> forces a workgroup target instead (sweeps).
static int wgrad_group_slices(int tiles_all, int wg_per_cu, int64_t maxs) {
    static const int force = [] { const char* e = getenv("ADVMIX_WGRAD_GROUP_BLOCKS"); return e ? atoi(e) : 0; }();
    if (force > 0) {
        int64_t ns = force / tiles_all;
        return (int)(ns < 1 ? 1 : (ns > maxs ? maxs : ns));
    }
    const int slots = 256 * wg_per_cu;
    int best = 1;
    double best_score = -1.0;
    for (int ns = 1; ns <= 64 && ns <= maxs; ++ns) {
        const int64_t w = (int64_t)tiles_all * ns;
        const int64_t rounds = (w + slots - 1) / slots;
        if (rounds > 3 && best_score >= 0) break;
        const double score = (double)w / (double)(rounds * slots) - 0.02 * (double)(rounds - 1);
        if (score > best_score + 1e-9) { best_score = score; best = ns; }
    }
    return best;
}

extern "C" int advmix_conv_wgrad_group(int n, const float* const* a, const float* const* b, float* const* dw,
                                       int N, int Ha, int Wa, int Ca, int Hb, int Wb, int Cb,
                                       int R, int S, int stride, int pad, void* stream) {
    if (n < 2 || n > WG_MAXG || !a || !b || !dw || N <= 0 || Ca <= 0 || Cb <= 0 || stride < 1) return ADVMIX_EINVAL;
    if (Ha != (Hb + 2 * pad - R) / stride + 1 || Wa != (Wb + 2 * pad - S) / stride + 1) return ADVMIX_EINVAL;
    const int Ntot = R * S * Cb;
    if (advmix_opts().deterministic) return ADVMIX_EINVAL;
    for (int i = 0; i < n; ++i)
        if (!a[i] || !b[i] || !dw[i]) return ADVMIX_EINVAL;
    {                                                      // 3x3 32 -> 32: the LDS-patch kernel, several slabs per workgroup
        int rc = advmix_wgrad_lds_group_dispatch(n, a, b, dw, N, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    if (Ca % 64 != 0 || Cb % 4 != 0) return ADVMIX_EINVAL;      // (ragged column tiles are bounds-checked: 3x3 64 -> 64 has 576)
    if (cdiv(Ntot, 128) * 128 > Ntot + Ntot / 6) return ADVMIX_EINVAL;   // a 128-column tile mostly empty (1x1 64 -> 256: 64 columns)
    WgGroup g;
    g.n = n;
    for (int i = 0; i < WG_MAXG; ++i) {
        g.a[i] = i < n ? a[i] : nullptr;
        g.b[i] = i < n ? b[i] : nullptr;
        g.dw[i] = i < n ? dw[i] : nullptr;
        if (i < n && (!a[i] || !b[i] || !dw[i])) return ADVMIX_EINVAL;
    }
    WgP p{nullptr, nullptr, nullptr, N, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad, 0, nullptr};
    p.xcd = wgrad_xcd_order();
    const int64_t P = (int64_t)N * Ha * Wa;
    hipStream_t st = (hipStream_t)stream;
    const bool big = Ca % 128 == 0;                        // 128 x 128 per workgroup (2 x 2 tiles per wave), else 64 x 128
    const int tiles = big ? cdiv(Ca, 128) * cdiv(Ntot, 128) : cdiv(Ca, 64) * cdiv(Ntot, 128);
    const int64_t maxs = (P + 63) / 64;
    const int64_t ns = wgrad_group_slices(tiles * n, big ? 2 : 3, maxs);    // (LDS: 64 KB / 48 KB per workgroup)
    const int64_t chunk = ((P + ns - 1) / ns + 31) / 32 * 32;
    p.chunk = (int)chunk;
    g.slices = (int)cdiv(P, chunk);
    p.excl = g.slices == 1;
    for (int i = 1; i < n && p.excl; ++i)                  // (the same buffer twice - shared weights - keeps the atomics)
        for (int j = 0; j < i; ++j)
            if (dw[i] == dw[j]) { p.excl = 0; break; }
    if (!wgrad_spans_ok(chunk, Ha, Wa, Ca, Hb, Wb, Cb)) return ADVMIX_EINVAL;
    if ((int64_t)g.slices * n > 65535) return ADVMIX_EINVAL;
    const dim3 grid(big ? cdiv(Ca, 128) : cdiv(Ca, 64), cdiv(Ntot, 128), g.slices * n);
    if (big) hipLaunchKernelGGL((conv_wgrad_group<2, 2, 2, 2>), grid, dim3(256), 0, st, p, g);
    else hipLaunchKernelGGL((conv_wgrad_group<1, 4, 2, 1>), grid, dim3(256), 0, st, p, g);
    if (advmix_opts().trace_shapes) {
        char nm[64], kd[24];
        snprintf(nm, sizeof nm, "conv_wgrad_group<%s>", big ? "2, 2, 2, 2" : "1, 4, 2, 1");
        snprintf(kd, sizeof kd, "wgrad x%d", n);
        advmix_trace_launch(nm, grid, kd, N, Hb, Wb, Cb, Ha, Wa, Ca, R, S, stride,
                            2.0 * n * N * (double)Ha * Wa * Ca * Cb * R * S);
    }
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// n (1 ... 16) weight gradients of ANY geometries as one launch (see WgMulti).  geoms: n x 11 ints, advmix_conv_wgrad's
// (N, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad) per problem.  Every channel count must be a multiple of 4 (the vector form
// of wgrad_body).  All problems share one pixel-slice length: the shortest (a multiple of 32, >= ADVMIX_WGRAD_MINPIX) that keeps
// the launch within ADVMIX_WGM_WGS (768 = three per CU) workgroups, so every workgroup carries about the same work.
// 0 = launched, 1 = not served (nothing launched: call advmix_conv_wgrad per problem), ADVMIX_EINVAL for bad arguments and
// in deterministic mode.
extern "C" int advmix_conv_wgrad_multi(int n, const float* const* a, const float* const* b, float* const* dw,
                                       const int* geoms, void* stream) {
    if (n < 1 || n > WG_MULTI || !a || !b || !dw || !geoms) return ADVMIX_EINVAL;
    if (advmix_opts().deterministic) return ADVMIX_EINVAL;
    static const int target = [] { const char* e = getenv("ADVMIX_WGM_WGS"); int t = e ? atoi(e) : 768; return t > 0 ? t : 768; }();
    WgMulti g;
    g.n = n;
    int64_t P[WG_MULTI];
    int tiles[WG_MULTI];
    int64_t pmax = 0;
    for (int i = 0; i < n; ++i) {
        const int* q = geoms + 11 * i;
        const int N = q[0], Ha = q[1], Wa = q[2], Ca = q[3], Hb = q[4], Wb = q[5], Cb = q[6], R = q[7], S = q[8], stride = q[9], pad = q[10];
        if (!a[i] || !b[i] || !dw[i] || N <= 0 || Ca <= 0 || Cb <= 0 || stride < 1 || R < 1 || S < 1) return ADVMIX_EINVAL;
        if (Ha != (Hb + 2 * pad - R) / stride + 1 || Wa != (Wb + 2 * pad - S) / stride + 1) return ADVMIX_EINVAL;
        if (Ca % 4 != 0 || Cb % 4 != 0) return 1;
        g.p[i] = WgP{a[i], b[i], dw[i], N, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad, 0, nullptr};
        g.p[i].excl = 0;
        g.p[i].xcd = 0;
        g.cfg[i] = Ca <= 32 ? 1 : 0;
        g.gx[i] = g.cfg[i] ? cdiv(Ca, 32) : cdiv(Ca, 64);
        g.gy[i] = g.cfg[i] ? cdiv(R * S * Cb, 128) : cdiv(R * S * Cb, 64);
        tiles[i] = g.gx[i] * g.gy[i];
        P[i] = (int64_t)N * Ha * Wa;
        if (P[i] > pmax) pmax = P[i];
    }
    int64_t chunk = (advmix_wgrad_min_pix() + 31) / 32 * 32;
    for (;; chunk += 32) {
        int64_t total = 0;
        for (int i = 0; i < n; ++i) total += (int64_t)tiles[i] * cdiv(P[i], chunk);
        if (total <= target || chunk >= pmax) break;
    }
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        const int* q = geoms + 11 * i;
        if (!wgrad_spans_ok(chunk, q[1], q[2], q[3], q[4], q[5], q[6])) return 1;
        g.p[i].chunk = (int)chunk;
        g.start[i] = (int)total;
        total += (int64_t)tiles[i] * cdiv(P[i], chunk);
        if (total > 0x3fffffff) return 1;
    }
    for (int i = n; i <= WG_MULTI; ++i) g.start[i] = (int)total;
    for (int i = n; i < WG_MULTI; ++i) { g.gx[i] = g.gy[i] = 1; g.cfg[i] = 0; g.p[i] = g.p[0]; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv_wgrad_multi, dim3((unsigned)total), dim3(256), 0, st, g);
    if (advmix_opts().trace_shapes) {
        double fl = 0;
        for (int i = 0; i < n; ++i) {
            const int* q = geoms + 11 * i;
            fl += 2.0 * q[0] * (double)q[1] * q[2] * q[3] * q[6] * q[7] * q[8];
        }
        char kd[24];
        snprintf(kd, sizeof kd, "wgrad multi x%d", n);
        const int* q = geoms;
        advmix_trace_launch("conv_wgrad_multi", dim3((unsigned)total), kd, q[0], q[4], q[5], q[6], q[1], q[2], q[3], q[7], q[8], q[9], fl);
    }
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// Deterministic weight gradient: every pixel slice STORES its partial tile into ``ws`` (ws_bytes; the call needs
// 4 * slices * Ca * R * S * Cb bytes, at most advmix_wgrad_det_ws_bytes) and a second launch adds the slices to dw in
// slice order - no fp32 atomics, bit-reproducible run to run.  ADVMIX_EINVAL when ws is too small.
extern "C" int advmix_conv_wgrad_det(const float* a, const float* b, float* dw,
                                     int N, int Ha, int Wa, int Ca, int Hb, int Wb, int Cb,
                                     int R, int S, int stride, int pad, void* ws, int64_t ws_bytes, void* stream) {
    if (!ws) return ADVMIX_EINVAL;
    return wgrad_impl(a, b, dw, N, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad, (float*)ws, ws_bytes / 4, stream);
}

extern "C" int64_t advmix_wgrad_det_ws_bytes(int Ca, int Cb, int R, int S) {
    // slices * tiles <= max(wgrad_target_blocks, tiles) and a tile holds at most 64 x 128 outputs of which Ca x Ntot are
    // real: slices * Ca * Ntot <= target_blocks * 64 * 128 floats, or one slice of the whole gradient
    const int64_t whole = (int64_t)Ca * R * S * Cb;
    const int64_t capped = (int64_t)wgrad_target_blocks() * 64 * 128;
    return 4 * (whole > capped ? whole : capped);
}

extern "C" int advmix_transpose_w(const float* in, float* out, int A, int T, int B, void* stream) {
    if (!in || !out || A <= 0 || T <= 0 || B <= 0) return ADVMIX_EINVAL;
    int64_t total = (int64_t)A * T * B;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(transpose_w_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, A, T, B);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

static int bias_grad_impl(const float* dy, float* db, int64_t rows, int C, float* part, int64_t part_floats, void* stream) {
    if (!dy || !db || rows <= 0 || C <= 0) return ADVMIX_EINVAL;
    static const int v4_blocks = [] { const char* e = getenv("ADVMIX_BIAS_BLOCKS"); int v = e ? atoi(e) : 128; return v > 0 ? v : 128; }();   // (128 / 256 / 512 / 1024 blocks: 17.9 / 20.0 / 25.2 / 33.1 us at 101 MB)
    const int target = C % 4 == 0 ? v4_blocks : 1024;      // (deterministic mode: the partial buffer holds <= 1024 blocks)
    int64_t rpb = (rows + target - 1) / target;
    if (rpb < 32) rpb = 32;
    int blocks = (int)((rows + rpb - 1) / rpb);
    if (part && (int64_t)blocks * C > part_floats) return ADVMIX_EINVAL;
    if (C % 4 == 0)
        hipLaunchKernelGGL(bias_grad_kernel_v4, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, db, rows, C, rpb, part);
    else
        hipLaunchKernelGGL(bias_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, db, rows, C, rpb, part);
    if (part)
        launch_reduce_slices(part, blocks, (int64_t)C, db, (hipStream_t)stream);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_bias_grad(const float* dy, float* db, int64_t rows, int C, void* stream) {
    return bias_grad_impl(dy, db, rows, C, nullptr, 0, stream);
}

// deterministic variant: block partials in ws (<= 1024 * C floats), summed in block order
extern "C" int advmix_bias_grad_det(const float* dy, float* db, int64_t rows, int C, void* ws, int64_t ws_bytes,
                                    void* stream) {
    if (!ws) return ADVMIX_EINVAL;
    return bias_grad_impl(dy, db, rows, C, (float*)ws, ws_bytes / 4, stream);
}

extern "C" int advmix_conv_tr_w(const float* x, const float* w, const float* bias, float* y,
                                int N, int Hs, int Ws, int Ck, int Hb, int Wb, int Cn,
                                int R, int S, int stride, int pad, void* stream) {
    if (!x || !w || !y || N <= 0 || Ck <= 0 || Cn <= 0 || stride < 1 || stride > 8) return ADVMIX_EINVAL;
    if (Hs != (Hb + 2 * pad - R) / stride + 1 || Ws != (Wb + 2 * pad - S) / stride + 1) return ADVMIX_EINVAL;
    if (!use_direct()) return ADVMIX_EINVAL;
    int64_t Mmax = (int64_t)N * cdiv(Hb, stride) * cdiv(Wb, stride);
    int rc = advmix_conv_direct_dispatch(1, x, w, bias, y, N, Hs, Ws, Ck, Hb, Wb, Cn, R, S, stride, pad, Mmax,
                                         (hipStream_t)stream, 1);
    return rc < 0 ? ADVMIX_EINVAL : rc;
}

// Conv2d forward with a fused epilogue (conv_direct only): eval-mode BatchNorm + residual + activation,
// and/or per-slab column statistics of the raw output for a following train-mode BatchNorm.
extern "C" int advmix_conv_fwd_ex(const float* x, const float* w, const float* bias, float* y,
                                  int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                                  int R, int S, int stride, int pad,
                                  const float* bn_gamma, const float* bn_beta, const float* bn_rm, const float* bn_rv,
                                  float bn_eps, const float* residual, int act, double* stats, int* stats_nbg,
                                  void* stream) {
    if (!x || !w || !y || N <= 0 || Ci <= 0 || Co <= 0 || stride < 1) return ADVMIX_EINVAL;
    if (Ho != (Hi + 2 * pad - R) / stride + 1 || Wo != (Wi + 2 * pad - S) / stride + 1) return ADVMIX_EINVAL;
    if ((bn_gamma != nullptr) != (bn_beta && bn_rm && bn_rv)) return ADVMIX_EINVAL;
    if (stats && !stats_nbg) return ADVMIX_EINVAL;
    if (!use_direct()) return ADVMIX_EINVAL;
    ConvEpi e{bn_gamma, bn_beta, bn_rm, bn_rv, residual, bn_eps, act, stats, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    int rc = advmix_conv_direct_dispatch(0, x, w, bias, y, N, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride, pad,
                                         (int64_t)N * Ho * Wo, (hipStream_t)stream, 0, &e, stats_nbg);
    return rc < 0 ? ADVMIX_EINVAL : rc;
}

// y = conv_transpose(x, w) + addend (same layout as y, may be NULL): the input gradient of a conv whose input has
// another consumer (residual, fuse layers) takes that consumer's gradient in the epilogue.  conv_direct only.
extern "C" int advmix_conv_tr_w_add(const float* x, const float* w, const float* addend, float* y,
                                    int N, int Hs, int Ws, int Ck, int Hb, int Wb, int Cn,
                                    int R, int S, int stride, int pad, void* stream) {
    if (!x || !w || !y || N <= 0 || Ck <= 0 || Cn <= 0 || stride < 1 || stride > 8) return ADVMIX_EINVAL;
    if (Hs != (Hb + 2 * pad - R) / stride + 1 || Ws != (Wb + 2 * pad - S) / stride + 1) return ADVMIX_EINVAL;
    if (!use_direct()) return ADVMIX_EINVAL;
    ConvEpi epi{nullptr, nullptr, nullptr, nullptr, addend, 0.f, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    int64_t Mmax = (int64_t)N * cdiv(Hb, stride) * cdiv(Wb, stride);
    int rc = advmix_conv_direct_dispatch(1, x, w, nullptr, y, N, Hs, Ws, Ck, Hb, Wb, Cn, R, S, stride, pad, Mmax,
                                         (hipStream_t)stream, 1, addend ? &epi : nullptr);
    return rc < 0 ? ADVMIX_EINVAL : rc;
}

// Input gradient of a conv whose INPUT is y = act(BN(c) + residual) of a train-mode BatchNorm:
//   g = (conv_transpose(x, w) + addend) * act'(y)        written to ``g_out`` (same layout as y)
//   stats[0][ch][slot] += sum g,  stats[1][ch][slot] += sum g * (c - mean) * invstd      (fp64 atomics, ns slots)
// i.e. everything BatchNorm backward needs from a pass over (dy, y, c) is produced here; advmix_norm_bwd_apply_slots
// finishes it.  The sign of y (act != ADVMIX_ACT_NONE) comes from ``act_mask`` - the bit-per-element mask
// advmix_norm_apply_slots wrote beside y - or, when act_mask is NULL, is recomputed from c with ``bn_gamma`` / ``bn_beta``
// (valid for y = act(BN(c)) WITHOUT a residual as advmix_norm_apply_slots computed it: the same fused multiply-add).
// ``*stats_ns``: in = slots per channel to use (0 = the library default), out = the number used.  Returns 1
// (ADVMIX_EINVAL) when the shape is not served (grid K split, tensors >= 2 GiB, Ck % 16 != 0, a mask with Cn % 16 != 0):
// nothing was launched and the caller runs the separate kernels.
extern "C" int advmix_conv_tr_w_bnb(const float* x, const float* w, const float* addend, float* g_out,
                                    int N, int Hs, int Ws, int Ck, int Hb, int Wb, int Cn,
                                    int R, int S, int stride, int pad,
                                    const unsigned char* act_mask, const float* bn_c, const float* bn_mean,
                                    const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                    int act, double* stats, int* stats_ns, void* stream) {
    if (!x || !w || !g_out || !bn_c || !bn_mean || !bn_invstd || !stats || !stats_ns) return ADVMIX_EINVAL;
    if (N <= 0 || Ck <= 0 || Cn <= 0 || stride < 1 || stride > 8) return ADVMIX_EINVAL;
    if (act != ADVMIX_ACT_NONE && !act_mask && !(bn_gamma && bn_beta)) return ADVMIX_EINVAL;
    if (Hs != (Hb + 2 * pad - R) / stride + 1 || Ws != (Wb + 2 * pad - S) / stride + 1) return ADVMIX_EINVAL;
    if (!use_direct()) return ADVMIX_EINVAL;
    ConvEpi epi{nullptr, nullptr, nullptr, nullptr, addend, 0.f, 0, stats, act_mask, bn_c, bn_mean, bn_invstd, bn_gamma, bn_beta, act};
    int64_t Mmax = (int64_t)N * cdiv(Hb, stride) * cdiv(Wb, stride);
    int rc = advmix_conv_direct_dispatch(1, x, w, nullptr, g_out, N, Hs, Ws, Ck, Hb, Wb, Cn, R, S, stride, pad, Mmax,
                                         (hipStream_t)stream, 1, &epi, stats_ns);
    return rc < 0 ? ADVMIX_EINVAL : rc;
}
