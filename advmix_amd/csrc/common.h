// Shared helpers for the gfx950 kernels (device code is CDNA4-only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/advmix_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ADVMIX_CHECK_LAUNCH() \
    do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return ADVMIX_ELAUNCH; } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Activations as selects on a slope that is uniform over the launch (ReLU 0, LeakyReLU 0.2, none 1): a chain of
// run-time ``if (act == ...)`` per element compiled to scalar branches with an s_waitcnt vmcnt(0) at every
// re-convergence in the memory-bound kernels.
__device__ __forceinline__ float act_neg_slope(int act) {
    return act == ADVMIX_ACT_RELU ? 0.f : (act == ADVMIX_ACT_LEAKY02 ? 0.2f : 1.f);
}
__device__ __forceinline__ float act_fwd(float v, int act) {
    const float s = act_neg_slope(act);
    return v > 0.f ? v : (s == 0.f ? 0.f : v * s);
}
// derivative expressed through the activation OUTPUT y (both activations are sign-preserving)
__device__ __forceinline__ float act_grad(float y, int act) {
    return y > 0.f ? 1.f : act_neg_slope(act);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

#define ADVMIX_STAT_SLOTS 64
// optional fused conv epilogue (forward gather only)
struct ConvEpi {
    const float *gamma, *beta, *rm, *rv;   // eval-mode BatchNorm (all four or none)
    const float* res;                      // residual added after BN
    float eps;
    int act;
    double* stats;                         // out: [2][Co][nbg] column (sum, sumsq) of the RAW conv output
};
// conv_direct.hip: second-generation conv; returns -1 when the shape is not eligible, -2 when only the
// fused epilogue is unavailable
int advmix_conv_direct_dispatch(int mode, const float* x, const float* w, const float* bias, float* y,
                                int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S,
                                int stride, int pad, int64_t Mmax, hipStream_t st, int bt = 0,
                                const ConvEpi* epi = nullptr, int* stats_nbg = nullptr);

// conv3x3_lds.hip: LDS-resident-patch 3x3/s1/p1 conv (flip = 1: its input gradient); -1 = not eligible
int advmix_conv3x3_lds_dispatch(int flip, const float* x, const float* w, const float* bias, float* y, int N, int H,
                                int W, int Ci, int Co, hipStream_t st);

// runtime-tunable dispatch options (advmix_set_option / ADVMIX_* environment at first use)
struct AdvmixOpts {
    int direct;            // 1: conv_direct / conv3x3_lds allowed, 0: first-generation conv_igemm only
    int conv3;             // 1: LDS-patch 3x3 kernel allowed
    int conv3_min_items;   // minimum (tile x chunk) items before the persistent 3x3 kernel is used
    int conv3_grid;        // persistent workgroups (256 = one per CU)
    int wgrad_direct;      // 1: register-fragment wgrad kernel allowed
    int mfma16;            // 1: conv_direct uses the 16x16x4 MFMA shape (16 pixel rows x 64 B per fragment load)
    int ksplit_wg;         // 1: layers with too few tiles split K inside the workgroup (fused epilogue kept), 0: across the grid
};
AdvmixOpts& advmix_opts();

// wgrad_direct.hip: both operands loaded in fragment layout; -1 = not eligible
int advmix_wgrad_direct_dispatch(const float* a, const float* b, float* dw, int N, int Ha, int Wa, int Ca, int Hb,
                                 int Wb, int Cb, int R, int S, int stride, int pad, hipStream_t st);
