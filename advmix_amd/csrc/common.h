// Shared helpers for the gfx950 kernels (device code is CDNA4-only: wave = 64 lanes).
#pragma once
#include <stdlib.h>
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

#define ADVMIX_STAT_SLOTS_MAX 64
// optional fused conv epilogue
struct ConvEpi {
    const float *gamma, *beta, *rm, *rv;   // forward gather: eval-mode BatchNorm (all four or none)
    const float* res;                      // forward gather: residual added after BN; transposed gather: addend
    float eps;
    int act;
    double* stats;                         // out: [2][Co][ns] fp64 slots.  forward gather: column (sum, sumsq) of the
                                           // RAW conv output; transposed gather: (sum g, sum g * xhat), see bnb_*
    // transposed gather + stats: the output is dL/dy of y = act(BN(c) + residual); the epilogue writes
    // g = that * act'(y) and the two BatchNorm-backward channel sums.  Sign of y: from bnb_mask (a bit per element, see
    // advmix_norm_apply_slots) or - no residual - recomputed from c with bnb_gamma / bnb_beta
    const unsigned char* bnb_mask;
    const float *bnb_c, *bnb_mean, *bnb_invstd, *bnb_gamma, *bnb_beta;
    int bnb_act;
};
int advmix_wgrad_lds_build_flags(void);   // wgrad_lds.hip: its bit of advmix_build_flags() (0 unless a measurement variant)
// conv_direct.hip: second-generation conv; returns -1 when the shape is not eligible, -2 when only the
// fused epilogue is unavailable
int advmix_conv_direct_dispatch(int mode, const float* x, const float* w, const float* bias, float* y,
                                int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S,
                                int stride, int pad, int64_t Mmax, hipStream_t st, int bt = 0,
                                const ConvEpi* epi = nullptr, int* stats_nbg = nullptr);

// conv_direct.hip: nb independent GEMMs c[b] = a[b] . w[b]^T in one launch (conv_wino4.hip); -1 = not served
int advmix_conv_direct_gemm_batched(const float* a, const float* w, float* c, int nb, int rows, int K, int Nc, hipStream_t st);

// conv_direct.hip: 2-4 problems of one kind in one launch; -1 = cannot be served as one launch (nothing launched)
struct ConvProb {
    const float *x, *w, *bias;
    float* y;
    int N, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride, pad;      // x / y sides as in advmix_conv_direct_dispatch
    int64_t Mmax;
    const ConvEpi* epi;                                    // or null
    int stats_nbg;                                         // in: slots to use (0 = default), out: slots used
};
int advmix_conv_direct_group(int mode, int bt, int n, ConvProb* probs, hipStream_t st);

// runtime-tunable dispatch options (advmix_set_option / ADVMIX_* environment at first use)
struct AdvmixOpts {
    int direct;            // 1: conv_direct allowed, 0: first-generation conv_igemm only
    int wgrad_direct;      // 1: register-fragment wgrad kernel allowed
    int ksplit_wg;         // 1: layers with too few tiles split K inside the workgroup (fused epilogue kept), 0: across the grid
    int trace_shapes;      // 1: log every MFMA launch's shape (advmix_trace_launch)
    int deterministic;     // 1: conv_direct never splits K across the grid (fp32 atomics); see ops.py set_deterministic
    int wgrad_lds;         // 1: LDS-patch weight gradient for 3x3 s1 32->32 when the batch fills the chip, 2: whenever eligible
    int stat_slots;        // fp64 slots per channel the statistics epilogues fold their workgroup sums onto (power of 2 <= 64; 0 = by grid size)
};
AdvmixOpts& advmix_opts();

// Workgroup cap of the grid-stride streaming kernels (ADVMIX_STREAM_WGS) and of the BatchNorm slot kernels
// (ADVMIX_SLOT_WGS).  Round 3: narrow HBM-bound launches leave CUs to the MFMA kernels of the other lanes - see DESIGN.md.
inline int advmix_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : dflt;
    return v > 0 ? v : dflt;
}

// Fewest pixels a weight-gradient workgroup reduces over (the pixel axis is split over workgroups to fill the chip; every slice
// merges its tile into dW with atomics that serialise per cache line).  ADVMIX_WGRAD_MINPIX.
inline int advmix_wgrad_min_pix() {
    static const int v = [] { int x = advmix_env_int("ADVMIX_WGRAD_MINPIX", 64); return x < 32 ? 32 : x; }();
    return v;
}
inline int advmix_stream_cap() {
    static const int v = advmix_env_int("ADVMIX_STREAM_WGS", 2048);
    return v;
}

// Measurement aid (off unless advmix_set_option("trace_shapes", 1) with ADVMIX_TRACE_SHAPES=<file> in the environment):
// one CSV line per MFMA launch - kernel template, grid, problem shape, algorithmic FLOPs - so that a rocprofv3 kernel
// trace, which groups by template only, can be split per SHAPE (tools/kernel_shapes.py joins on kernel + grid).
void advmix_trace_launch(const char* kernel, dim3 grid, const char* kind, int N, int Hi, int Wi, int Ci, int Ho, int Wo,
                         int Co, int R, int S, int stride, double flops);

// wgrad_direct.hip: both operands loaded in fragment layout; -1 = not eligible
int advmix_wgrad_direct_dispatch(const float* a, const float* b, float* dw, int N, int Ha, int Wa, int Ca, int Hb,
                                 int Wb, int Cb, int R, int S, int stride, int pad, hipStream_t st);

// wgrad_lds.hip: 3x3 s1 32->32 with both operands staged once in LDS; -1 = not eligible.  With ``part`` the workgroups
// store their partial tiles ([*nslices][32*9*32], slab order, summed by the caller) instead of adding them to dw atomically.
int advmix_wgrad_lds_group_dispatch(int n, const float* const* a, const float* const* b, float* const* dw, int N, int Ha,
                                    int Wa, int Ca, int Hb, int Wb, int Cb, int R, int S, int stride, int pad, hipStream_t st);
int advmix_wgrad_lds_dispatch(const float* a, const float* b, float* dw, int N, int Ha, int Wa, int Ca, int Hb, int Wb,
                              int Cb, int R, int S, int stride, int pad, float* part, int64_t part_floats, int* nslices,
                              hipStream_t st);
