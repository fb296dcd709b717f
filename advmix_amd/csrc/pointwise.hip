// HBM-bound pointwise / data-movement kernels on NHWC fp32: activations, channel-slice copies
// (cat), HRNet's nearest-upsample fuse sum, 3x3/s2 max-pool, axpy/fill.
// Reference sites: pose_hrnet.py:35,54-55,206,254-265; pose_resnet.py:115;
// Unet_generator.py:42,44,83.
#include "common.h"

namespace {

static int stream_blocks(int64_t total) {
    int64_t b = (total + 255) / 256;
    const int cap = advmix_stream_cap();
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

template <bool VEC>
__global__ __launch_bounds__(256) void act_copy_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                       int ldy, int64_t rows, int C, int act) {
    constexpr int V = VEC ? 4 : 1;
    const int CV = C / V;
    const int64_t total = rows * CV;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i / CV;
        int c = (int)(i - r * CV) * V;
        if (VEC) {
            f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], act);
            *reinterpret_cast<f32x4*>(y + r * ldy + c) = v;
        } else {
            y[r * ldy + c] = act_fwd(x[r * ldx + c], act);
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, int lddy,
                                                      const float* __restrict__ y, int ldy, float* __restrict__ dx,
                                                      int lddx, int64_t rows, int C, int act) {
    constexpr int V = VEC ? 4 : 1;
    const int CV = C / V;
    const int64_t total = rows * CV;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i / CV;
        int c = (int)(i - r * CV) * V;
#pragma unroll
        for (int e = 0; e < V; ++e)
            dx[r * lddx + c + e] = dy[r * lddy + c + e] * act_grad(y[r * ldy + c + e], act);
    }
}

struct d4s { double v[4]; };

struct FuseArgs {
    const float* in[4];
    float* din[4];
    int shift[4];
    int n;
};

// y = act(sum_j in_j[n, h>>s_j, w>>s_j, c])
__global__ __launch_bounds__(256) void fuse_sum_kernel(FuseArgs a, float* __restrict__ y, int N, int H, int W,
                                                       int CV, int act) {
    const int64_t total = (int64_t)N * H * W * CV;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cv = (int)(i % CV);
        int64_t pix = i / CV;
        int w = (int)(pix % W);
        int64_t q = pix / W;
        int h = (int)(q % H);
        int n = (int)(q / H);
        f32x4 s = {0, 0, 0, 0};
        for (int j = 0; j < a.n; ++j) {
            int sh = a.shift[j];
            int64_t off = (((int64_t)n * (H >> sh) + (h >> sh)) * (W >> sh) + (w >> sh)) * CV + cv;
            s += reinterpret_cast<const f32x4*>(a.in[j])[off];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = act_fwd(s[e], act);
        reinterpret_cast<f32x4*>(y)[i] = s;
    }
}

__global__ __launch_bounds__(256) void mask_grad_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        float* __restrict__ g, int64_t n4, int act) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 d = reinterpret_cast<const f32x4*>(dy)[i];
        f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] *= act_grad(yv[e], act);
        reinterpret_cast<f32x4*>(g)[i] = d;
    }
}

// din[n, hs, ws, c] = sum over the f x f block of g (f = 1 << shift); g is at full resolution
__global__ __launch_bounds__(256) void pool_sum_kernel(const float* __restrict__ g, float* __restrict__ din, int N,
                                                       int H, int W, int CV, int shift) {
    const int Hs = H >> shift, Ws = W >> shift, f = 1 << shift;
    const int64_t total = (int64_t)N * Hs * Ws * CV;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cv = (int)(i % CV);
        int64_t pix = i / CV;
        int ws = (int)(pix % Ws);
        int64_t q = pix / Ws;
        int hs = (int)(q % Hs);
        int n = (int)(q / Hs);
        f32x4 s = {0, 0, 0, 0};
        for (int dh = 0; dh < f; ++dh)
            for (int dw = 0; dw < f; ++dw) {
                int64_t off = (((int64_t)n * H + (hs * f + dh)) * W + (ws * f + dw)) * CV + cv;
                s += reinterpret_cast<const f32x4*>(g)[off];
            }
        reinterpret_cast<f32x4*>(din)[i] = s;
    }
}

// ---- fuse-sum backward in ONE launch, with the BatchNorm-backward sums of the sources -----------------------------------
// HRNet's fuse layers end in conv -> BN (no activation) -> sum with the other branches -> ReLU (pose_hrnet.py:196-265).
// The gradient of each BN output comes out of this kernel - g = dy * act'(y) for the sources at the sum's resolution, its
// 2^s x 2^s block sum for the up-sampled ones - so the kernel also leaves the two channel sums their BatchNorm backward
// needs (sum g_j, sum g_j * xhat_j) in the source's fp64 slots: norm_bwd_apply_slots then replaces the separate
// statistics pass + finalize + apply (the last 53 conv + BN pairs per HRNet-W32 backward pass that still took those).
// One launch instead of mask_grad + a pool_sum per source: segment 0 of the grid writes g, segment k the pooled gradient
// of one up-sampled source straight from (dy, y) - no dependency between segments.
struct FuseBwdArgs {
    int n;                       // segments: 0 = g at full resolution, k >= 1 = pooled source src[k]
    int start[6];                // first block of each segment (start[n] = grid size)
    int shift[5];                // per segment (segment 0: 0)
    float* out[5];               // segment 0: g (may be null if nobody needs it), k: the source's gradient
    // BatchNorm-backward targets.  Segment 0 serves up to three same-resolution sources (tc0[i]), segment k one.
    int nt0;
    const float* tc0[3]; const float* tmean0[3]; const float* tinvstd0[3]; double* tslots0[3];
    const float* tc[5]; const float* tmean[5]; const float* tinvstd[5]; double* tslots[5];
    int ns;                      // slots per channel and statistic in use
};

__global__ __launch_bounds__(256) void fuse_bwd_kernel(FuseBwdArgs a, const float* __restrict__ dy,
                                                       const float* __restrict__ y, int N, int H, int W, int CV, int act) {
    __shared__ d4s red[256];
    const int tid = threadIdx.x;
    int seg = 0;
#pragma unroll
    for (int k = 1; k < 5; ++k)
        if (k < a.n && (int)blockIdx.x >= a.start[k]) seg = k;
    const int blk = blockIdx.x - a.start[seg], nblk = a.start[seg + 1] - a.start[seg];
    const int C = CV * 4, cv = tid % CV;                  // 256 % CV == 0: a thread keeps its four channels
    const int64_t stride = (int64_t)nblk * 256;
    const float slope = act_neg_slope(act);
    double s1[4] = {0, 0, 0, 0}, s2[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    int nt;
    const float* tc[3]; f32x4 mu[3], is[3]; double* tsl[3];
    if (seg == 0) {
        nt = a.nt0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < nt) {
                tc[i] = a.tc0[i]; tsl[i] = a.tslots0[i];
                mu[i] = *reinterpret_cast<const f32x4*>(a.tmean0[i] + cv * 4);
                is[i] = *reinterpret_cast<const f32x4*>(a.tinvstd0[i] + cv * 4);
            }
        const int64_t total = (int64_t)N * H * W * CV;
        float* const g = a.out[0];
        for (int64_t i = (int64_t)blk * 256 + tid; i < total; i += stride) {
            f32x4 d = reinterpret_cast<const f32x4*>(dy)[i];
            const f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = yv[e] > 0.f ? d[e] : d[e] * slope;
            if (g) reinterpret_cast<f32x4*>(g)[i] = d;
            if (nt > 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) s1[e] += (double)d[e];
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    if (t < nt) {
                        const f32x4 cvv = reinterpret_cast<const f32x4*>(tc[t])[i];
#pragma unroll
                        for (int e = 0; e < 4; ++e) s2[t][e] += (double)d[e] * (double)((cvv[e] - mu[t][e]) * is[t][e]);
                    }
            }
        }
    } else {
        nt = a.tc[seg] ? 1 : 0;
        if (nt) {
            tc[0] = a.tc[seg]; tsl[0] = a.tslots[seg];
            mu[0] = *reinterpret_cast<const f32x4*>(a.tmean[seg] + cv * 4);
            is[0] = *reinterpret_cast<const f32x4*>(a.tinvstd[seg] + cv * 4);
        }
        const int sh = a.shift[seg], Hs = H >> sh, Ws = W >> sh, f = 1 << sh;
        const int64_t total = (int64_t)N * Hs * Ws * CV;
        float* const din = a.out[seg];
        for (int64_t i = (int64_t)blk * 256 + tid; i < total; i += stride) {
            const int64_t pix = i / CV;
            const int ws = (int)(pix % Ws);
            const int64_t q = pix / Ws;
            const int hs = (int)(q % Hs), n = (int)(q / Hs);
            f32x4 s = {0, 0, 0, 0};
            for (int dh = 0; dh < f; ++dh)
                for (int dw = 0; dw < f; ++dw) {
                    const int64_t off = (((int64_t)n * H + (hs * f + dh)) * W + (ws * f + dw)) * CV + cv;
                    f32x4 d = reinterpret_cast<const f32x4*>(dy)[off];
                    const f32x4 yv = reinterpret_cast<const f32x4*>(y)[off];
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = yv[e] > 0.f ? d[e] : d[e] * slope;
                    s += d;
                }
            reinterpret_cast<f32x4*>(din)[i] = s;
            if (nt) {
                const f32x4 cvv = reinterpret_cast<const f32x4*>(tc[0])[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1[e] += (double)s[e];
                    s2[0][e] += (double)s[e] * (double)((cvv[e] - mu[0][e]) * is[0][e]);
                }
            }
        }
    }
    if (nt == 0) return;                                    // uniform over the segment's blocks
    // block reduction per channel group (threads tid % CV == cv), one quantity at a time; then fp64 atomics into the
    // slots of every target: sum g is shared by the targets of segment 0
    const int pb = blk % a.ns;
#pragma unroll
    for (int qn = 0; qn < 4; ++qn) {                        // (unrolled: statically indexed accumulators)
        if (qn > nt) break;
        d4s v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v.v[e] = qn == 0 ? s1[e] : s2[qn == 0 ? 0 : qn - 1][e];
        red[tid] = v;
        __syncthreads();
        if (tid < CV) {
            d4s acc = red[tid];
            for (int k = tid + CV; k < 256; k += CV)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc.v[e] += red[k].v[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = tid * 4 + e;
                if (qn == 0) {
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        if (t < nt) atomicAdd(tsl[t] + (int64_t)pb * C + ch, acc.v[e]);       // slot-major: [2][ns][C]
                } else {
                    atomicAdd(tsl[qn == 0 ? 0 : qn - 1] + ((int64_t)a.ns + pb) * C + ch, acc.v[e]);
                }
            }
        }
        __syncthreads();
    }
}

// 3x3 stride-2 pad-1 max pool; idx = first maximum in (kh, kw) scan order (torch semantics)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      uint8_t* __restrict__ idx, int N, int H, int W, int C, int Ho,
                                                      int Wo) {
    const int64_t total = (int64_t)N * Ho * Wo * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        int64_t pix = i / C;
        int wo = (int)(pix % Wo);
        int64_t q = pix / Wo;
        int ho = (int)(q % Ho);
        int n = (int)(q / Ho);
        float best = -INFINITY;
        int bi = 0;
        bool first = true;
        for (int kh = 0; kh < 3; ++kh) {
            int h = ho * 2 - 1 + kh;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                int w = wo * 2 - 1 + kw;
                if ((unsigned)w >= (unsigned)W) continue;
                float v = x[(((int64_t)n * H + h) * W + w) * C + c];
                if (first || v > best || v != v) { best = v; bi = kh * 3 + kw; first = false; }
            }
        }
        y[i] = best;
        idx[i] = (uint8_t)bi;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, int N, int H, int W, int C, int Ho,
                                                          int Wo) {
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        int64_t pix = i / C;
        int w = (int)(pix % W);
        int64_t q = pix / W;
        int h = (int)(q % H);
        int n = (int)(q / H);
        float s = 0.f;
        // windows (ho, kh) with ho*2 - 1 + kh == h
        for (int kh = 0; kh < 3; ++kh) {
            int t = h + 1 - kh;
            if (t < 0 || (t & 1)) continue;
            int ho = t >> 1;
            if (ho >= Ho) continue;
            for (int kw = 0; kw < 3; ++kw) {
                int u = w + 1 - kw;
                if (u < 0 || (u & 1)) continue;
                int wo = u >> 1;
                if (wo >= Wo) continue;
                int64_t o = (((int64_t)n * Ho + ho) * Wo + wo) * C + c;
                if (idx[o] == kh * 3 + kw) s += dy[o];
            }
        }
        dx[i] = s;
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ a, const float* __restrict__ b, float alpha,
                                                   int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        a[i] += alpha * b[i];
}

// out = a + b, 4 floats per thread when n % 4 == 0 (gradient fan-in inside a launch chain)
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, int64_t n, int vec) {
    if (vec) {
        const int64_t n4 = n >> 2;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
            reinterpret_cast<f32x4*>(out)[i] =
                reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
    } else {
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            out[i] = a[i] + b[i];
    }
}

__global__ __launch_bounds__(256) void scale_dev_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                        const float* __restrict__ s_dev, float s_host, int64_t n) {
    const float s = (s_dev ? *s_dev : 1.0f) * s_host;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = x[i] * s;
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, float v, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = v;
}

}  // namespace

extern "C" int advmix_act_copy(const float* x, int ldx, float* y, int ldy, int64_t rows, int C, int act,
                               void* stream) {
    if (!x || !y || rows <= 0 || C <= 0 || ldx < C || ldy < C) return ADVMIX_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
    if (vec)
        hipLaunchKernelGGL((act_copy_kernel<true>), dim3(stream_blocks(rows * (C / 4))), dim3(256), 0, st, x, ldx, y,
                           ldy, rows, C, act);
    else
        hipLaunchKernelGGL((act_copy_kernel<false>), dim3(stream_blocks(rows * C)), dim3(256), 0, st, x, ldx, y, ldy,
                           rows, C, act);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_act_bwd(const float* dy, int lddy, const float* y, int ldy, float* dx, int lddx,
                              int64_t rows, int C, int act, void* stream) {
    if (!dy || !y || !dx || rows <= 0 || C <= 0) return ADVMIX_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    bool vec = (C % 4 == 0);
    if (vec)
        hipLaunchKernelGGL((act_bwd_kernel<true>), dim3(stream_blocks(rows * (C / 4))), dim3(256), 0, st, dy, lddy, y,
                           ldy, dx, lddx, rows, C, act);
    else
        hipLaunchKernelGGL((act_bwd_kernel<false>), dim3(stream_blocks(rows * C)), dim3(256), 0, st, dy, lddy, y, ldy,
                           dx, lddx, rows, C, act);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_fuse_sum(const float* const* ins, const int* shifts, int n_in, float* y, int N, int H, int W,
                               int C, int act, void* stream) {
    if (!ins || !shifts || !y || n_in < 1 || n_in > 4 || C % 4 != 0) return ADVMIX_EINVAL;
    FuseArgs a{};
    a.n = n_in;
    for (int j = 0; j < n_in; ++j) {
        if (!ins[j] || shifts[j] < 0 || (H % (1 << shifts[j])) || (W % (1 << shifts[j]))) return ADVMIX_EINVAL;
        a.in[j] = ins[j];
        a.shift[j] = shifts[j];
    }
    int64_t total = (int64_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(fuse_sum_kernel, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, a, y, N, H, W,
                       C / 4, act);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_fuse_sum_bwd(const float* dy, const float* y, float* g_out, float* const* dins,
                                   const int* shifts, int n_in, int N, int H, int W, int C, int act, void* stream) {
    if (!dy || !y || !g_out || !dins || !shifts || n_in < 1 || n_in > 4 || C % 4 != 0) return ADVMIX_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int64_t n4 = (int64_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(mask_grad_kernel, dim3(stream_blocks(n4)), dim3(256), 0, st, dy, y, g_out, n4, act);
    for (int j = 0; j < n_in; ++j) {
        if (shifts[j] == 0 || !dins[j]) continue;
        int64_t t = n4 >> (2 * shifts[j]);
        hipLaunchKernelGGL(pool_sum_kernel, dim3(stream_blocks(t)), dim3(256), 0, st, g_out, dins[j], N, H, W, C / 4,
                           shifts[j]);
    }
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// advmix_fuse_sum_bwd in one launch + the BatchNorm-backward channel sums of the sources that are conv + BN outputs.
// bnb_c[j] != null: source j is the output (no activation) of a train-mode conv + BN whose raw conv output is bnb_c[j],
// saved statistics bnb_mean[j] / bnb_invstd[j]; its sums are ADDED to bnb_slots[j] ([2][ns][C] fp64, pre-zeroed).
// g_out may be null when no same-resolution source needs a gradient.  ADVMIX_EINVAL: shape not served (C / 4 must
// divide 256, at most three same-resolution targets) - nothing launched, use advmix_fuse_sum_bwd.
extern "C" int advmix_fuse_sum_bwd_bnb(const float* dy, const float* y, float* g_out, float* const* dins,
                                       const int* shifts, int n_in, int N, int H, int W, int C, int act,
                                       const float* const* bnb_c, const float* const* bnb_mean,
                                       const float* const* bnb_invstd, double* const* bnb_slots, int ns, void* stream) {
    if (!dy || !y || !dins || !shifts || n_in < 1 || n_in > 4 || C % 4 != 0 || !bnb_c || !bnb_mean || !bnb_invstd ||
        !bnb_slots)
        return ADVMIX_EINVAL;
    const int CV = C / 4;
    if (CV > 256 || 256 % CV != 0 || ns < 1 || ns > ADVMIX_STAT_SLOTS_MAX || (ns & (ns - 1))) return ADVMIX_EINVAL;
    FuseBwdArgs a{};
    a.ns = ns;
    a.n = 1;
    a.shift[0] = 0;
    a.out[0] = g_out;
    int64_t work[5];
    work[0] = (int64_t)N * H * W * CV;
    bool need0 = g_out != nullptr;
    for (int j = 0; j < n_in; ++j) {
        if (shifts[j] < 0 || (H % (1 << shifts[j])) || (W % (1 << shifts[j]))) return ADVMIX_EINVAL;
        const bool tgt = bnb_c[j] != nullptr;
        if (tgt && (!bnb_mean[j] || !bnb_invstd[j] || !bnb_slots[j])) return ADVMIX_EINVAL;
        if (shifts[j] == 0) {
            if (!tgt) continue;
            if (a.nt0 >= 3 || !g_out) return ADVMIX_EINVAL;
            a.tc0[a.nt0] = bnb_c[j]; a.tmean0[a.nt0] = bnb_mean[j]; a.tinvstd0[a.nt0] = bnb_invstd[j];
            a.tslots0[a.nt0] = bnb_slots[j];
            ++a.nt0;
        } else {
            if (!dins[j]) continue;
            const int k = a.n++;
            a.shift[k] = shifts[j];
            a.out[k] = dins[j];
            work[k] = work[0] >> (2 * shifts[j]);
            if (tgt) { a.tc[k] = bnb_c[j]; a.tmean[k] = bnb_mean[j]; a.tinvstd[k] = bnb_invstd[j]; a.tslots[k] = bnb_slots[j]; }
        }
    }
    int start = 0;
    for (int k = 0; k < a.n; ++k) {
        a.start[k] = start;
        if (k == 0 && !need0) continue;                     // nobody needs g: segment 0 gets no blocks
        // pooled segments read f x f inputs per output: fewer elements per block
        int64_t per = k == 0 ? 1024 : (1024 >> (2 * a.shift[k]) > 64 ? 1024 >> (2 * a.shift[k]) : 64);
        int64_t b = (work[k] + per - 1) / per;
        if (b > 1024) b = 1024;
        if (b < 1) b = 1;
        start += (int)b;
    }
    for (int k = a.n; k < 6; ++k) a.start[k] = start;
    if (start == 0) return ADVMIX_OK;
    hipLaunchKernelGGL(fuse_bwd_kernel, dim3(start), dim3(256), 0, (hipStream_t)stream, a, dy, y, N, H, W, CV, act);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_maxpool3x3s2(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, int Ho,
                                   int Wo, void* stream) {
    if (!x || !y || !idx || Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return ADVMIX_EINVAL;
    int64_t total = (int64_t)N * Ho * Wo * C;
    hipLaunchKernelGGL(maxpool_kernel, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, idx, N, H,
                       W, C, Ho, Wo);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int N, int H, int W, int C,
                                       int Ho, int Wo, void* stream) {
    if (!dy || !idx || !dx) return ADVMIX_EINVAL;
    int64_t total = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx,
                       N, H, W, C, Ho, Wo);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_axpy(float* a, const float* b, float alpha, int64_t n, void* stream) {
    if (!a || !b || n < 0) return ADVMIX_EINVAL;
    if (n == 0) return ADVMIX_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(stream_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, alpha, n);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n < 0) return ADVMIX_EINVAL;
    if (n == 0) return ADVMIX_OK;
    const int vec = (n % 4 == 0) && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0);
    hipLaunchKernelGGL(add_kernel, dim3(stream_blocks(vec ? n / 4 : n)), dim3(256), 0, (hipStream_t)stream, a, b, out,
                       n, vec);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_fill(float* p, float value, int64_t n, void* stream) {
    if (!p || n < 0) return ADVMIX_EINVAL;
    if (n == 0) return ADVMIX_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(stream_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, value, n);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_scale_dev(float* y, const float* x, const float* s_dev, float s_host, int64_t n, void* stream) {
    if (!y || !x || n < 0) return ADVMIX_EINVAL;
    if (n == 0) return ADVMIX_OK;
    hipLaunchKernelGGL(scale_dev_kernel, dim3(stream_blocks(n)), dim3(256), 0, (hipStream_t)stream, y, x, s_dev, s_host,
                       n);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}
