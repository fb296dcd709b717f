// BatchNorm2d (train / eval) and InstanceNorm2d(affine=False) on NHWC fp32, forward + backward.
// HBM-bound: each kernel streams rows with 16-byte lanes; per-channel statistics are accumulated
// in fp64 (block partials -> fp64 finalize): deterministic (no atomics), and E[x^2]-E[x]^2 keeps
// its precision when |mean| >> std.
//
// groups = 1  : BatchNorm over all rows      (lib/models/pose_hrnet.py:34 and every nn.BatchNorm2d)
// groups = N  : InstanceNorm per image       (lib/models/Unet_generator.py:19,43,45)
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int MAX_PARTIAL_BLOCKS = 512;

struct RowSplit { int nbg; int64_t rows_per_block; };

static RowSplit split_rows(int groups, int64_t Mg) {
    int cap = MAX_PARTIAL_BLOCKS / groups;
    if (cap < 1) cap = 1;
    int64_t nb = (Mg + 63) / 64;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    RowSplit s;
    s.rows_per_block = (Mg + nb - 1) / nb;
    s.nbg = (int)((Mg + s.rows_per_block - 1) / s.rows_per_block);
    return s;
}

struct d4 { double v[4]; };
__device__ __forceinline__ void d4_zero(d4& a) { a.v[0] = a.v[1] = a.v[2] = a.v[3] = 0.0; }
__device__ __forceinline__ void d4_add(d4& a, const d4& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) a.v[e] += b.v[e];
}

// KIND 0: (x, x^2)     KIND 1: (g, g*xhat) with g = dy*act'(y).   Accumulated in fp64: the
// variance is later formed as E[x^2]-E[x]^2, which needs ~2x the input precision when
// |mean| >> std (fp32 partials lose it: relative variance error ~1e-7*(1+mean^2/var)).
// slope of the activation through its OUTPUT y, with the activation known at compile time (a run-time ``act``
// put a scalar branch and an s_waitcnt vmcnt(0) around every element of the hot loops)
template <int ACT>
__device__ __forceinline__ float act_slope(float y) {
    if (ACT == ADVMIX_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (ACT == ADVMIX_ACT_LEAKY02) return y > 0.f ? 1.f : 0.2f;
    return 1.f;
}

// Several rows per thread with ALL their loads issued before the first use (the row loop of the first version
// waited for each row's loads before touching the next: 6 dependent memory round trips per thread).
template <int KIND, int ACT, int U>
__device__ __forceinline__ void accum_rows(d4& s0, d4& s1, const float* x, const float* dy, const float* y,
                                           int64_t r, int64_t rstep, int64_t r1, int ldy, int C, int c,
                                           const f32x4& mu, const f32x4& is) {
    f32x4 xv[U], gv[U], yv[U];
    float ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t row = r + u * rstep;
        ok[u] = row < r1 ? 1.f : 0.f;
        const int64_t rr = row < r1 ? row : r;               // a valid address; its contribution is masked
        xv[u] = *reinterpret_cast<const f32x4*>(x + rr * C + c);
        if (KIND == 1) {
            gv[u] = *reinterpret_cast<const f32x4*>(dy + rr * ldy + c);
            if (ACT != ADVMIX_ACT_NONE) yv[u] = *reinterpret_cast<const f32x4*>(y + rr * ldy + c);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (KIND == 0) {
                const double d = (double)(xv[u][e] * ok[u]);
                s0.v[e] += d;
                s1.v[e] += d * d;
            } else {
                float g = gv[u][e] * ok[u];
                if (ACT != ADVMIX_ACT_NONE) g *= act_slope<ACT>(yv[u][e]);
                s0.v[e] += (double)g;
                s1.v[e] += (double)g * (double)((xv[u][e] - mu[e]) * is[e]);
            }
        }
    }
}

template <int KIND>
__device__ __forceinline__ void accum4(d4& s0, d4& s1, const float* x, const float* dy, const float* y,
                                       int64_t row, int ldy, int C, int c, const f32x4& mu, const f32x4& is,
                                       int act) {
    f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * C + c);
    if (KIND == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double d = (double)xv[e];
            s0.v[e] += d;
            s1.v[e] += d * d;
        }
    } else {
        f32x4 g = *reinterpret_cast<const f32x4*>(dy + row * ldy + c);
        if (act != ADVMIX_ACT_NONE) {
            f32x4 yv = *reinterpret_cast<const f32x4*>(y + row * ldy + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] *= act_grad(yv[e], act);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s0.v[e] += (double)g[e];
            s1.v[e] += (double)g[e] * (double)((xv[e] - mu[e]) * is[e]);
        }
    }
}

// partial[g][blk][2][C] (double)
template <int KIND, int ACT>
__global__ __launch_bounds__(256) void norm_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                    const float* __restrict__ y, int ldy,
                                                    const float* __restrict__ mean,
                                                    const float* __restrict__ invstd, double* __restrict__ partial,
                                                    int64_t Mg, int C, int64_t rows_per_block, int act) {
    __shared__ d4 red[2][256];
    const int g = blockIdx.y, blk = blockIdx.x, nbg = gridDim.x;
    const int64_t r0 = (int64_t)g * Mg + blk * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    const int64_t gend = (int64_t)(g + 1) * Mg;
    if (r1 > gend) r1 = gend;
    const int CV = C >> 2;
    double* out = partial + (int64_t)g * 2 * C * nbg + blk;      // [g][2][C][nbg]: element (k, c) at (k*C + c)*nbg
    const int tid = threadIdx.x;
    if (CV >= 256) {
        for (int cv = tid; cv < CV; cv += 256) {
            d4 s0, s1;
            d4_zero(s0); d4_zero(s1);
            f32x4 mu = {0, 0, 0, 0}, is = {0, 0, 0, 0};
            if (KIND == 1) {
                mu = *reinterpret_cast<const f32x4*>(mean + (int64_t)g * C + cv * 4);
                is = *reinterpret_cast<const f32x4*>(invstd + (int64_t)g * C + cv * 4);
            }
            for (int64_t r = r0; r < r1; r += 4)
                accum_rows<KIND, ACT, 4>(s0, s1, x, dy, y, r, 1, r1, ldy, C, cv * 4, mu, is);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                out[(int64_t)(cv * 4 + e) * nbg] = s0.v[e];
                out[(int64_t)(C + cv * 4 + e) * nbg] = s1.v[e];
            }
        }
        return;
    }
    const int RP = 256 / CV;
    const int rr = tid / CV, cv = tid - rr * CV;
    d4 s0, s1;
    d4_zero(s0); d4_zero(s1);
    if (rr < RP) {
        f32x4 mu = {0, 0, 0, 0}, is = {0, 0, 0, 0};
        if (KIND == 1) {
            mu = *reinterpret_cast<const f32x4*>(mean + (int64_t)g * C + cv * 4);
            is = *reinterpret_cast<const f32x4*>(invstd + (int64_t)g * C + cv * 4);
        }
        for (int64_t r = r0 + rr; r < r1; r += 3 * (int64_t)RP)
            accum_rows<KIND, ACT, 3>(s0, s1, x, dy, y, r, RP, r1, ldy, C, cv * 4, mu, is);
    }
    red[0][tid] = s0;
    red[1][tid] = s1;
    __syncthreads();
    // tree over the RP row-lanes that share a column vector
    for (int step = 1; step < RP; step <<= 1) {
        if (rr < RP && (rr % (2 * step)) == 0 && rr + step < RP) {
            d4_add(red[0][tid], red[0][tid + step * CV]);
            d4_add(red[1][tid], red[1][tid + step * CV]);
        }
        __syncthreads();
    }
    if (rr == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            out[(int64_t)(cv * 4 + e) * nbg] = red[0][tid].v[e];
            out[(int64_t)(C + cv * 4 + e) * nbg] = red[1][tid].v[e];
        }
    }
}

// scalar fallback for C % 4 != 0 (one thread per channel, serial rows; tiny shapes only)
template <int KIND>
__global__ void norm_partial_scalar(const float* __restrict__ x, const float* __restrict__ dy,
                                    const float* __restrict__ y, int ldy, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, double* __restrict__ partial, int64_t Mg,
                                    int C, int64_t rows_per_block, int act) {
    const int g = blockIdx.y, blk = blockIdx.x, nbg = gridDim.x;
    const int64_t r0 = (int64_t)g * Mg + blk * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    const int64_t gend = (int64_t)(g + 1) * Mg;
    if (r1 > gend) r1 = gend;
    double* out = partial + (int64_t)g * 2 * C * nbg + blk;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double s0 = 0.0, s1 = 0.0;
        float mu = KIND ? mean[(int64_t)g * C + c] : 0.f, is = KIND ? invstd[(int64_t)g * C + c] : 0.f;
        for (int64_t r = r0; r < r1; ++r) {
            float xv = x[r * C + c];
            if (KIND == 0) { s0 += (double)xv; s1 += (double)xv * (double)xv; }
            else {
                float gg = dy[r * ldy + c] * act_grad(act != ADVMIX_ACT_NONE ? y[r * ldy + c] : 1.f, act);
                s0 += (double)gg; s1 += (double)gg * (double)((xv - mu) * is);
            }
        }
        out[(int64_t)c * nbg] = s0;
        out[(int64_t)(C + c) * nbg] = s1;
    }
}

// One wave per (group, channel): lanes stride over the block partials, then a wave reduction.
// (Measured and rejected: letting the last-arriving partial block do this in the same launch -
// sc1 partial stores + ticket + acquire, placement-independent and correct, but the serial tail on
// one CU costs more than the launch it saves: 108.6 vs 87.7 ms per AdvMix step.)
// ``sm`` (slot-major): the conv epilogues' slots[2][nbg][C] (groups == 1) instead of norm_partial's [g][2][C][nbg].
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partial, int nbg, int g, int c, int C,
                                                int lane, double& s, double& ss, bool sm = false) {
    s = 0.0; ss = 0.0;
    const double* p0 = sm ? partial + c : partial + ((int64_t)g * 2 * C + c) * nbg;
    const double* p1 = sm ? partial + (int64_t)nbg * C + c : partial + ((int64_t)g * 2 * C + C + c) * nbg;
    const int64_t str = sm ? C : 1;
    for (int b = lane; b < nbg; b += 64) {
        s += p0[b * str];
        ss += p1[b * str];
    }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
}

__global__ __launch_bounds__(256) void norm_finalize_fwd(const double* __restrict__ partial, int nbg, int groups,
                                                         int64_t Mg, int C, float eps, float* __restrict__ mean,
                                                         float* __restrict__ invstd, float* running_mean,
                                                         float* running_var, int64_t* nbt, float momentum,
                                                         double* zero_after) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i == 0 && lane == 0 && nbt) *nbt += 1;
    if (i >= groups * C) return;
    const int g = i / C, c = i - g * C;
    double s, ss;
    reduce_partials(partial, nbg, g, c, C, lane, s, ss, zero_after != nullptr);
    if (zero_after) {                                    // atomically accumulated slots ([2][nbg][C]): leave them clean
        for (int b = lane; b < nbg; b += 64) {
            zero_after[(int64_t)b * C + c] = 0.0;
            zero_after[((int64_t)nbg + b) * C + c] = 0.0;
        }
    }
    if (lane != 0) return;
    double m = s / (double)Mg;
    double var = ss / (double)Mg - m * m;
    if (var < 0) var = 0;
    mean[i] = (float)m;
    invstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {   // groups == 1
        double unb = Mg > 1 ? var * (double)Mg / (double)(Mg - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + (double)momentum * m);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + (double)momentum * unb);
    }
}

// coef[g][0][C] = mean(g), coef[g][1][C] = mean(g*xhat); dgamma/dbeta += (groups == 1)
__global__ __launch_bounds__(256) void norm_finalize_bwd(const double* __restrict__ partial, int nbg, int groups,
                                                         int64_t Mg, int C, float* __restrict__ coef,
                                                         float* dgamma, float* dbeta) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= groups * C) return;
    const int g = i / C, c = i - g * C;
    double s, ss;
    reduce_partials(partial, nbg, g, c, C, lane, s, ss);
    if (lane != 0) return;
    coef[((int64_t)g * 2) * C + c] = (float)(s / (double)Mg);
    coef[((int64_t)g * 2 + 1) * C + c] = (float)(ss / (double)Mg);
    if (dbeta) dbeta[c] += (float)s;
    if (dgamma) dgamma[c] += (float)ss;
}

template <bool VEC>
__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const float* __restrict__ res, float* __restrict__ y,
                                                         int ldy, int64_t Mg, int64_t rows, int C, int act) {
    constexpr int V = VEC ? 4 : 1;
    const int CV = C / V;
    const int64_t total = rows * CV;
    const bool small = total < 0x7fffffff;                 // (a 64-bit division is ~100 instructions of software)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = small ? (int64_t)((unsigned)i / (unsigned)CV) : i / CV;   // 32-bit divide when it fits
        int c = (int)(i - r * CV) * V;
        int64_t gc = (small ? (int64_t)((unsigned)r / (unsigned)Mg) : r / Mg) * C + c;
        if (VEC) {
            f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * C + c);
            f32x4 mu = *reinterpret_cast<const f32x4*>(mean + gc);
            f32x4 is = *reinterpret_cast<const f32x4*>(invstd + gc);
            f32x4 o = (xv - mu) * is;
            if (gamma) o = o * *reinterpret_cast<const f32x4*>(gamma + c) + *reinterpret_cast<const f32x4*>(beta + c);
            if (res) o += *reinterpret_cast<const f32x4*>(res + r * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = act_fwd(o[e], act);
            *reinterpret_cast<f32x4*>(y + r * ldy + c) = o;
        } else {
            float o = (x[r * C + c] - mean[gc]) * invstd[gc];
            if (gamma) o = o * gamma[c] + beta[c];
            if (res) o += res[r * C + c];
            y[r * ldy + c] = act_fwd(o, act);
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_eval_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ rm,
                                                      const float* __restrict__ rv, float eps,
                                                      const float* __restrict__ res, float* __restrict__ y,
                                                      int64_t rows, int C, int act) {
    constexpr int V = VEC ? 4 : 1;
    const int CV = C / V;
    const int64_t total = rows * CV;
    const bool small = total < 0x7fffffff;                 // (a 64-bit division is ~100 instructions of software)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = small ? (int64_t)((unsigned)i / (unsigned)CV) : i / CV;   // 32-bit divide when it fits
        int c = (int)(i - r * CV) * V;
        if (VEC) {
            f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * C + c);
            f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
            f32x4 m = *reinterpret_cast<const f32x4*>(rm + c), v = *reinterpret_cast<const f32x4*>(rv + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (xv[e] - m[e]) * (1.0f / sqrtf(v[e] + eps)) * g[e] + bt[e];
            if (res) o += *reinterpret_cast<const f32x4*>(res + r * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = act_fwd(o[e], act);
            *reinterpret_cast<f32x4*>(y + r * C + c) = o;
        } else {
            float is = 1.0f / sqrtf(rv[c] + eps);
            float o = (x[r * C + c] - rm[c]) * is * gamma[c] + beta[c];
            if (res) o += res[r * C + c];
            y[r * C + c] = act_fwd(o, act);
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void norm_bwd_apply(const float* __restrict__ dy, const float* __restrict__ y,
                                                      int ldy, const float* __restrict__ x,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ coef, float* __restrict__ dx,
                                                      float* __restrict__ dres, int64_t Mg, int64_t rows, int C,
                                                      int act) {
    constexpr int V = VEC ? 4 : 1;
    const int CV = C / V;
    const int64_t total = rows * CV;
    const bool small = total < 0x7fffffff;                 // (a 64-bit division is ~100 instructions of software)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = small ? (int64_t)((unsigned)i / (unsigned)CV) : i / CV;   // 32-bit divide when it fits
        int c = (int)(i - r * CV) * V;
        int64_t g = small ? (int64_t)((unsigned)r / (unsigned)Mg) : r / Mg;
        if (VEC) {
            f32x4 gg = *reinterpret_cast<const f32x4*>(dy + r * ldy + c);
            if (act != ADVMIX_ACT_NONE) {
                f32x4 yv = *reinterpret_cast<const f32x4*>(y + r * ldy + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) gg[e] *= act_grad(yv[e], act);
            }
            f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * C + c);
            f32x4 is = *reinterpret_cast<const f32x4*>(invstd + g * C + c);
            f32x4 mu = *reinterpret_cast<const f32x4*>(mean + g * C + c);
            f32x4 c1 = *reinterpret_cast<const f32x4*>(coef + (g * 2) * C + c);
            f32x4 c2 = *reinterpret_cast<const f32x4*>(coef + (g * 2 + 1) * C + c);
            f32x4 k = is;
            if (gamma) k = k * *reinterpret_cast<const f32x4*>(gamma + c);
            f32x4 xh = (xv - mu) * is;
            *reinterpret_cast<f32x4*>(dx + r * C + c) = k * (gg - c1 - xh * c2);
            if (dres) *reinterpret_cast<f32x4*>(dres + r * C + c) = gg;
        } else {
            float gg = dy[r * ldy + c];
            if (act != ADVMIX_ACT_NONE) gg *= act_grad(y[r * ldy + c], act);
            float is = invstd[g * C + c];
            float xh = (x[r * C + c] - mean[g * C + c]) * is;
            float k = gamma ? gamma[c] * is : is;
            dx[r * C + c] = k * (gg - coef[(g * 2) * C + c] - xh * coef[(g * 2 + 1) * C + c]);
            if (dres) dres[r * C + c] = gg;
        }
    }
}


// ---- statistics folded into their consumer ------------------------------------------------------------------------
// The conv epilogues (conv_direct: forward column sums; transposed gather: BatchNorm-backward sums) add their
// per-workgroup fp64 partial sums into ``ns`` slots per channel, slots[2][ns][C].  The kernels below reduce the slots
// THEMSELVES (every workgroup, for the <= 64 channels it streams) instead of waiting for a one-wave-per-channel
// finalize launch in between: 2 x 593 launches of ~4.9 us per AdvMix step were 6.9 % of the kernel time and sat on the
// critical path of every conv -> BN -> conv chain.  The slots are per layer and zero-filled once per network pass
// (ops.py), so nobody re-zeroes them in-kernel.
constexpr int SLOT_CT = 64;                              // channels per workgroup (channel tile)

// sums[k][ch] = sum over the ns slots of statistic k of channel ct0 + ch (ch < ctn <= 64); ns a power of two <= 64.
// Slot-major layout (round 4), slots[2][ns][C]: every (statistic, channel) pair is summed by 256 / pairs threads ("parts"),
// each taking ITS run of slots - one 8-byte load per slot, a wave's loads of one slot contiguous over the channels - all
// issued before the first add (a load -> shuffle loop waited one L2 round trip per iteration: 18.7 instead of 6.3 us for
// the whole apply launch at 64 slots), then the parts meet in LDS.
__device__ __forceinline__ void reduce_slots(const double* __restrict__ slots, int ns, int C, int ct0, int ctn,
                                             double (*sums)[SLOT_CT], double* red /* [4][2 * SLOT_CT] */) {
    const int tid = threadIdx.x;
    const int pairs = 2 * ctn;                           // <= 128
    int parts = 256 / pairs;                             // >= 2
    if (parts > 4) parts = 4;
    if (parts > ns) parts = ns;
    const int per = ns / parts;                          // slots per part (ns and parts are powers of two): <= 32
    const int pr = tid % pairs, part = tid / pairs;
    if (part < parts) {
        const int k = pr / ctn, ch = pr - k * ctn;
        const double* src = slots + ((int64_t)k * ns + part * per) * C + ct0 + ch;
        double v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if (i < per) v[i] = src[(int64_t)i * C];
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if (i < per) acc += v[i];
        red[part * (2 * SLOT_CT) + pr] = acc;
    }
    __syncthreads();
    if (tid < pairs) {
        double a = red[tid];
        for (int q = 1; q < parts; ++q) a += red[q * (2 * SLOT_CT) + tid];
        sums[tid / ctn][tid % ctn] = a;
    }
}

struct RowTile { int64_t r0, r1; int tpr, rpi, cv, rr; };
__device__ __forceinline__ RowTile row_tile(int64_t rows, int ctn) {
    RowTile t;
    const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    t.r0 = (int64_t)blockIdx.x * per;
    t.r1 = t.r0 + per < rows ? t.r0 + per : rows;
    t.tpr = ctn >> 2;                                    // threads per row slice (float4 each)
    t.rpi = 256 / t.tpr;                                 // rows per iteration
    t.cv = threadIdx.x % t.tpr;
    t.rr = threadIdx.x / t.tpr;
    return t;
}

// y = act((c - mean) * invstd * gamma + beta + res), train mode, statistics from the conv epilogue's slots.
// Workgroup (0, ct) also publishes mean / invstd (saved for backward) and updates the running statistics.
__global__ __launch_bounds__(256) void norm_apply_slots_kernel(
        const float* __restrict__ c, const double* __restrict__ slots, int ns, int64_t rows, int C, float eps,
        const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ res,
        float* __restrict__ y, int act, float* __restrict__ mean_out, float* __restrict__ invstd_out,
        float* running_mean, float* running_var, int64_t* nbt, float momentum, unsigned char* __restrict__ mask) {
    __shared__ double sums[2][SLOT_CT];
    __shared__ double red[4 * 2 * SLOT_CT];
    __shared__ float smu[SLOT_CT], sis[SLOT_CT];
    const int tid = threadIdx.x;
    const int ct0 = blockIdx.y * SLOT_CT;
    const int ctn = C - ct0 < SLOT_CT ? C - ct0 : SLOT_CT;
    reduce_slots(slots, ns, C, ct0, ctn, sums, red);
    __syncthreads();
    if (tid < ctn) {
        const double m = sums[0][tid] / (double)rows;
        double var = sums[1][tid] / (double)rows - m * m;
        if (var < 0) var = 0;
        const float mu = (float)m, is = (float)(1.0 / sqrt(var + (double)eps));
        smu[tid] = mu;
        sis[tid] = is;
        if (blockIdx.x == 0) {
            const int ch = ct0 + tid;
            mean_out[ch] = mu;
            invstd_out[ch] = is;
            if (running_mean) {
                const double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
                running_mean[ch] = (float)((1.0 - momentum) * (double)running_mean[ch] + (double)momentum * m);
                running_var[ch] = (float)((1.0 - momentum) * (double)running_var[ch] + (double)momentum * unb);
            }
            if (ch == 0 && nbt) *nbt += 1;
        }
    }
    __syncthreads();
    const RowTile t = row_tile(rows, ctn);
    if (t.rr >= t.rpi) return;
    const int cl = t.cv * 4, ch = ct0 + cl;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(&smu[cl]), is = *reinterpret_cast<const f32x4*>(&sis[cl]);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + ch), b = *reinterpret_cast<const f32x4*>(beta + ch);
    constexpr int U = 4;
    for (int64_t r = t.r0 + t.rr; r < t.r1; r += (int64_t)U * t.rpi) {
        f32x4 xv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = r + (int64_t)u * t.rpi;
            const int64_t rr_ = row < t.r1 ? row : r;
            xv[u] = *reinterpret_cast<const f32x4*>(c + rr_ * C + ch);
            if (res) rv[u] = *reinterpret_cast<const f32x4*>(res + rr_ * C + ch);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = r + (int64_t)u * t.rpi;
            if (row >= t.r1) break;
            f32x4 o = (xv[u] - mu) * is;
            // ONE fused multiply-add, spelled out: the BatchNorm-backward epilogue of the consumer's input-gradient conv
            // (conv_direct.hip) recomputes the sign of y from c with this very expression when there is no residual
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(o[e], g[e], b[e]);
            if (res) o += rv[u];
            if (mask) {                                     // a bit per element, a byte per 4 channels: 1/16 of y's bytes
                const unsigned nib = (o[0] > 0.f ? 1u : 0u) | (o[1] > 0.f ? 2u : 0u) | (o[2] > 0.f ? 4u : 0u) | (o[3] > 0.f ? 8u : 0u);
                mask[(row * C + ch) >> 2] = (unsigned char)nib;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = act_fwd(o[e], act);
            *reinterpret_cast<f32x4*>(y + row * C + ch) = o;
        }
    }
}

// dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)) with g ALREADY multiplied by the activation's slope
// (the producing input-gradient conv did that in its epilogue and left sum(g), sum(g * xhat) in the slots).
// Workgroup (0, ct) also accumulates dgamma / dbeta.
__global__ __launch_bounds__(256) void norm_bwd_apply_slots_kernel(
        const float* __restrict__ g, const float* __restrict__ c, const float* __restrict__ mean,
        const float* __restrict__ invstd, const float* __restrict__ gamma, const double* __restrict__ slots, int ns,
        int64_t rows, int C, float* __restrict__ dx, float* dgamma, float* dbeta) {
    __shared__ double sums[2][SLOT_CT];
    __shared__ double red[4 * 2 * SLOT_CT];
    __shared__ float sc1[SLOT_CT], sc2[SLOT_CT];
    const int tid = threadIdx.x;
    const int ct0 = blockIdx.y * SLOT_CT;
    const int ctn = C - ct0 < SLOT_CT ? C - ct0 : SLOT_CT;
    reduce_slots(slots, ns, C, ct0, ctn, sums, red);
    __syncthreads();
    if (tid < ctn) {
        sc1[tid] = (float)(sums[0][tid] / (double)rows);
        sc2[tid] = (float)(sums[1][tid] / (double)rows);
        if (blockIdx.x == 0) {
            if (dbeta) dbeta[ct0 + tid] += (float)sums[0][tid];
            if (dgamma) dgamma[ct0 + tid] += (float)sums[1][tid];
        }
    }
    __syncthreads();
    const RowTile t = row_tile(rows, ctn);
    if (t.rr >= t.rpi) return;
    const int cl = t.cv * 4, ch = ct0 + cl;
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(&sc1[cl]), c2 = *reinterpret_cast<const f32x4*>(&sc2[cl]);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + ch), is = *reinterpret_cast<const f32x4*>(invstd + ch);
    f32x4 k = is;
    if (gamma) k = k * *reinterpret_cast<const f32x4*>(gamma + ch);
    constexpr int U = 4;
    for (int64_t r = t.r0 + t.rr; r < t.r1; r += (int64_t)U * t.rpi) {
        f32x4 gv[U], xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = r + (int64_t)u * t.rpi;
            const int64_t rr_ = row < t.r1 ? row : r;
            gv[u] = *reinterpret_cast<const f32x4*>(g + rr_ * C + ch);
            xv[u] = *reinterpret_cast<const f32x4*>(c + rr_ * C + ch);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = r + (int64_t)u * t.rpi;
            if (row >= t.r1) break;
            const f32x4 xh = (xv[u] - mu) * is;
            *reinterpret_cast<f32x4*>(dx + row * C + ch) = k * (gv[u] - c1 - xh * c2);
        }
    }
}

// Deterministic mode: the conv epilogues STORE one partial per row tile (stats[2][C][cnt], conv_direct.hip
// ConvD::stats_tiles); one wave per (statistic, channel) adds them in a fixed order - lane l takes l, l + 64, ... in
// sequence, then a fixed butterfly - and writes the total as slot 0 of the ns = 1 layout [2][1][C] the consumers read.
__global__ __launch_bounds__(256) void stats_fold_kernel(const double* __restrict__ part, int cnt, int pairs,
                                                         double* __restrict__ out) {
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pair >= pairs) return;
    const double* src = part + (int64_t)pair * cnt;
    double acc = 0.0;
    for (int i = lane; i < cnt; i += 64) acc += src[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) out[pair] = acc;
}

// grid for the slot kernels: (row blocks, channel tiles); at least 4 row-iterations each, at most ADVMIX_SLOT_WGS (512)
// workgroups: 2,048 is the fastest launch on its own (9.1 vs 8.3 us at 98,304 x 32 says otherwise even there), but in the step
// these HBM-bound launches run beside other lanes' convolutions and a narrower one leaves them the CUs
static dim3 slot_grid(int64_t rows, int C) {
    const int nct = cdiv(C, SLOT_CT);
    const int ctn = C < SLOT_CT ? C : SLOT_CT;
    const int rpi = 256 / (ctn / 4);
    int64_t nb = (rows + 4 * rpi - 1) / (4 * rpi);
    static const int wgs = advmix_env_int("ADVMIX_SLOT_WGS", 512);
    const int64_t cap = wgs / nct > 0 ? wgs / nct : 1;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return dim3((unsigned)nb, (unsigned)nct);
}

static bool slots_ok(int ns, int C) {
    return ns >= 1 && ns <= 64 && !(ns & (ns - 1)) && C % 4 == 0;   // (threads past the last whole row slice idle)
}

static int stream_blocks(int64_t total) {
    int64_t b = (total + 255) / 256;
    const int cap = advmix_stream_cap();
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int64_t advmix_norm_ws_bytes(int groups, int C) {
    return (int64_t)MAX_PARTIAL_BLOCKS * 2 * C * (int64_t)sizeof(double) +
           (int64_t)2 * groups * C * (int64_t)sizeof(float);
}

extern "C" int advmix_norm_stats(const float* x, int groups, int64_t Mg, int C, float eps, float* mean,
                                 float* invstd, float* running_mean, float* running_var, int64_t* nbt,
                                 float momentum, void* ws, void* stream) {
    if (!x || !mean || !invstd || !ws || groups <= 0 || Mg <= 0 || C <= 0) return ADVMIX_EINVAL;
    if (running_mean && groups != 1) return ADVMIX_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    RowSplit sp = split_rows(groups, Mg);
    double* partial = (double*)ws;
    dim3 g(sp.nbg, groups);
    if (C % 4 == 0)
        hipLaunchKernelGGL((norm_partial<0, ADVMIX_ACT_NONE>), g, dim3(256), 0, st, x, nullptr, nullptr, 0, nullptr,
                           nullptr, partial, Mg, C, sp.rows_per_block, 0);
    else
        hipLaunchKernelGGL((norm_partial_scalar<0>), g, dim3(256), 0, st, x, nullptr, nullptr, 0, nullptr, nullptr,
                           partial, Mg, C, sp.rows_per_block, 0);
    hipLaunchKernelGGL(norm_finalize_fwd, dim3(cdiv((int64_t)groups * C, 4)), dim3(256), 0, st, partial, sp.nbg,
                       groups, Mg, C, eps, mean, invstd, running_mean, running_var, nbt, momentum, (double*)nullptr);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_norm_finalize(double* partial, int nbg, int64_t rows, int C, float eps, float* mean,
                                    float* invstd, float* running_mean, float* running_var, int64_t* nbt,
                                    float momentum, void* stream) {
    if (!partial || nbg <= 0 || rows <= 0 || C <= 0 || !mean || !invstd) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(norm_finalize_fwd, dim3(cdiv((int64_t)C, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const double*)partial, nbg, 1, rows, C, eps, mean, invstd, running_mean, running_var, nbt,
                       momentum, partial);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_norm_apply(const float* x, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, const float* residual, float* y, int ldy, int groups,
                                 int64_t Mg, int C, int act, void* stream) {
    if (!x || !mean || !invstd || !y || ldy < C || (gamma && !beta)) return ADVMIX_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int64_t rows = (int64_t)groups * Mg;
    if (C % 4 == 0 && ldy % 4 == 0)
        hipLaunchKernelGGL((norm_apply_kernel<true>), dim3(stream_blocks(rows * (C / 4))), dim3(256), 0, st, x, mean,
                           invstd, gamma, beta, residual, y, ldy, Mg, rows, C, act);
    else
        hipLaunchKernelGGL((norm_apply_kernel<false>), dim3(stream_blocks(rows * C)), dim3(256), 0, st, x, mean,
                           invstd, gamma, beta, residual, y, ldy, Mg, rows, C, act);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_bn_eval(const float* x, const float* gamma, const float* beta, const float* rm,
                              const float* rv, float eps, const float* residual, float* y, int64_t rows, int C,
                              int act, void* stream) {
    if (!x || !gamma || !beta || !rm || !rv || !y) return ADVMIX_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0)
        hipLaunchKernelGGL((bn_eval_kernel<true>), dim3(stream_blocks(rows * (C / 4))), dim3(256), 0, st, x, gamma,
                           beta, rm, rv, eps, residual, y, rows, C, act);
    else
        hipLaunchKernelGGL((bn_eval_kernel<false>), dim3(stream_blocks(rows * C)), dim3(256), 0, st, x, gamma, beta,
                           rm, rv, eps, residual, y, rows, C, act);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_norm_bwd(const float* dy, const float* y, int ldy, const float* x, const float* mean,
                               const float* invstd, const float* gamma, float* dx, float* dres, float* dgamma,
                               float* dbeta, int groups, int64_t Mg, int C, int act, void* ws, void* stream) {
    if (!dy || !x || !mean || !invstd || !dx || !ws || ldy < C) return ADVMIX_EINVAL;
    if (act != ADVMIX_ACT_NONE && !y) return ADVMIX_EINVAL;
    if ((dgamma || dbeta) && groups != 1) return ADVMIX_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    RowSplit sp = split_rows(groups, Mg);
    double* partial = (double*)ws;
    float* coef = (float*)(partial + (int64_t)MAX_PARTIAL_BLOCKS * 2 * C);
    dim3 g(sp.nbg, groups);
    const bool vec = (C % 4 == 0) && (ldy % 4 == 0);
    if (vec && act == ADVMIX_ACT_RELU)
        hipLaunchKernelGGL((norm_partial<1, ADVMIX_ACT_RELU>), g, dim3(256), 0, st, x, dy, y, ldy, mean, invstd, partial,
                           Mg, C, sp.rows_per_block, act);
    else if (vec && act == ADVMIX_ACT_LEAKY02)
        hipLaunchKernelGGL((norm_partial<1, ADVMIX_ACT_LEAKY02>), g, dim3(256), 0, st, x, dy, y, ldy, mean, invstd,
                           partial, Mg, C, sp.rows_per_block, act);
    else if (vec)
        hipLaunchKernelGGL((norm_partial<1, ADVMIX_ACT_NONE>), g, dim3(256), 0, st, x, dy, y, ldy, mean, invstd, partial,
                           Mg, C, sp.rows_per_block, act);
    else
        hipLaunchKernelGGL((norm_partial_scalar<1>), g, dim3(256), 0, st, x, dy, y, ldy, mean, invstd, partial, Mg, C,
                           sp.rows_per_block, act);
    hipLaunchKernelGGL(norm_finalize_bwd, dim3(cdiv((int64_t)groups * C, 4)), dim3(256), 0, st, partial, sp.nbg,
                       groups, Mg, C, coef, dgamma, dbeta);
    int64_t rows = (int64_t)groups * Mg;
    if (vec)
        hipLaunchKernelGGL((norm_bwd_apply<true>), dim3(stream_blocks(rows * (C / 4))), dim3(256), 0, st, dy, y, ldy,
                           x, mean, invstd, gamma, coef, dx, dres, Mg, rows, C, act);
    else
        hipLaunchKernelGGL((norm_bwd_apply<false>), dim3(stream_blocks(rows * C)), dim3(256), 0, st, dy, y, ldy, x,
                           mean, invstd, gamma, coef, dx, dres, Mg, rows, C, act);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// Train-mode BatchNorm forward whose statistics were accumulated by the producing conv's epilogue into
// slots[2][ns][C] (fp64, see conv_direct.hip): reduces them, writes mean / invstd, updates the running statistics and
// applies y = act(BN(c) + residual) in ONE launch.  ``act_mask`` (may be NULL): [rows][C / 4] bytes, bit e of byte
// (row, ch / 4) = "BN(c) + residual > 0 at channel 4 * (ch / 4) + e" - what a ReLU / LeakyReLU backward needs of y, at 1/16
// of its bytes (advmix_conv_tr_w_bnb reads it).  Returns ADVMIX_EINVAL (and launches nothing) for shapes it does
// not serve (C % 4 != 0, a channel tile that does not divide the workgroup): the caller falls back to
// advmix_norm_finalize + advmix_norm_apply.
extern "C" int advmix_norm_apply_slots(const float* c, const double* slots, int ns, int64_t rows, int C, float eps,
                                       const float* gamma, const float* beta, const float* residual, float* y,
                                       int act, float* mean, float* invstd, float* running_mean, float* running_var,
                                       int64_t* nbt, float momentum, unsigned char* act_mask, void* stream) {
    if (!c || !slots || !gamma || !beta || !y || !mean || !invstd || rows <= 0 || C <= 0) return ADVMIX_EINVAL;
    if ((running_mean != nullptr) != (running_var != nullptr)) return ADVMIX_EINVAL;
    if (!slots_ok(ns, C)) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(norm_apply_slots_kernel, slot_grid(rows, C), dim3(256), 0, (hipStream_t)stream, c, slots, ns, rows,
                       C, eps, gamma, beta, residual, y, act, mean, invstd, running_mean, running_var, nbt, momentum, act_mask);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// BatchNorm backward (groups = 1) whose two channel sums were left in slots[2][ns][C] by the input-gradient conv
// that produced ``g`` (= dy * act'(y), see advmix_conv_tr_w_bnb): dx, and dgamma += / dbeta += when given.
extern "C" int advmix_norm_bwd_apply_slots(const float* g, const float* c, const float* mean, const float* invstd,
                                           const float* gamma, const double* slots, int ns, int64_t rows, int C,
                                           float* dx, float* dgamma, float* dbeta, void* stream) {
    if (!g || !c || !mean || !invstd || !slots || !dx || rows <= 0 || C <= 0) return ADVMIX_EINVAL;
    if (!slots_ok(ns, C)) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(norm_bwd_apply_slots_kernel, slot_grid(rows, C), dim3(256), 0, (hipStream_t)stream, g, c, mean,
                       invstd, gamma, slots, ns, rows, C, dx, dgamma, dbeta);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// Deterministic mode: fold the per-tile partial sums a conv epilogue stored (partials[2][C][count], see
// advmix_conv_fwd_ex / advmix_conv_tr_w_bnb with *stats_nbg = -capacity) into slots_out[2][1][C] in a fixed order.
extern "C" int advmix_stats_fold(const double* partials, int count, int C, double* slots_out, void* stream) {
    if (!partials || !slots_out || count <= 0 || C <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(stats_fold_kernel, dim3(cdiv(2 * C, 4)), dim3(256), 0, (hipStream_t)stream, partials, count, 2 * C,
                       slots_out);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}
