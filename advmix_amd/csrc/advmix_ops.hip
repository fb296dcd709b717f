// AdvMix glue kernels (HBM-bound): view concat, softmax-mix fwd/bwd, joints loss fwd+bwd,
// heat-map argmax, flat Adam.
// Reference sites: lib/core/function.py:137-144; lib/core/loss.py:25-65;
// lib/core/inference.py:22-49; lib/utils/utils.py:89-92 (torch.optim.Adam defaults).
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {

static int stream_blocks(int64_t total) {
    int64_t b = (total + 255) / 256;
    const int cap = advmix_stream_cap();
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

// views are NCHW [N,3,H,W]; out NHWC [N,H,W,9]: G_input = torch.cat(inputs, 1) (function.py:137)
__global__ __launch_bounds__(256) void cat_views_kernel(const float* __restrict__ v0, const float* __restrict__ v1,
                                                        const float* __restrict__ v2, float* __restrict__ out, int N,
                                                        int HW) {
    const int64_t total = (int64_t)N * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t n = i / HW, p = i - n * HW;
        const int64_t base = n * 3 * HW + p;
        float* o = out + i * 9;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[c] = v0[base + (int64_t)c * HW];
            o[3 + c] = v1[base + (int64_t)c * HW];
            o[6 + c] = v2[base + (int64_t)c * HW];
        }
    }
}

__device__ __forceinline__ void softmax3(const float* l, float* w) {
    float m = fmaxf(l[0], fmaxf(l[1], l[2]));
    float e0 = expf(l[0] - m), e1 = expf(l[1] - m), e2 = expf(l[2] - m);
    float inv = 1.0f / (e0 + e1 + e2);
    w[0] = e0 * inv; w[1] = e1 * inv; w[2] = e2 * inv;
}

// tmp = sum_k view_k * softmax(logits)_k (function.py:138-144); one thread per pixel
__global__ __launch_bounds__(256) void mix_fwd_kernel(const float* __restrict__ v0, const float* __restrict__ v1,
                                                      const float* __restrict__ v2, const float* __restrict__ logits,
                                                      float* __restrict__ tmp, int N, int HW) {
    const int64_t total = (int64_t)N * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t n = i / HW, p = i - n * HW;
        const int64_t base = n * 3 * HW + p;
        float w[3];
        softmax3(logits + i * 3, w);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int64_t o = base + (int64_t)c * HW;
            float t = v0[o] * w[0];
            t += v1[o] * w[1];
            t += v2[o] * w[2];
            tmp[i * 3 + c] = t;
        }
    }
}

__global__ __launch_bounds__(256) void mix_bwd_kernel(const float* __restrict__ v0, const float* __restrict__ v1,
                                                      const float* __restrict__ v2, const float* __restrict__ logits,
                                                      const float* __restrict__ dtmp, float* __restrict__ dlogits,
                                                      int N, int HW) {
    const int64_t total = (int64_t)N * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t n = i / HW, p = i - n * HW;
        const int64_t base = n * 3 * HW + p;
        float w[3];
        softmax3(logits + i * 3, w);
        float dw[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int64_t o = base + (int64_t)c * HW;
            float d = dtmp[i * 3 + c];
            dw[0] += d * v0[o];
            dw[1] += d * v1[o];
            dw[2] += d * v2[o];
        }
        float dot = w[0] * dw[0] + w[1] * dw[1] + w[2] * dw[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) dlogits[i * 3 + k] = w[k] * (dw[k] - dot);
    }
}

// pred NHWC [B,HW,J]; element index e = (b*HW + p)*J + j.  One or two targets: loss_out += la L(pred, target) + lb L(pred,
// target_b), grad = ga dL(pred, target) + gb dL(pred, target_b) - the AdvMix student's heat-map + distillation loss
// (function.py:151-153) is one pass over pred; (la, ga, lb, gb) = (1, grad_scale, 0, 0) with target_b NULL is the single loss.
__global__ __launch_bounds__(256) void joints_loss_kernel(const float* __restrict__ pred,
                                                          const float* __restrict__ target, int target_nhwc,
                                                          const float* __restrict__ target_b, int b_nhwc,
                                                          const float* __restrict__ tw, float* __restrict__ loss_out,
                                                          float* __restrict__ grad, float la, float ga, float lb, float gb,
                                                          int B, int J, int HW, int mse, double* part) {
    __shared__ double red[4];
    const int64_t total = (int64_t)B * HW * J;
    const float norm = 0.5f / ((float)J * (float)B * (float)HW);
    double acc = 0.0, accb = 0.0;
    auto term = [&](float d, float& l, float& g) {
        if (mse) { l = d * d; g = 2.0f * d; }
        else {
            float ad = fabsf(d);
            if (ad < 1.0f) { l = 0.5f * d * d; g = d; }
            else { l = ad - 0.5f; g = d > 0.f ? 1.0f : -1.0f; }
        }
    };
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int j = (int)(e % J);
        int64_t bp = e / J;
        int64_t b = bp / HW, p = bp - b * HW;
        float w = tw ? tw[b * J + j] : 1.0f;
        float t = target_nhwc ? target[e] : target[(b * J + j) * HW + p];
        float pw = pred[e] * w;                           // loss.py:58-60: pred.mul(w), gt.mul(w)
        float l, g;
        term(pw - t * w, l, g);
        acc += (double)l;
        float gsum = ga * norm * w * g;
        if (target_b) {
            float tb = b_nhwc ? target_b[e] : target_b[(b * J + j) * HW + p];
            float l2, g2;
            term(pw - tb * w, l2, g2);
            accb += (double)l2;
            gsum += gb * norm * w * g2;
        }
        if (grad) grad[e] = gsum;
    }
    acc = wave_sum_d(acc * (double)la + accb * (double)lb);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = red[0] + red[1] + red[2] + red[3];
        if (part) part[blockIdx.x] = s * (double)norm;     // deterministic mode: summed in block order by loss_sum_kernel
        else atomicAdd(loss_out, (float)(s * (double)norm));
    }
}

__global__ void loss_sum_kernel(const double* __restrict__ part, int n, float* __restrict__ loss_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += part[i];
        *loss_out += (float)s;
    }
}

// one wave per (b, j); first-occurrence argmax (numpy.argmax semantics, inference.py:33)
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ hm, int nhwc, int32_t* __restrict__ idx,
                                                     float* __restrict__ mx, int B, int J, int HW) {
    const int lane = threadIdx.x & 63;
    const int64_t bj = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bj >= (int64_t)B * J) return;
    const int64_t b = bj / J, j = bj - b * J;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int p = lane; p < HW; p += 64) {
        float v = nhwc ? hm[(b * HW + p) * J + j] : hm[bj * HW + p];
        if (v > best || (v == best && p < bi) || (bi == 0x7fffffff)) { best = v; bi = p; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_down(best, o, 64);
        int oi = __shfl_down(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { idx[bj] = bi == 0x7fffffff ? 0 : bi; mx[bj] = best; }
}

__global__ void adam_tick_kernel(int64_t* step) { *step += 1; }

// torch.optim.SGD single-tensor math (dampening 0): g += wd * p; buf = momentum * buf + g; g = nesterov ? g + momentum *
// buf : buf; p -= lr * g.  hyper = {lr, momentum, weight_decay, nesterov (0 / 1)} in device memory (graph-replay safe).
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, int64_t n, const float* __restrict__ hyper) {
    const float lr = hyper[0], mom = hyper[1], wd = hyper[2];
    const bool nesterov = hyper[3] != 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        if (wd != 0.f) gi = gi + wd * p[i];
        if (mom != 0.f) {
            const float b = buf[i] * mom + gi;
            buf[i] = b;
            gi = nesterov ? gi + mom * b : b;
        }
        p[i] -= lr * gi;
    }
}

// torch.optim.Adam single-tensor math (no wd, no amsgrad)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   const float* __restrict__ hyper, const int64_t* __restrict__ step) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3];
    const double t = (double)*step;
    const float bc1 = (float)(1.0 - pow((double)b1, t));
    const float bc2s = (float)sqrt(1.0 - pow((double)b2, t));
    const float step_size = lr / bc1;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        float mi = m[i] + (gi - m[i]) * (1.0f - b1);          // lerp_
        float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        float denom = sqrtf(vi) / bc2s + eps;
        p[i] -= step_size * (mi / denom);
    }
}




// ---- ConvTranspose2d 4x4 / stride 2 / pad 1 with a NARROW output (Cout <= 4): the U-Net's last layer, 128 -> 3 at
// 256x192 (Unet_generator.py:51-57).  On the MFMA path it computes 32 output columns to keep 3 (507 us per step,
// 0.06 of peak); here it is a VALU dot product bound by reading the input once.  Output pixels of one PARITY class
// (oh % 2, ow % 2) use the same 2 x 2 taps, so a workgroup takes one class: thread (pixel lane 0..15, channel lane
// 0..15) keeps the weights of its 8 input channels x 4 taps x Cout in registers and walks over its pixels; the 16
// channel lanes of a pixel read 512 contiguous bytes per tap and meet in a 4-step shuffle.
template <int CO>
__global__ __launch_bounds__(256) void deconv4x4s2_narrow_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ y,
                                                                 int N, int Hi, int Wi, int Ci, int pix_per_block) {
    const int tid = threadIdx.x, cl = tid & 15, pl = tid >> 4;
    const int phase = blockIdx.y, ph = phase >> 1, pw = phase & 1;
    // taps of this class: kh in {kh0, kh0 + 2} reads input row a + dh, with a = oh / 2
    const int kh0 = (ph + 1) & 1, kw0 = (pw + 1) & 1;
    const int dh[2] = {ph == 0 ? 0 : 1, ph == 0 ? -1 : 0};   // kh0 -> dh[0], kh0 + 2 -> dh[1]
    const int dw[2] = {pw == 0 ? 0 : 1, pw == 0 ? -1 : 0};
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const int64_t P = (int64_t)N * Hi * Wi;                  // output pixels of one class
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = p0 + pix_per_block < P ? p0 + pix_per_block : P;
    float acc_b[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc_b[o] = bias ? bias[o] : 0.f;
    for (int c0 = 0; c0 < Ci; c0 += 128) {
        const int ci = c0 + cl * 8;
        const bool cok = ci < Ci;                            // (Ci % 8 == 0 is checked on the host)
        float wr[2][2][8][CO];                               // this thread's weights: [tap h][tap w][channel][co]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int o = 0; o < CO; ++o)
                        wr[a][b][c][o] = cok ? w[(((int64_t)(ci + c) * 4 + kh0 + 2 * a) * 4 + kw0 + 2 * b) * CO + o] : 0.f;
        for (int64_t p = p0 + pl; p < p1; p += 16) {
            const int n = (int)(p / ((int64_t)Hi * Wi));
            const int rem = (int)(p - (int64_t)n * Hi * Wi);
            const int a_ = rem / Wi, b_ = rem - a_ * Wi;
            float s[CO];
#pragma unroll
            for (int o = 0; o < CO; ++o) s[o] = 0.f;
            f32x4 v[2][2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int ih = a_ + dh[a], iw = b_ + dw[b];
                    const bool ok = cok && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi;
                    const float* src = x + (((int64_t)n * Hi + (ok ? ih : 0)) * Wi + (ok ? iw : 0)) * Ci + (cok ? ci : 0);
                    v[a][b][0] = ok ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
                    v[a][b][1] = ok ? *reinterpret_cast<const f32x4*>(src + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 8; ++c)
#pragma unroll
                        for (int o = 0; o < CO; ++o) s[o] = fmaf(v[a][b][c >> 2][c & 3], wr[a][b][c][o], s[o]);
#pragma unroll
            for (int o = 0; o < CO; ++o) {
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) s[o] += __shfl_xor(s[o], off, 64);   // the 16 channel lanes
            }
            if (cl == 0) {
                float* dst = y + (((int64_t)n * Ho + 2 * a_ + ph) * Wo + 2 * b_ + pw) * CO;
#pragma unroll
                for (int o = 0; o < CO; ++o) {
                    if (c0 == 0) dst[o] = s[o] + acc_b[o];
                    else dst[o] += s[o];                    // further 128-channel passes (same thread, same pixel)
                }
            }
        }
    }
}

// Transposed gather with a NARROW output (<= 4 channels): the input gradient of a network's first conv (3 image
// channels; needed when the images come from the generator) computed 32 MFMA columns for 3 (3x3 s2 64->3 @256x192,
// B = 32: 242 us; 7x7 s2: 743 us).  Here the Ck / 4 lanes of a pixel each take 4 upstream channels (a wave reads whole
// 256-byte pixel rows: one thread per pixel with 16-byte loads from 64 different rows per instruction is L1-bound
// and no faster than the MFMA kernel), the taps come from LDS as float4 (co padded to 4) laid out so that a lane
// group reads 16-byte neighbours (k-major rows put 16 lanes on two banks: 8-way conflicts, 190 us), four pixels per
// lane share them, the lanes meet in a shuffle tree.
template <int CO, int L>                                   // L = Ck / 4 lanes per pixel
__global__ __launch_bounds__(256) void conv_tr_narrow_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ y, int N, int Hs, int Ws, int Ck, int Hb,
                                                             int Wb, int R, int S, int stride, int pad) {
    extern __shared__ __attribute__((aligned(16))) float wl_[];
    f32x4* wl = reinterpret_cast<f32x4*>(wl_);
    const int tid = threadIdx.x;
    for (int i = tid; i < R * S * Ck; i += 256) {
        const int tap = i / Ck, ck = i - tap * Ck;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < CO; ++o) v[o] = w[((int64_t)ck * R * S + tap) * CO + o];
        wl[(tap * 4 + (ck & 3)) * (Ck >> 2) + (ck >> 2)] = v;  // [tap][k % 4][k / 4]: a lane group reads 16-byte neighbours
    }
    __syncthreads();
    const int cl = tid & (L - 1), c4 = cl * 4;
    // branch-free loads: an invalid pixel gets an out-of-range buffer offset and reads as zeros (a select around a
    // plain load compiles to a branch and a wait per load: four serialized round trips per tap)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, N * Hs * Ws * Ck * 4, 0x00020000);
    // Positions in stride-phase coordinates: output pixel (a * stride + rh, b * stride + rw).  A lane group takes PB = 4
    // consecutive b of one (n, a) - the taps it reads from LDS serve four pixels, the four upstream rows per tap are
    // contiguous - for every phase in turn (the phases share upstream rows; the taps of one pass are uniform over the
    // workgroup).  A workgroup stages the taps once and walks many positions.
    constexpr int PB = 4;
    const int Ha = (Hb + stride - 1) / stride, Wa = (Wb + stride - 1) / stride, Wq = (Wa + PB - 1) / PB;
    const int Pq = N * Ha * Wq, HWq = Ha * Wq;               // (32-bit: the host checks the tensor sizes)
    for (int p = blockIdx.x * (256 / L) + tid / L; p < Pq; p += gridDim.x * (256 / L)) {
        const int n = p / HWq;
        const int rem = p - n * HWq;
        const int a_ = rem / Wq, b0 = (rem - a_ * Wq) * PB;
        for (int rh = 0; rh < stride; ++rh) {
            const int hb = a_ * stride + rh;
            if (hb >= Hb) continue;
            for (int rw = 0; rw < stride; ++rw) {
                float s[PB][CO];
#pragma unroll
                for (int j = 0; j < PB; ++j)
#pragma unroll
                    for (int o = 0; o < CO; ++o) s[j][o] = 0.f;
                for (int kh = (rh + pad) % stride; kh < R; kh += stride) {
                    const int hs = (hb + pad - kh) / stride;         // exact: kh = (hb + pad) mod stride
                    if (hb + pad < kh || hs >= Hs) continue;
                    for (int kw = (rw + pad) % stride; kw < S; kw += stride) {
                        const int ws0 = (b0 * stride + rw + pad - kw) / stride;   // exact; may be -1 at the left border
                        f32x4 v[PB];
#pragma unroll
                        for (int j = 0; j < PB; ++j) {
                            const int ws = ws0 + j;
                            const bool ok = ws >= 0 && ws < Ws && b0 * stride + rw + pad >= kw - j * stride;
                            v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                rx, ok ? (unsigned)((((n * Hs + hs) * Ws + ws) * Ck + c4) * 4) : 0x80000000u, 0, 0));
                        }
                        const f32x4* wt = wl + ((kh * S + kw) * 4) * L + cl;       // [tap][e][lane]: conflict-free
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f32x4 wv = wt[e * L];
#pragma unroll
                            for (int j = 0; j < PB; ++j)
#pragma unroll
                                for (int o = 0; o < CO; ++o) s[j][o] = fmaf(v[j][e], wv[o], s[j][o]);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < PB; ++j)
#pragma unroll
                    for (int o = 0; o < CO; ++o)
#pragma unroll
                        for (int off = L >> 1; off > 0; off >>= 1) s[j][o] += __shfl_xor(s[j][o], off, 64);
                if (cl == 0) {
#pragma unroll
                    for (int j = 0; j < PB; ++j) {
                        const int wb = (b0 + j) * stride + rw;
                        if (wb < Wb) {
#pragma unroll
                            for (int o = 0; o < CO; ++o) y[(((int64_t)n * Hb + hb) * Wb + wb) * CO + o] = s[j][o];
                        }
                    }
                }
            }
        }
    }
}

}  // namespace

AdvmixOpts& advmix_opts() {
    static AdvmixOpts o = [] {
        AdvmixOpts d{1, 1, 1, 0, 0, 1, 0};
        const char* e;
        if ((e = getenv("ADVMIX_CONV")) && e[0] == 'i') d.direct = 0;
        if ((e = getenv("ADVMIX_WGRAD"))) d.wgrad_direct = e[0] != '0';
        if ((e = getenv("ADVMIX_WGRAD_LDS"))) d.wgrad_lds = atoi(e);
        if ((e = getenv("ADVMIX_KSPLIT_WG"))) d.ksplit_wg = e[0] != '0';
        if ((e = getenv("ADVMIX_STAT_SLOTS"))) d.stat_slots = atoi(e);
        return d;
    }();
    return o;
}

void advmix_trace_launch(const char* kernel, dim3 grid, const char* kind, int N, int Hi, int Wi, int Ci, int Ho, int Wo,
                         int Co, int R, int S, int stride, double flops) {
    if (!advmix_opts().trace_shapes) return;
    static FILE* f = [] {
        const char* path = getenv("ADVMIX_TRACE_SHAPES");
        FILE* h = path ? fopen(path, "w") : nullptr;
        if (h) fprintf(h, "kernel,grid_x,grid_y,grid_z,kind,N,Hi,Wi,Ci,Ho,Wo,Co,R,S,stride,flops\n");
        return h;
    }();
    if (!f) return;
    fprintf(f, "\"%s\",%u,%u,%u,%s,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.0f\n", kernel, grid.x, grid.y, grid.z, kind, N, Hi, Wi,
            Ci, Ho, Wo, Co, R, S, stride, flops);
    fflush(f);
}

extern "C" int advmix_version(void) { return 1; }

extern "C" int advmix_set_option(const char* name, int value) {
    if (!name) return ADVMIX_EINVAL;
    AdvmixOpts& o = advmix_opts();
    if (!strcmp(name, "direct")) o.direct = value;
    else if (!strcmp(name, "wgrad_direct")) o.wgrad_direct = value;
    else if (!strcmp(name, "wgrad_lds")) o.wgrad_lds = value;
    else if (!strcmp(name, "ksplit_wg")) o.ksplit_wg = value;
    else if (!strcmp(name, "stat_slots")) o.stat_slots = value;
    else if (!strcmp(name, "trace_shapes")) o.trace_shapes = value;
    else if (!strcmp(name, "deterministic")) o.deterministic = value;
    else return ADVMIX_EINVAL;
    return ADVMIX_OK;
}

extern "C" int advmix_cat_views(const float* v0, const float* v1, const float* v2, float* out, int N, int H, int W,
                                void* stream) {
    if (!v0 || !v1 || !v2 || !out || N <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(cat_views_kernel, dim3(stream_blocks((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream,
                       v0, v1, v2, out, N, H * W);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_mix_fwd(const float* v0, const float* v1, const float* v2, const float* logits, float* tmp,
                              int N, int H, int W, void* stream) {
    if (!v0 || !v1 || !v2 || !logits || !tmp || N <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(mix_fwd_kernel, dim3(stream_blocks((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, v0,
                       v1, v2, logits, tmp, N, H * W);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_mix_bwd(const float* v0, const float* v1, const float* v2, const float* logits,
                              const float* dtmp, float* dlogits, int N, int H, int W, void* stream) {
    if (!v0 || !v1 || !v2 || !logits || !dtmp || !dlogits || N <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(mix_bwd_kernel, dim3(stream_blocks((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, v0,
                       v1, v2, logits, dtmp, dlogits, N, H * W);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

static int joints_loss_impl(const float* pred, const float* target, int target_nhwc, const float* target_b, int b_nhwc,
                            const float* tw, float* loss_out, float* grad, float la, float ga, float lb, float gb, int B, int J,
                            int HW, int mse, double* part, void* stream) {
    if (!pred || !target || !loss_out || B <= 0 || J <= 0 || HW <= 0) return ADVMIX_EINVAL;
    int64_t total = (int64_t)B * J * HW;
    int blocks = stream_blocks(total);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(joints_loss_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, target_nhwc,
                       target_b, b_nhwc, tw, loss_out, grad, la, ga, lb, gb, B, J, HW, mse, part);
    if (part) hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, blocks, loss_out);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_joints_loss(const float* pred, const float* target, int target_nhwc, const float* tw,
                                  float* loss_out, float* grad, float grad_scale, int B, int J, int HW, int mse,
                                  void* stream) {
    return joints_loss_impl(pred, target, target_nhwc, nullptr, 0, tw, loss_out, grad, 1.0f, grad_scale, 0.f, 0.f, B, J, HW, mse,
                            nullptr, stream);
}

// deterministic variant: the <= 512 block sums go to ws (>= 4 KiB) and are added in block order
extern "C" int advmix_joints_loss_det(const float* pred, const float* target, int target_nhwc, const float* tw,
                                      float* loss_out, float* grad, float grad_scale, int B, int J, int HW, int mse,
                                      void* ws, void* stream) {
    if (!ws) return ADVMIX_EINVAL;
    return joints_loss_impl(pred, target, target_nhwc, nullptr, 0, tw, loss_out, grad, 1.0f, grad_scale, 0.f, 0.f, B, J, HW, mse,
                            (double*)ws, stream);
}

// loss_out += scale_a L(pred, target_a) + scale_b L(pred, target_b), grad = the same blend of the two gradients, in ONE pass
// over pred (target_b may be NULL: a scaled single loss, e.g. the generator's -adv_loss_weight L).  Replaces the torch
// arithmetic around two criterion calls at lib/core/function.py:151-153 (and :161).  ws != NULL: deterministic block order.
extern "C" int advmix_joints_loss_blend(const float* pred, const float* target_a, int a_nhwc, const float* target_b, int b_nhwc,
                                        const float* tw, float* loss_out, float* grad, float scale_a, float scale_b, int B,
                                        int J, int HW, int mse, void* ws, void* stream) {
    return joints_loss_impl(pred, target_a, a_nhwc, target_b, b_nhwc, tw, loss_out, grad, scale_a, scale_a, scale_b, scale_b, B, J,
                            HW, mse, (double*)ws, stream);
}

extern "C" int advmix_heatmap_argmax(const float* hm, int nhwc, int32_t* idx_out, float* max_out, int B, int J,
                                     int HW, void* stream) {
    if (!hm || !idx_out || !max_out || B <= 0 || J <= 0 || HW <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(argmax_kernel, dim3(cdiv((int64_t)B * J, 4)), dim3(256), 0, (hipStream_t)stream, hm, nhwc,
                       idx_out, max_out, B, J, HW);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_adam(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                           int64_t* step, void* stream) {
    if (!p || !g || !m || !v || !hyper || !step || n <= 0) return ADVMIX_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step);
    hipLaunchKernelGGL(adam_kernel, dim3(stream_blocks(n)), dim3(256), 0, st, p, g, m, v, n, hyper, step);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// y[N,Hb,Wb,Cn] = conv_transpose(x[N,Hs,Ws,Ck], w[Ck][R][S][Cn]) for Cn <= 4 (a Conv2d's input gradient with the
// weights in their own layout, as advmix_conv_tr_w).  ADVMIX_EINVAL (nothing launched) for other shapes.
extern "C" int advmix_conv_tr_narrow(const float* x, const float* w, float* y, int N, int Hs, int Ws, int Ck, int Hb,
                                     int Wb, int Cn, int R, int S, int stride, int pad, void* stream) {
    if (!x || !w || !y || N <= 0 || Ck <= 0 || stride < 1) return ADVMIX_EINVAL;
    const int L = Ck / 4;                                    // lanes per pixel
    if (Cn < 1 || Cn > 4 || Ck % 4 != 0 || (L != 16 && L != 32) || stride > 8) return ADVMIX_EINVAL;
    if (Hs != (Hb + 2 * pad - R) / stride + 1 || Ws != (Wb + 2 * pad - S) / stride + 1) return ADVMIX_EINVAL;
    const size_t lds = (size_t)R * S * Ck * 16;
    if (lds > 64 * 1024) return ADVMIX_EINVAL;
    // byte sizes (the buffer resource's num_records and the per-load offsets are 32-bit BYTE quantities)
    if ((int64_t)N * Hb * Wb * 4 * 4 >= 0x7fffffffLL || (int64_t)N * Hs * Ws * Ck * 4 >= 0x7fffffffLL) return ADVMIX_EINVAL;
    const int64_t P = (int64_t)N * cdiv(Hb, stride) * cdiv(cdiv(Wb, stride), 4);   // positions of 4 pixels in stride-phase coordinates
    int64_t gx = (P + 256 / L - 1) / (256 / L);
    if (gx > 3072) gx = 3072;
    dim3 g((unsigned)gx);
    hipStream_t st = (hipStream_t)stream;
#define NARROW(CO_)                                                                                               \
    do {                                                                                                          \
        if (L == 16) hipLaunchKernelGGL((conv_tr_narrow_kernel<CO_, 16>), g, dim3(256), lds, st, x, w, y, N, Hs, Ws, Ck, Hb, Wb, R, S, stride, pad); \
        else hipLaunchKernelGGL((conv_tr_narrow_kernel<CO_, 32>), g, dim3(256), lds, st, x, w, y, N, Hs, Ws, Ck, Hb, Wb, R, S, stride, pad); \
    } while (0)
    switch (Cn) {
        case 1: NARROW(1); break;
        case 2: NARROW(2); break;
        case 3: NARROW(3); break;
        default: NARROW(4); break;
    }
#undef NARROW
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// ConvTranspose2d(Cin, Cout <= 4, kernel 4, stride 2, padding 1) forward, weights in their own layout
// w[Cin][4][4][Cout]; x [N,Hi,Wi,Cin] -> y [N,2Hi,2Wi,Cout].  ADVMIX_EINVAL (nothing launched) for other shapes.
extern "C" int advmix_deconv4x4s2_narrow(const float* x, const float* w, const float* bias, float* y, int N, int Hi,
                                         int Wi, int Ci, int Co, void* stream) {
    if (!x || !w || !y || N <= 0 || Hi <= 0 || Wi <= 0) return ADVMIX_EINVAL;
    if (Co < 1 || Co > 4 || Ci % 8 != 0) return ADVMIX_EINVAL;
    const int64_t P = (int64_t)N * Hi * Wi;
    int ppb = 256;                                           // 16 pixels per pass x 16 passes per workgroup
    while ((P + ppb - 1) / ppb > 4096) ppb *= 2;
    dim3 g((unsigned)((P + ppb - 1) / ppb), 4);
    hipStream_t st = (hipStream_t)stream;
    switch (Co) {
        case 1: hipLaunchKernelGGL((deconv4x4s2_narrow_kernel<1>), g, dim3(256), 0, st, x, w, bias, y, N, Hi, Wi, Ci, ppb); break;
        case 2: hipLaunchKernelGGL((deconv4x4s2_narrow_kernel<2>), g, dim3(256), 0, st, x, w, bias, y, N, Hi, Wi, Ci, ppb); break;
        case 3: hipLaunchKernelGGL((deconv4x4s2_narrow_kernel<3>), g, dim3(256), 0, st, x, w, bias, y, N, Hi, Wi, Ci, ppb); break;
        default: hipLaunchKernelGGL((deconv4x4s2_narrow_kernel<4>), g, dim3(256), 0, st, x, w, bias, y, N, Hi, Wi, Ci, ppb); break;
    }
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// The same layer as a GEMM and a gather (round 4).  deconv4x4s2_narrow_kernel reads every input element 16 times - four
// parity classes x four taps - from L2 / L1: 3.2 GB for the U-Net's 128 -> 3 tail, 395 us, 0.06 of the matrix peak and L2-bound.
// Here every input pixel's 16 x Cout products p[pixel][kh][kw][co] = sum_ci x[pixel][ci] w[ci][kh][kw][co] are ONE 1 x 1
// transposed-weight convolution on the matrix pipe (w[Cin][4][4][Cout] IS its k-major weight, Cn = 16 Cout: no padding, no
// re-layout; the input is read once), and an output pixel is the sum of the four products that land on it:
// (oh, ow) <- kh = (oh + 1) % 2 (+ 2), a = (oh + 1 - kh) / 2, likewise kw, b.
template <int CO>
__global__ __launch_bounds__(256) void deconv4x4s2_gather_kernel(const float* __restrict__ p, const float* __restrict__ bias,
                                                                 float* __restrict__ y, int N, int Hi, int Wi) {
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const int64_t total = (int64_t)N * Ho * Wo;
    float bv[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) bv[o] = bias ? bias[o] : 0.f;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
        const int ow = (int)(i % Wo);
        const int64_t t = i / Wo;
        const int oh = (int)(t % Ho), n = (int)(t / Ho);
        const int kh0 = (oh + 1) & 1, kw0 = (ow + 1) & 1;
        float s[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) s[o] = bv[o];
#pragma unroll
        for (int dh = 0; dh < 4; dh += 2) {
            const int kh = kh0 + dh, a = (oh + 1 - kh) / 2;               // (oh + 1 - kh is even; -2 -> row -1: outside)
            if ((unsigned)a >= (unsigned)Hi) continue;
#pragma unroll
            for (int dw = 0; dw < 4; dw += 2) {
                const int kw = kw0 + dw, b = (ow + 1 - kw) / 2;
                if ((unsigned)b >= (unsigned)Wi) continue;
                const float* src = p + (((int64_t)n * Hi + a) * Wi + b) * (16 * CO) + (kh * 4 + kw) * CO;
#pragma unroll
                for (int o = 0; o < CO; ++o) s[o] += src[o];
            }
        }
#pragma unroll
        for (int o = 0; o < CO; ++o) y[i * CO + o] = s[o];
    }
}

extern "C" int64_t advmix_deconv4x4s2_narrow_ws_bytes(int N, int Hi, int Wi, int Co) {
    return (int64_t)N * Hi * Wi * 16 * Co * 4;
}

// ConvTranspose2d(Cin, Cout <= 4, 4, 2, 1) forward through ``ws`` (advmix_deconv4x4s2_narrow_ws_bytes; the products of every
// input pixel).  ADVMIX_EINVAL (nothing launched) where the 1 x 1 kernel does not serve the shape (Cin % 16) or ws is too small:
// call advmix_deconv4x4s2_narrow.
extern "C" int advmix_deconv4x4s2_narrow_gemm(const float* x, const float* w, const float* bias, float* y, float* ws,
                                              int64_t ws_bytes, int N, int Hi, int Wi, int Ci, int Co, void* stream) {
    if (!x || !w || !y || !ws || N <= 0 || Hi <= 0 || Wi <= 0) return ADVMIX_EINVAL;
    if (Co < 1 || Co > 4 || Ci % 16 != 0) return ADVMIX_EINVAL;
    if (ws_bytes < advmix_deconv4x4s2_narrow_ws_bytes(N, Hi, Wi, Co)) return ADVMIX_EINVAL;
    const int rc = advmix_conv_tr_w_add(x, w, nullptr, ws, N, Hi, Wi, Ci, Hi, Wi, 16 * Co, 1, 1, 1, 0, stream);
    if (rc != ADVMIX_OK) return rc;
    const int64_t total = (int64_t)N * 4 * Hi * Wi;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    dim3 g((unsigned)blocks);
    hipStream_t st = (hipStream_t)stream;
    switch (Co) {
        case 1: hipLaunchKernelGGL((deconv4x4s2_gather_kernel<1>), g, dim3(256), 0, st, ws, bias, y, N, Hi, Wi); break;
        case 2: hipLaunchKernelGGL((deconv4x4s2_gather_kernel<2>), g, dim3(256), 0, st, ws, bias, y, N, Hi, Wi); break;
        case 3: hipLaunchKernelGGL((deconv4x4s2_gather_kernel<3>), g, dim3(256), 0, st, ws, bias, y, N, Hi, Wi); break;
        default: hipLaunchKernelGGL((deconv4x4s2_gather_kernel<4>), g, dim3(256), 0, st, ws, bias, y, N, Hi, Wi); break;
    }
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// One SGD step over a flat parameter buffer (lib/utils/utils.py:80-88: optim.SGD(lr, momentum, weight_decay, nesterov)).
extern "C" int advmix_sgd(float* p, const float* g, float* buf, int64_t n, const float* hyper, void* stream) {
    if (!p || !g || !buf || !hyper || n <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(sgd_kernel, dim3(stream_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, hyper);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}
