// Device side of the three-view input pipeline (HBM-bound, a few microseconds per batch).
// Reference sites: tools/train.py:116-126 (ToTensor + Normalize), lib/dataset/advaug.py:111-170 (grid_aug,
// called with use_h = use_w = True, rotate = 1, offset = False, mode = 1: advaug.py:189-202),
// lib/dataset/JointsDataset.py:412-491 (generate_target, gaussian branch).
// The loader keeps what needs the CPU (image decode, cv2 warp, PIL AutoAugment, the numpy RNG draws) and
// hands over ONE uint8 crop (+ the AutoAugment uint8 crop) per sample instead of three float32 tensors:
// 12x fewer PCIe bytes, and the normalisation / GridMask / target rendering leave the worker processes.
#include "common.h"

namespace {

static int stream_blocks(int64_t total) {
    int64_t b = (total + 255) / 256;
    const int cap = advmix_stream_cap();
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

// 1 where grid_aug's mask (mode = 1, i.e. 1 - mask) keeps the pixel: on a row stripe or on a column stripe.
// Stripes live on the 1.5x canvas (hh x ww), the image is its centre crop (advaug.py:116-146).
__device__ __forceinline__ bool grid_keep(int y, int x, int H, int W, int d, int l, int st_h, int st_w) {
    const int hh = (int)(1.5 * H), ww = (int)(1.5 * W);
    const int Y = y + (hh - H) / 2, X = x + (ww - W) / 2;
    bool row = false, col = false;
    int q = Y - st_h;
    if (q >= 0) { int i = q / d; row = i < hh / d && q - i * d < l; }
    q = X - st_w;
    if (q >= 0) { int i = q / d; col = i < ww / d && q - i * d < l; }
    return row || col;
}

// base / aug: uint8 [B][H][W][3]; views: float32 NCHW [B][3][H][W].
// v = (u8 / 255 - mean) / std in float32, each step rounded like torchvision's ToTensor + Normalize.
template <bool VEC>
__global__ __launch_bounds__(256) void make_views_kernel(const uint8_t* __restrict__ base,
                                                         const uint8_t* __restrict__ aug,
                                                         const int32_t* __restrict__ grid, float m0, float m1, float m2,
                                                         float s0, float s1, float s2, float* __restrict__ v0,
                                                         float* __restrict__ v1, float* __restrict__ v2, int B, int H,
                                                         int W) {
    constexpr int PX = VEC ? 4 : 1;                        // pixels per thread (W % 4 == 0: 12 B in, 9 x 16 B out)
    const int64_t HW = (int64_t)H * W, total = (int64_t)B * HW / PX;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    for (int64_t it = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = it * PX;
        const int64_t b = i / HW, p = i - b * HW;
        const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
        int d = 0, l = 0, sh = 0, sw = 0;
        if (grid) { d = grid[b * 4]; l = grid[b * 4 + 1]; sh = grid[b * 4 + 2]; sw = grid[b * 4 + 3]; }
        uint8_t ub[3 * PX], ua[3 * PX];
        if (VEC) {
            const uint32_t* pb = reinterpret_cast<const uint32_t*>(base + i * 3);
            uint32_t w0 = pb[0], w1 = pb[1], w2 = pb[2];
            __builtin_memcpy(ub, &w0, 4); __builtin_memcpy(ub + 4, &w1, 4); __builtin_memcpy(ub + 8, &w2, 4);
            if (aug) {
                const uint32_t* pa = reinterpret_cast<const uint32_t*>(aug + i * 3);
                w0 = pa[0]; w1 = pa[1]; w2 = pa[2];
                __builtin_memcpy(ua, &w0, 4); __builtin_memcpy(ua + 4, &w1, 4); __builtin_memcpy(ua + 8, &w2, 4);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) { ub[c] = base[i * 3 + c]; if (aug) ua[c] = aug[i * 3 + c]; }
        }
        float keep[PX];
#pragma unroll
        for (int k = 0; k < PX; ++k) keep[k] = (d > 0 && !grid_keep(y, x + k, H, W, d, l, sh, sw)) ? 0.0f : 1.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t o = (b * 3 + c) * HW + p;
            float n[PX], na[PX], ng[PX];
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                n[k] = __fdiv_rn(__fdiv_rn((float)ub[k * 3 + c], 255.0f) - mean[c], sd[c]);
                na[k] = aug ? __fdiv_rn(__fdiv_rn((float)ua[k * 3 + c], 255.0f) - mean[c], sd[c]) : n[k];
                ng[k] = n[k] * keep[k];                      // a product, like the reference (keeps -0.0)
            }
            if (VEC) {
                *reinterpret_cast<f32x4*>(v0 + o) = f32x4{n[0], n[PX > 1 ? 1 : 0], n[PX > 2 ? 2 : 0], n[PX > 3 ? 3 : 0]};
                if (v1) *reinterpret_cast<f32x4*>(v1 + o) = f32x4{na[0], na[PX > 1 ? 1 : 0], na[PX > 2 ? 2 : 0], na[PX > 3 ? 3 : 0]};
                if (v2) *reinterpret_cast<f32x4*>(v2 + o) = f32x4{ng[0], ng[PX > 1 ? 1 : 0], ng[PX > 2 ? 2 : 0], ng[PX > 3 ? 3 : 0]};
            } else {
                v0[o] = n[0];
                if (v1) v1[o] = na[0];
                if (v2) v2[o] = ng[0];
            }
        }
    }
}

// One workgroup per (b, j): heat-map [Hh][Wh] with the 13x13 (2*tmp+1) patch g centred on the joint, or zeros.
__global__ __launch_bounds__(256) void render_targets_kernel(const double* __restrict__ joints,
                                                             const double* __restrict__ vis,
                                                             const int32_t* __restrict__ grid,
                                                             const float* __restrict__ g, int tmp,
                                                             const float* __restrict__ jw, float* __restrict__ target,
                                                             float* __restrict__ tw, double* __restrict__ vis_out,
                                                             int J, int H, int W, int Hh, int Wh) {
    const int64_t bj = blockIdx.x;
    const int64_t b = bj / J;
    const int j = (int)(bj - b * J);
    const double jx = joints[bj * 3], jy = joints[bj * 3 + 1];
    double v0 = vis[bj * 3], v1 = vis[bj * 3 + 1];
    if (grid && grid[b * 4] > 0) {                          // advaug.py:159-168: joints under the mask turn invisible
        int tx = (int)jx, ty = (int)jy;
        tx = min(tx, W - 1); tx = max(tx, 0);
        ty = min(ty, H - 1); ty = max(ty, 0);
        if (!grid_keep(ty, tx, H, W, grid[b * 4], grid[b * 4 + 1], grid[b * 4 + 2], grid[b * 4 + 3])) { v0 = 0.0; v1 = 0.0; }
    }
    if (vis_out && threadIdx.x == 0) { vis_out[bj * 3] = v0; vis_out[bj * 3 + 1] = v1; vis_out[bj * 3 + 2] = vis[bj * 3 + 2]; }
    float w = (float)v0;                                     // target_weight[:, 0] = joints_vis[:, 0]
    const double fsx = (double)W / (double)Wh, fsy = (double)H / (double)Hh;      // feat_stride
    const int mu_x = (int)(jx / fsx + 0.5), mu_y = (int)(jy / fsy + 0.5);
    const int ulx = mu_x - tmp, uly = mu_y - tmp, brx = mu_x + tmp + 1, bry = mu_y + tmp + 1;
    if (ulx >= Wh || uly >= Hh || brx < 0 || bry < 0) w = 0.f;
    const bool draw = w > 0.5f;
    const int size = 2 * tmp + 1;
    float* t = target + bj * (int64_t)Hh * Wh;
    for (int p = threadIdx.x; p < Hh * Wh; p += blockDim.x) {
        const int y = p / Wh, x = p - y * Wh;
        const int gy = y - uly, gx = x - ulx;
        t[p] = (draw && gy >= 0 && gy < size && gx >= 0 && gx < size) ? g[gy * size + gx] : 0.f;
    }
    if (threadIdx.x == 0) tw[bj] = jw ? w * jw[j] : w;
}

// ---- the AutoAugment view on the device (lib/dataset/advaug.py:10-108, applied per sample at JointsDataset.py:124) ------
// ImageNetPolicy's table (advaug.py:22-35) only reaches five Pillow operations: ImageOps.equalize / posterize / solarize /
// invert (256-entry look-up tables) and ImageEnhance.Sharpness (3x3 SMOOTH filter + Image.blend).  The worker keeps the
// draws (Python's ``random``: dataset.advaug.autoaug_params) and ships (code, parameter) pairs; one workgroup per image
// applies up to two operations back to back, bit for bit what Pillow computes:
//   equalize   per band: histogram h, step = (sum of non-zero bins - last non-zero bin) / 255, lut[i] = min(255, n / step)
//              with n = step / 2 + h[0] + ... + h[i-1]; identity when <= 1 non-zero bin or step == 0 (ImageOps.py);
//   posterize  i & mask;   solarize  i < threshold ? i : 255 - i (threshold is a float);   invert  255 - i;
//   sharpness  d = SMOOTH(im): float32, 0.5 + row below + row itself + row above, each ((a*k0 + b*k1) + c*k2), k = {1,1,1,
//              1,5,1,1,1,1} / 13.0f, truncated, borders copied (Filter.c); out = d + alpha * (im - d) in float32, truncated,
//              clipped to 0..255 when alpha is outside [0, 1] (Blend.c).  No fused multiply-adds anywhere (x86-64 Pillow
//              has none): aa_mul.
enum { AA_NONE = 0, AA_EQUALIZE = 1, AA_POSTERIZE = 2, AA_SOLARIZE = 3, AA_INVERT = 4, AA_SHARPNESS = 5 };

// a product the compiler cannot fuse into a following add (hipcc contracts a * b + c into v_fmac_f32 by default, and the
// __fmul_rn / __fadd_rn intrinsics are plain operators to it): the value passes through an empty asm
__device__ __forceinline__ float aa_mul(float a, float b) {
    float m = a * b;
    asm volatile("" : "+v"(m));
    return m;
}

__device__ __forceinline__ uint8_t aa_clip8(float v) { return v <= 0.f ? 0 : (v >= 255.f ? 255 : (uint8_t)(int)v); }

__global__ __launch_bounds__(1024) void autoaug_kernel(const uint8_t* __restrict__ base, const int32_t* __restrict__ ops,
                                                       uint8_t* __restrict__ tmp, uint8_t* __restrict__ out, int H, int W) {
    __shared__ unsigned hist[3][256];
    __shared__ uint8_t lut[3][256];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int64_t n = (int64_t)H * W * 3;
    const uint8_t* src = base + b * n;
    int code[2] = {ops[b * 4], ops[b * 4 + 2]}, par[2] = {ops[b * 4 + 1], ops[b * 4 + 3]};
    const int nops = (code[0] != AA_NONE) + (code[1] != AA_NONE);
    const bool vec4 = (n & 3) == 0 && ((((uintptr_t)base | (uintptr_t)tmp | (uintptr_t)out)) & 3) == 0;   // (b * n keeps the alignment)
    if (nops == 0) {                                        // neither operation fired: the view is the crop itself
        if (vec4) {
            for (int64_t i = (int64_t)tid * 4; i < n; i += (int64_t)nt * 4)
                *reinterpret_cast<uint32_t*>(out + b * n + i) = *reinterpret_cast<const uint32_t*>(src + i);
        } else {
            for (int64_t i = tid; i < n; i += nt) out[b * n + i] = src[i];
        }
        return;
    }
    int done = 0;
    for (int k = 0; k < 2; ++k) {
        if (code[k] == AA_NONE) continue;
        uint8_t* dst = (done + 1 == nops ? out : tmp) + b * n;
        if (code[k] == AA_SHARPNESS) {
            const float alpha = __builtin_bit_cast(float, par[k]);
            const float k1 = __fdiv_rn(1.0f, 13.0f), k5 = __fdiv_rn(5.0f, 13.0f);
            const int W3 = W * 3;
            for (int64_t i = tid; i < n; i += nt) {
                const int y = (int)(i / W3), r = (int)(i - (int64_t)y * W3), x = r / 3;
                const int im = src[i];
                int d = im;
                if (y > 0 && y < H - 1 && x > 0 && x < W - 1) {
                    const uint8_t* p1 = src + i + W3;       // row below
                    const uint8_t* p0 = src + i;
                    const uint8_t* pm = src + i - W3;       // row above
                    float ss = 0.5f;
                    ss = __fadd_rn(ss, __fadd_rn(__fadd_rn(aa_mul((float)p1[-3], k1), aa_mul((float)p1[0], k1)), aa_mul((float)p1[3], k1)));
                    ss = __fadd_rn(ss, __fadd_rn(__fadd_rn(aa_mul((float)p0[-3], k1), aa_mul((float)p0[0], k5)), aa_mul((float)p0[3], k1)));
                    ss = __fadd_rn(ss, __fadd_rn(__fadd_rn(aa_mul((float)pm[-3], k1), aa_mul((float)pm[0], k1)), aa_mul((float)pm[3], k1)));
                    d = aa_clip8(ss);
                }
                uint8_t o;
                if (alpha == 0.0f) o = (uint8_t)d;
                else if (alpha == 1.0f) o = (uint8_t)im;
                else {
                    const float t = __fadd_rn((float)d, aa_mul(alpha, (float)(im - d)));
                    o = (alpha >= 0.f && alpha <= 1.f) ? (uint8_t)(int)t : aa_clip8(t);
                }
                dst[i] = o;
            }
        } else {
            if (code[k] == AA_EQUALIZE) {
                for (int i = tid; i < 768; i += nt) hist[i >> 8][i & 255] = 0;
                __syncthreads();
                if (vec4) {                                 // four bytes per load
                    for (int64_t i = (int64_t)tid * 4; i < n; i += (int64_t)nt * 4) {
                        const uint32_t w4 = *reinterpret_cast<const uint32_t*>(src + i);
                        const int c0 = (int)(i % 3);
#pragma unroll
                        for (int j = 0; j < 4; ++j) atomicAdd(&hist[(c0 + j) % 3][(w4 >> (8 * j)) & 255u], 1u);
                    }
                } else {
                    for (int64_t i = tid; i < n; i += nt) atomicAdd(&hist[i % 3][src[i]], 1u);
                }
                __syncthreads();
                if (tid < 3) {                              // 256 serial steps per band: nothing next to the passes over the image
                    unsigned total = 0, last = 0;
                    int nz = 0;
                    for (int i = 0; i < 256; ++i) { const unsigned h = hist[tid][i]; if (h) { ++nz; total += h; last = h; } }
                    const unsigned step = nz <= 1 ? 0 : (total - last) / 255;
                    unsigned acc = step / 2;
                    for (int i = 0; i < 256; ++i) {
                        unsigned v = step ? acc / step : (unsigned)i;
                        lut[tid][i] = (uint8_t)(v > 255 ? 255 : v);      // Image.point clips the table
                        acc += hist[tid][i];
                    }
                }
            } else if (tid < 256) {
                int v = tid;
                if (code[k] == AA_POSTERIZE) v = tid & par[k];
                else if (code[k] == AA_SOLARIZE) v = ((float)tid < __builtin_bit_cast(float, par[k])) ? tid : 255 - tid;
                else v = 255 - tid;                         // AA_INVERT
                lut[0][tid] = lut[1][tid] = lut[2][tid] = (uint8_t)v;
            }
            __syncthreads();
            if (vec4) {
                for (int64_t i = (int64_t)tid * 4; i < n; i += (int64_t)nt * 4) {
                    const uint32_t w4 = *reinterpret_cast<const uint32_t*>(src + i);
                    const int c0 = (int)(i % 3);
                    uint32_t o = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o |= (uint32_t)lut[(c0 + j) % 3][(w4 >> (8 * j)) & 255u] << (8 * j);
                    *reinterpret_cast<uint32_t*>(dst + i) = o;
                }
            } else {
                for (int64_t i = tid; i < n; i += nt) dst[i] = lut[i % 3][src[i]];
            }
        }
        ++done;
        src = dst;
        __syncthreads();                                    // (workgroup-scope fence: the second operation reads what this one wrote)
    }
}

}  // namespace

extern "C" int advmix_make_views(const uint8_t* base, const uint8_t* aug, const int32_t* grid, const float* mean,
                                 const float* std_, float* v0, float* v1, float* v2, int B, int H, int W,
                                 void* stream) {
    if (!base || !mean || !std_ || !v0 || B <= 0 || H <= 0 || W <= 0) return ADVMIX_EINVAL;
    const int64_t total = (int64_t)B * H * W;
    const bool vec = W % 4 == 0 && (((uintptr_t)base | (uintptr_t)aug) & 3) == 0 &&
                     (((uintptr_t)v0 | (uintptr_t)v1 | (uintptr_t)v2) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(make_views_kernel<true>, dim3(stream_blocks(total / 4)), dim3(256), 0, (hipStream_t)stream,
                           base, aug, grid, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2], v0, v1, v2, B, H, W);
    else
        hipLaunchKernelGGL(make_views_kernel<false>, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                           base, aug, grid, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2], v0, v1, v2, B, H, W);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_render_targets(const double* joints, const double* vis, const int32_t* grid, const float* g,
                                     int tmp_size, const float* joints_weight, float* target, float* target_weight,
                                     double* vis_out, int B, int J, int H, int W, int Hh, int Wh, void* stream) {
    if (!joints || !vis || !g || !target || !target_weight || B <= 0 || J <= 0 || tmp_size < 0) return ADVMIX_EINVAL;
    if (H <= 0 || W <= 0 || Hh <= 0 || Wh <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(render_targets_kernel, dim3(B * J), dim3(256), 0, (hipStream_t)stream, joints, vis, grid, g,
                       tmp_size, joints_weight, target, target_weight, vis_out, J, H, W, Hh, Wh);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// The AutoAugment view of a batch: out[b] = op2(op1(base[b])) with ops[b] = {code1, param1, code2, param2} (int32;
// code 0 = operation did not fire; posterize: param = the bit mask; solarize / sharpness: param = the float's bits -
// threshold / blend factor).  ``tmp``: B*H*W*3 bytes of scratch (images that take two operations).  One workgroup per image.
extern "C" int advmix_autoaug(const uint8_t* base, const int32_t* ops, uint8_t* tmp, uint8_t* out, int B, int H, int W,
                              void* stream) {
    if (!base || !ops || !tmp || !out || B <= 0 || H <= 0 || W <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(autoaug_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, base, ops, tmp, out, H, W);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}
