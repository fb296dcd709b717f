// validate()-side device kernels (HBM / latency bound): the flip test and get_final_preds.
// Reference sites: lib/core/function.py:240-261 (flip test), lib/utils/transforms.py:16-41 (flip_back),
// lib/core/inference.py:22-95 (get_max_preds, get_final_preds), lib/utils/transforms.py:57-107
// (transform_preds / get_affine_transform with rot = 0, inv = 1 / affine_transform).
// The reference moves every heat-map to the host twice per batch (flip_back and get_final_preds run in
// numpy); here only [B,J,3] floats cross PCIe.
#include "common.h"

namespace {

static int stream_blocks(int64_t total) {
    int64_t b = (total + 255) / 256;
    const int cap = advmix_stream_cap();
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

// input.flip(3) (function.py:241): x dense NCHW; y dense NCHW or NHWC.  Threads walk x in memory order.
__global__ __launch_bounds__(256) void flip_w_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int H,
                                                     int W, int64_t total, int y_nhwc) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i / W;
        int w = (int)(i - r * W);
        const int wf = W - 1 - w;
        if (!y_nhwc) {
            y[r * W + wf] = x[i];
        } else {
            int64_t bc = r / H;
            int h = (int)(r - bc * H);
            int64_t b = bc / C;
            int c = (int)(bc - b * C);
            y[((b * H + h) * W + wf) * C + c] = x[i];
        }
    }
}

// F[b,j,h,w] = flipped[b, partner[j], h, W-1-w]                      (flip_back, transforms.py:24,36-39)
// F'[..., w] = F[..., w-1] for w >= 1, F'[..., 0] = F[..., 0]        (SHIFT_HEATMAP, function.py:257-259)
// y = merge ? (out + F') * 0.5f : F'                                 (function.py:261)
__global__ __launch_bounds__(256) void flip_merge_kernel(const float* __restrict__ out,
                                                         const float* __restrict__ flipped,
                                                         const int32_t* __restrict__ partner, float* __restrict__ y,
                                                         int J, int H, int W, int64_t total, int nhwc, int shift,
                                                         int merge) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t b;
        int j, h, w;
        if (nhwc) {                                    // i = ((b*H + h)*W + w)*J + j
            int64_t p = i / J;
            j = (int)(i - p * J);
            int64_t q = p / W;
            w = (int)(p - q * W);
            b = q / H;
            h = (int)(q - b * H);
        } else {                                       // i = ((b*J + j)*H + h)*W + w
            int64_t p = i / W;
            w = (int)(i - p * W);
            int64_t q = p / H;
            h = (int)(p - q * H);
            b = q / J;
            j = (int)(q - b * J);
        }
        const int ws = (shift && w >= 1) ? w - 1 : w;
        const int wf = W - 1 - ws;
        const int pj = partner[j];
        const int64_t src = nhwc ? ((b * H + h) * W + wf) * J + pj : ((b * J + pj) * H + h) * W + wf;
        const float f = flipped[src];
        y[i] = merge ? (out[i] + f) * 0.5f : f;
    }
}

__device__ __forceinline__ float hm_at(const float* hm, int nhwc, int64_t b, int j, int J, int W, int HW, int py,
                                       int px) {
    const int p = py * W + px;
    return nhwc ? hm[(b * HW + p) * J + j] : hm[(b * J + j) * HW + p];
}

__device__ __forceinline__ float signf(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : d); }   // numpy.sign

// One wave per (b, j): first-occurrence argmax + max (get_max_preds, inference.py:30-47), quarter-pixel shift
// toward the higher neighbour (inference.py:64-76), then the inverse crop transform.  With rot = 0 the three
// point pairs get_affine_transform hands to cv2.getAffineTransform (transforms.py:79-98) are
//   dst (heat-map): (W/2, H/2), (W/2, H/2 - W/2), (0, H/2 - W/2)
//   src (image)   : (cx, cy),   (cx, cy1),        (cx - d, cy1)    cy1 = f32(cy - 100*s_x), d = f32(cy - cy1)
// all rounded to float32 exactly where the reference stores them in float32 arrays.  The affine map through
// them is axis-aligned, so the 6x6 solve reduces to two slopes; it is evaluated in fp64 like the reference
// (cv2 solves in double, affine_transform multiplies in double) and rounded to fp32 once (the store into the
// float32 ``preds``, inference.py:78-84).
__global__ __launch_bounds__(256) void final_preds_kernel(const float* __restrict__ hm, int nhwc,
                                                          const float* __restrict__ center,
                                                          const float* __restrict__ scale, int B, int J, int H, int W,
                                                          int post_process, float* __restrict__ coords_out,
                                                          float* __restrict__ preds, float* __restrict__ maxvals) {
    const int lane = threadIdx.x & 63;
    const int64_t bj = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bj >= (int64_t)B * J) return;
    const int64_t b = bj / J;
    const int j = (int)(bj - b * J);
    const int HW = H * W;
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
    if (lane != 0) return;
    if (bi == 0x7fffffff) bi = 0;
    float x = (float)(bi % W), y = (float)(bi / W);
    if (!(best > 0.0f)) { x = 0.f; y = 0.f; }                          // pred_mask (inference.py:43-46)
    if (post_process) {
        const int px = (int)floorf(x + 0.5f), py = (int)floorf(y + 0.5f);
        if (1 < px && px < W - 1 && 1 < py && py < H - 1) {
            const float dx = hm_at(hm, nhwc, b, j, J, W, HW, py, px + 1) - hm_at(hm, nhwc, b, j, J, W, HW, py, px - 1);
            const float dy = hm_at(hm, nhwc, b, j, J, W, HW, py + 1, px) - hm_at(hm, nhwc, b, j, J, W, HW, py - 1, px);
            x += signf(dx) * 0.25f;
            y += signf(dy) * 0.25f;
        }
    }
    if (coords_out) { coords_out[bj * 2] = x; coords_out[bj * 2 + 1] = y; }
    maxvals[bj] = best;
    const float cx = center[b * 2], cy = center[b * 2 + 1];
    const float sw = scale[b * 2] * 200.0f;                            // scale_tmp[0], float32 (transforms.py:73-74)
    const float cy1 = (float)((double)cy + (double)(sw * -0.5f));      // src[1].y (transforms.py:85)
    const float d = cy - cy1;                                          // get_3rd_point: direct[1]
    const float s2x = cx + (-d);                                       // src[2].x
    const double half_w = (double)W * 0.5, half_h = (double)H * 0.5;
    const double mx = ((double)cx - (double)s2x) / half_w;             // d src_x / d dst_x
    const double my = ((double)cy - (double)cy1) / half_w;             // d src_y / d dst_y
    const double tx = (double)cx - mx * half_w, ty = (double)cy - my * half_h;
    preds[bj * 2] = (float)(mx * (double)x + tx);
    preds[bj * 2 + 1] = (float)(my * (double)y + ty);
}

}  // namespace

extern "C" int advmix_flip_w(const float* x, float* y, int B, int C, int H, int W, int y_nhwc, void* stream) {
    if (!x || !y || x == y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return ADVMIX_EINVAL;
    const int64_t total = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(flip_w_kernel, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, C, H, W,
                       total, y_nhwc);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_flip_merge(const float* out, const float* flipped, const int32_t* partner, float* y, int B,
                                 int J, int H, int W, int nhwc, int shift, void* stream) {
    if (!flipped || !partner || !y || y == flipped || B <= 0 || J <= 0 || H <= 0 || W <= 0) return ADVMIX_EINVAL;
    const int64_t total = (int64_t)B * J * H * W;
    hipLaunchKernelGGL(flip_merge_kernel, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, out, flipped,
                       partner, y, J, H, W, total, nhwc, shift, out != nullptr);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_final_preds(const float* hm, int nhwc, const float* center, const float* scale, int B, int J,
                                  int H, int W, int post_process, float* coords, float* preds, float* maxvals,
                                  void* stream) {
    if (!hm || !center || !scale || !preds || !maxvals || B <= 0 || J <= 0 || H <= 0 || W <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(final_preds_kernel, dim3(cdiv((int64_t)B * J, 4)), dim3(256), 0, (hipStream_t)stream, hm, nhwc,
                       center, scale, B, J, H, W, post_process, coords, preds, maxvals);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}
