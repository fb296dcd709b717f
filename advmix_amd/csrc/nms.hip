// lib/nms on gfx950.
//   nms_mask_kernel : the reference's only CUDA kernel (lib/nms/nms_kernel.cu:33-77) rebuilt for a
//                     64-lane wavefront: one wave per 64x64 tile of the pairwise IoU matrix, the
//                     64-bit suppression word of a row is produced directly as the wave's
//                     __ballot over the 64 column lanes (no per-thread 64-iteration loop, no LDS).
//   advmix_nms_host : drop-in for `_nms` (gpu_nms.hpp:1-2): H2D, mask, D2H, sequential greedy
//                     OR-reduce on the host (nms_kernel.cu:126-138).
//   oks_matrix      : float64 OKS similarity (lib/nms/nms.py:75-94) for all pairs.
// Bit-exactness: devIoU is evaluated in fp32 with one IEEE operation per statement; the build
// uses -ffp-contract=off for this file's arithmetic (no FMA contraction) and IEEE division.
#include "common.h"
#include <stdio.h>
#include <vector>
#include <string.h>

namespace {

__device__ __forceinline__ float dev_iou(const float* a, const float* b) {
#pragma clang fp contract(off)
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
    float interS = width * height;
    float Sa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
    float Sb = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
    return __fdiv_rn(interS, (Sa + Sb - interS));
}

// grid (col_blocks, row_blocks), block = 256 threads = 4 waves; wave w handles rows
// row_start*64 + w*16 .. +16 of the tile, lane = column within the tile.
__global__ __launch_bounds__(256) void nms_mask_kernel(int n, float thresh, const float* __restrict__ boxes,
                                                       unsigned long long* __restrict__ mask) {
    const int col_blk = blockIdx.x, row_blk = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col_blocks = (n + 63) / 64;
    const int cj = col_blk * 64 + lane;
    float cb[4] = {0.f, 0.f, 0.f, 0.f};
    const bool cvalid = cj < n;
    if (cvalid) {
#pragma unroll
        for (int k = 0; k < 4; ++k) cb[k] = boxes[cj * 5 + k];
    }
    for (int rr = 0; rr < 16; ++rr) {
        const int ri = row_blk * 64 + w * 16 + rr;       // wave-uniform
        if (ri >= n) break;
        float rb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) rb[k] = boxes[ri * 5 + k];
        bool hit = false;
        if (cvalid && (row_blk != col_blk || cj > ri)) hit = dev_iou(rb, cb) > thresh;
        unsigned long long word = __ballot(hit);
        if (lane == 0) mask[(int64_t)ri * col_blocks + col_blk] = word;
    }
}

// OKS of person i against every person j (lib/nms/nms.py:75-94), fp64.  ``np.sum`` over the K per-joint terms is
// numpy's pairwise summation: for 8 <= K <= 128 eight running sums over strides of 8, combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the K % 8 tail added one by one; for K < 8 a plain running sum.
// The kernel adds in exactly that order, so the only difference left against the reference is the last-bit
// rounding of exp() (device libm vs the host's).
// ``use_vis`` (nms.py:90-92, ``in_vis_thre``): only the joints with d's visibility > vis_thre count - the reference's
// ``list(vg > t) and list(vd > t)`` IS ``list(vd > t)`` - and numpy sums the COMPACTED terms, so the pairwise order runs
// over the surviving joints (counted first); no survivor: 0.0.
template <bool VIS>
__global__ void oks_matrix_kernel(const double* __restrict__ gk, const double* __restrict__ ga,
                                  const double* __restrict__ dk, const double* __restrict__ da,
                                  const double* __restrict__ sigmas, int n, int K, double vis_thre,
                                  double* __restrict__ ious) {
#pragma clang fp contract(off)                            /* numpy rounds dx**2, dy**2 and their sum separately: no fma */
    int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;      // row = person i of g, column = person j of d
    if (j >= n) return;
    const double* g = gk + (int64_t)i * K * 3;
    const double* d = dk + (int64_t)j * K * 3;
    const double eps = 2.220446049250313e-16;             // np.spacing(1)
    const double denom = (ga[i] + da[j]) / 2 + eps;
    auto term = [&](int k) {
        double var = (sigmas[k] * 2) * (sigmas[k] * 2);
        double dx = d[3 * k] - g[3 * k], dy = d[3 * k + 1] - g[3 * k + 1];
        double e = (dx * dx + dy * dy) / var / denom / 2;
        return exp(-e);
    };
    int m = K;                                             // terms that enter the sum
    if (VIS) {
        m = 0;
        for (int k = 0; k < K; ++k) m += d[3 * k + 2] > vis_thre;
    }
    double s = 0.0;
    if (m < 8) {
        for (int k = 0; k < K; ++k)
            if (!VIS || d[3 * k + 2] > vis_thre) s += term(k);
    } else {
        // c = index among the surviving terms: the first 8 start the eight running sums, c < m - m % 8 add to sum c % 8,
        // the rest are the tail
        double r[8];
        const int body = m - (m % 8);
        int c = 0, k = 0;
        for (; c < body; ++k) {
            if (VIS && !(d[3 * k + 2] > vis_thre)) continue;
            if (c < 8) r[c] = term(k); else r[c & 7] += term(k);
            ++c;
        }
        s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; k < K; ++k)
            if (!VIS || d[3 * k + 2] > vis_thre) s += term(k);
    }
    ious[(int64_t)i * n + j] = m ? s / m : 0.0;
}

// Greedy pass of oks_nms (lib/nms/nms.py:97-125) on the device: candidates in ``order`` (score-descending, the host's
// argsort - numpy's tie order is part of the result); a candidate survives unless an EARLIER SURVIVOR has OKS > thresh
// with it (the reference keeps ``oks_ovr <= thresh``).  One workgroup; position p of the order is decided in round p
// (uniform), then all threads strike the later positions it suppresses.  Only the kept indices leave the GPU.
__global__ __launch_bounds__(256) void oks_greedy_kernel(const double* __restrict__ ious, const int* __restrict__ order, int n,
                                                         double thresh, int* __restrict__ keep, int* __restrict__ count) {
    extern __shared__ int dead[];                          // [n]
    for (int j = threadIdx.x; j < n; j += blockDim.x) dead[j] = 0;
    __syncthreads();
    int kept = 0;
    for (int p = 0; p < n; ++p) {
        if (!dead[p]) {                                    // uniform: dead[] is only written between barriers
            const int i = order[p];
            if (threadIdx.x == 0) keep[kept] = i;
            ++kept;
            for (int q = p + 1 + threadIdx.x; q < n; q += blockDim.x)
                if (!(ious[(int64_t)i * n + order[q]] <= thresh)) dead[q] = 1;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = kept;
}

// soft_oks_nms's rescoring loop (lib/nms/nms.py:139-177) on the device.  The reference keeps the candidate at the head of
// the (score-descending) order, multiplies every remaining score by exp(-oks^2 / thresh) (float64) and re-sorts the rest
// with ``scores.argsort()[::-1]``, at most ``max_dets`` (20) times.  One workgroup: ``order`` / ``score`` ping-pong between
// two halves of the caller's scratch; the re-sort is a rank computation (position = how many of the others sort before me:
// strictly greater, or equal and LATER in the current arrangement - a stable ascending sort read backwards).  Without
// exact ties that IS numpy's argsort()[::-1]; among EQUAL scores numpy's order depends on its build (introsort, or the
// AVX-512 argsort, which is unstable even for nine elements), so there the reference is defined only up to "a maximum of
// the scores left is kept next", which is what this does.  NaN scores sort as numpy sorts them (last ascending = first
// after the reversal).  Only the kept indices leave the GPU.
__device__ __forceinline__ bool soft_lt(double a, double b) { return a < b || (b != b && a == a); }   // numpy's order, NaN last

__global__ __launch_bounds__(256) void soft_oks_kernel(const double* __restrict__ ious, const int* __restrict__ order0,
                                                       const double* __restrict__ score0, int n, double thresh, int max_dets,
                                                       double* __restrict__ sc, int* __restrict__ od, int* __restrict__ keep,
                                                       int* __restrict__ count) {
#pragma clang fp contract(off)
    double* sa = sc;           double* sb = sc + n;        // current / rescored-unsorted (then swapped)
    int* oa = od;              int* ob = od + n;
    for (int j = threadIdx.x; j < n; j += blockDim.x) { sa[j] = score0[j]; oa[j] = order0[j]; }
    __syncthreads();
    int m = n, kept = 0;
    while (m > 0 && kept < max_dets) {
        const int i = oa[0];
        if (threadIdx.x == 0) keep[kept] = i;
        ++kept;
        --m;                                               // the rest: positions 1 .. m of the current arrangement
        for (int q = threadIdx.x; q < m; q += blockDim.x) {
            const double o = ious[(int64_t)i * n + oa[q + 1]];
            sb[q] = sa[q + 1] * exp(-(o * o) / thresh);    // scores[1:] * np.exp(-oks_ovr ** 2 / thresh)
            ob[q] = oa[q + 1];
        }
        __syncthreads();
        for (int q = threadIdx.x; q < m; q += blockDim.x) {
            const double v = sb[q];
            int pos = 0;
            for (int r = 0; r < m; ++r) {
                const double u = sb[r];
                pos += (soft_lt(v, u) || (!soft_lt(u, v) && r > q)) ? 1 : 0;
            }
            sa[pos] = v;
            oa[pos] = ob[q];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = kept;
}

}  // namespace

extern "C" int advmix_nms_mask(const float* boxes_dev, int n, float thresh, uint64_t* mask_dev, void* stream) {
    if (!boxes_dev || !mask_dev || n <= 0) return ADVMIX_EINVAL;
    int cb = (n + 63) / 64;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb), dim3(256), 0, (hipStream_t)stream, n, thresh, boxes_dev,
                       (unsigned long long*)mask_dev);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

extern "C" int advmix_nms_host(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
                               float nms_overlap_thresh, int device_id) {
    if (!keep_out || !num_out) return ADVMIX_EINVAL;
    if (boxes_num == 0) { *num_out = 0; return ADVMIX_OK; }
    if (!boxes_host || boxes_num < 0 || boxes_dim != 5) return ADVMIX_EINVAL;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return ADVMIX_ELAUNCH;
    if (cur != device_id && hipSetDevice(device_id) != hipSuccess) return ADVMIX_ELAUNCH;
    const int cb = (boxes_num + 63) / 64;
    float* bd = nullptr;
    unsigned long long* md = nullptr;
    int rc = ADVMIX_OK;
    std::vector<unsigned long long> mh((size_t)boxes_num * cb), remv(cb, 0ULL);
    if (hipMalloc(&bd, sizeof(float) * 5 * boxes_num) != hipSuccess) return ADVMIX_ELAUNCH;
    if (hipMalloc(&md, sizeof(unsigned long long) * (size_t)boxes_num * cb) != hipSuccess) { (void)hipFree(bd); return ADVMIX_ELAUNCH; }
    if (hipMemcpy(bd, boxes_host, sizeof(float) * 5 * boxes_num, hipMemcpyHostToDevice) != hipSuccess) rc = ADVMIX_ELAUNCH;
    if (rc == ADVMIX_OK) rc = advmix_nms_mask(bd, boxes_num, nms_overlap_thresh, (uint64_t*)md, nullptr);
    if (rc == ADVMIX_OK &&
        hipMemcpy(mh.data(), md, sizeof(unsigned long long) * mh.size(), hipMemcpyDeviceToHost) != hipSuccess)
        rc = ADVMIX_ELAUNCH;
    (void)hipFree(bd);
    (void)hipFree(md);
    if (rc != ADVMIX_OK) return rc;
    int k = 0;
    for (int i = 0; i < boxes_num; ++i) {
        int nb = i / 64, ib = i % 64;
        if (!(remv[nb] & (1ULL << ib))) {
            keep_out[k++] = i;
            const unsigned long long* p = mh.data() + (size_t)i * cb;
            for (int j = nb; j < cb; ++j) remv[j] |= p[j];
        }
    }
    *num_out = k;
    return ADVMIX_OK;
}

// The reference's own native symbol (lib/nms/gpu_nms.hpp:1-2; C++ linkage - `_Z4_nmsPiS_PKfiifi` - bound by the Cython
// wrapper gpu_nms.pyx:10-11,31): same arguments, same contract - returns nothing, an error is PRINTED and execution
// continues (CUDA_CHECK, nms_kernel.cu:11-18), here with *num_out = 0 so that the caller's `keep[:num_out]` is empty rather
// than uninitialised.  With this symbol the reference's gpu_nms.pyx links against -ladvmix_hip in place of nms_kernel.cu
// (INTEGRATION.md section 3); include/gpu_nms.hpp carries the declaration.
void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float nms_overlap_thresh,
          int device_id) {
    const int rc = advmix_nms_host(keep_out, num_out, boxes_host, boxes_num, boxes_dim, nms_overlap_thresh, device_id);
    if (rc != ADVMIX_OK) {
        printf("_nms: %s\n", rc == ADVMIX_EINVAL ? "invalid argument" : hipGetErrorString(hipGetLastError()));
        if (num_out) *num_out = 0;
    }
}

extern "C" int advmix_oks_matrix(const double* kpts, const double* areas, const double* sigmas, int n, int K,
                                 double* ious, void* stream) {
    if (!kpts || !areas || !sigmas || !ious || n <= 0 || K <= 0 || K > 128) return ADVMIX_EINVAL;   // (numpy recurses past 128)
    hipLaunchKernelGGL(oks_matrix_kernel<false>, dim3(cdiv(n, 64), n), dim3(64), 0, (hipStream_t)stream, kpts, areas, kpts,
                       areas, sigmas, n, K, 0.0, ious);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// oks_iou itself (lib/nms/nms.py:75-94), ng ground-truth persons against nd detections: ious[ng][nd].  use_vis != 0 is the
// reference's ``in_vis_thre`` (nms.py:90-92): joints of DETECTION j whose visibility d_kpts[j][3k + 2] is not above
// vis_thre leave the sum and the divisor; a pair with no joint left has OKS 0.
extern "C" int advmix_oks_iou(const double* g_kpts, const double* g_areas, int ng, const double* d_kpts, const double* d_areas,
                              int nd, const double* sigmas, int K, int use_vis, double vis_thre, double* ious, void* stream) {
    if (!g_kpts || !g_areas || !d_kpts || !d_areas || !sigmas || !ious || ng <= 0 || nd <= 0 || K <= 0 || K > 128)
        return ADVMIX_EINVAL;
    if (use_vis && vis_thre != vis_thre) return ADVMIX_EINVAL;                                      // NaN threshold
    const dim3 grid(cdiv(nd, 64), ng);
    if (use_vis)
        hipLaunchKernelGGL(oks_matrix_kernel<true>, grid, dim3(64), 0, (hipStream_t)stream, g_kpts, g_areas, d_kpts, d_areas,
                           sigmas, nd, K, vis_thre, ious);
    else
        hipLaunchKernelGGL(oks_matrix_kernel<false>, grid, dim3(64), 0, (hipStream_t)stream, g_kpts, g_areas, d_kpts, d_areas,
                           sigmas, nd, K, 0.0, ious);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// oks_nms's greedy pass on the device (see oks_greedy_kernel).  ious: [n][n] fp64 from advmix_oks_matrix; order: int32
// [n] candidate indices, best first; keep_out: int32 [n]; count_out: int32 [1].  n <= 8192.
extern "C" int advmix_oks_greedy(const double* ious, const int* order, int n, double thresh, int* keep_out, int* count_out,
                                 void* stream) {
    if (!ious || !order || !keep_out || !count_out || n <= 0 || n > 8192) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(oks_greedy_kernel, dim3(1), dim3(256), n * sizeof(int), (hipStream_t)stream, ious, order, n, thresh,
                       keep_out, count_out);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// soft_oks_nms's loop on the device (see soft_oks_kernel).  ious: [n][n] fp64; order: int32 [n] candidate indices, best
// first (the host's argsort, as for advmix_oks_greedy); scores_sorted: fp64 [n] = scores[order]; scratch_scores: fp64 [2n],
// scratch_order: int32 [2n] (caller-owned, no hidden allocation); keep_out: int32 [max_dets]; count_out: int32 [1].
extern "C" int advmix_soft_oks_greedy(const double* ious, const int* order, const double* scores_sorted, int n, double thresh,
                                      int max_dets, double* scratch_scores, int* scratch_order, int* keep_out, int* count_out,
                                      void* stream) {
    if (!ious || !order || !scores_sorted || !scratch_scores || !scratch_order || !keep_out || !count_out || n <= 0 ||
        n > 8192 || max_dets <= 0)
        return ADVMIX_EINVAL;
    if (thresh != thresh || thresh == 0.0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(soft_oks_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ious, order, scores_sorted, n, thresh,
                       max_dets, scratch_scores, scratch_order, keep_out, count_out);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}
