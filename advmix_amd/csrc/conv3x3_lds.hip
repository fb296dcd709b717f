// 3x3 / stride 1 / pad 1 convolution with the input patch resident in LDS ("im2col in LDS"),
// persistent workgroups, double-buffered across work items.
//
// Why (all measured on MI355X, 3x3 32->32 @64x48, B=32; ideal MFMA time 11.5 us):
//   conv_igemm  28.1 us  both operands via LDS, two barriers per 16-deep k tile
//   conv_direct 26.2 us  A fragments straight from global memory.  With the MFMAs compiled out it
//               still takes 18.4 us and forcing every tap onto the same pixels (perfect L1
//               locality) changes nothing: the wall is the vector-memory pipe (L1/TA, ~30 B/clk/CU
//               for 32-row x 32-byte fragment loads) - each pixel is re-read once per tap.
//   one-shot LDS patch (load patch + weights, barrier, 144 MFMAs): 26.2 us - 2.7x less L1 traffic
//               but no overlap of load and multiply inside a workgroup, 1.5 waves of workgroups.
// So: ONE workgroup per CU walks a stream of work items (tile x 32-channel chunk).  While the
// four waves multiply item i out of LDS buffer i&1 (144 MFMAs per wave, both operands by
// conflict-free ds_read_b128, no barrier inside), the loads of item i+1 (a (8+2)x(16+2)-pixel
// patch chunk, 128-B pixel rows, and the [9][32][32] weight chunk) are in flight into registers;
// they are written to the other buffer after the multiply and ONE barrier separates items.
// 9216 MFMA cycles per item hide any load latency, so the pipe idles only for the first load and
// the ~100-cycle item seams.
// Result: 27.9 us on that shape (no better), 28.7 vs 34.9 us on 128->128 @16x12, 161 vs 170 us on
// 256->32 @64x48.  tools/mfma_peak.hip explains the floor: a register-only kernel issuing the
// same 432 MFMAs per SIMD and nothing else takes 16 us launch to launch (109 TFLOP/s; 146-150
// TFLOP/s only in ~0.5 ms kernels) - at B=32 these convs are launch-ramp sized, so the remaining
// lever is work per launch, not the inner loop.
//
// One item = 8 x 16 output pixels x 32 output channels x 32 input channels; wave w owns rows
// 2w, 2w+1.  flip = 0: forward (w = [Co][3][3][Ci]); flip = 1: the input gradient (x = dY, w =
// transposed weights [Ci][3][3][Co], taps mirrored) - 3x3/s1/p1 is its own adjoint up to the mirror.
#include "common.h"
#include <stdlib.h>

namespace c3 {

constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, KC = 32, BN = 32, LD = KC + 4;
constexpr int PATCH = PH * PW;                       // 180 pixels
constexpr int XS = PATCH * LD, WS = 9 * BN * LD;     // floats per buffer
constexpr unsigned OOB = 0x80000000u;
constexpr int XSL = (PATCH * 8 + 255) / 256;         // float4 slots per thread: patch (6)
constexpr int WSL = (9 * BN * 8) / 256;              // weights (9)

struct P3 {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    int N, H, W, Ci, Co;
    int tiles_h, tiles_w, co_blocks;
    int xbytes, wbytes;
    int flip;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

__global__ __launch_bounds__(256) void conv3x3_lds(P3 p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [2][XS] patches, [2][WS] weights
    float* Xs = smem;
    float* Ws = smem + 2 * XS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int spatial = p.N * p.tiles_h * p.tiles_w;
    const int total_tiles = spatial * p.co_blocks;          // co block outermost: neighbours share weights
    const int nchunks = p.Ci / KC;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);

    // fixed LDS destinations of this thread's staging slots
    int x_dst[XSL], x_ph[XSL], x_pw[XSL], x_c4[XSL];
#pragma unroll
    for (int i = 0; i < XSL; ++i) {
        int s = tid + 256 * i;
        int px = s >> 3;
        x_c4[i] = (s & 7) * 4;
        x_ph[i] = px / PW;
        x_pw[i] = px - x_ph[i] * PW;
        x_dst[i] = s < PATCH * 8 ? px * LD + x_c4[i] : -1;
    }
    int w_dst[WSL], w_row[WSL], w_c4[WSL];
#pragma unroll
    for (int i = 0; i < WSL; ++i) {
        int s = tid + 256 * i;
        w_row[i] = s >> 3;                                  // tap * 32 + co
        w_c4[i] = (s & 7) * 4;
        w_dst[i] = w_row[i] * LD + w_c4[i];
    }
    const int py = 2 * wid + (l31 >> 4), px = l31 & 15;     // this lane's output pixel in the tile

    // ---- stream cursor of the item being LOADED ------------------------------------------------
    int ltile = blockIdx.x, lchunk = 0;
    f32x4 xv[XSL], wv[WSL];
    auto issue = [&]() {                                     // loads of item (ltile, lchunk)
        const int sp = ltile % spatial, cb = ltile / spatial;
        const int tw_i = sp % p.tiles_w;
        const int th_i = (sp / p.tiles_w) % p.tiles_h;
        const int n = sp / (p.tiles_w * p.tiles_h);
        const int h0 = th_i * TH - 1, w0 = tw_i * TW - 1, c0 = lchunk * KC;
#pragma unroll
        for (int i = 0; i < XSL; ++i) {
            int h = h0 + x_ph[i], w = w0 + x_pw[i];
            bool ok = x_dst[i] >= 0 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            xv[i] = bload(xr, ok ? (unsigned)((((n * p.H + h) * p.W + w) * p.Ci + c0 + x_c4[i]) * 4) : OOB);
        }
#pragma unroll
        for (int i = 0; i < WSL; ++i) {
            int co = cb * BN + (w_row[i] & 31), tap = w_row[i] >> 5;
            wv[i] = bload(wr, co < p.Co ? (unsigned)(((co * 9 + tap) * p.Ci + c0 + w_c4[i]) * 4) : OOB);
        }
        if (++lchunk == nchunks) { lchunk = 0; ltile += gridDim.x; }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XSL; ++i)
            if (x_dst[i] >= 0) *reinterpret_cast<f32x4*>(&Xs[buf * XS + x_dst[i]]) = xv[i];
#pragma unroll
        for (int i = 0; i < WSL; ++i) *reinterpret_cast<f32x4*>(&Ws[buf * WS + w_dst[i]]) = wv[i];
    };

    if (ltile >= total_tiles) return;
    // (two interleaved accumulator chains were measured: no change - a single dependent
    // 32x32x2 chain already issues back to back, tools/mfma_peak.hip)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    issue();
    stage(0);
    __syncthreads();

    int ctile = blockIdx.x, cchunk = 0, buf = 0;             // item being MULTIPLIED
    while (true) {
        const bool have_next = ltile < total_tiles;
        if (have_next) issue();
        const float* xb = Xs + buf * XS;
        const float* wb = Ws + buf * WS;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, s = tap % 3;
            const int pr = p.flip ? 2 - r : r, ps = p.flip ? 2 - s : s;
            const float* ap = xb + ((py + pr) * PW + px + ps) * LD + lh * 4;
            const float* bp = wb + (tap * BN + l31) * LD + lh * 4;
#pragma unroll
            for (int q = 0; q < KC / 8; ++q) {
                f32x4 a = *reinterpret_cast<const f32x4*>(ap + q * 8);
                f32x4 b = *reinterpret_cast<const f32x4*>(bp + q * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
            }
        }
        if (++cchunk == nchunks) {                           // tile finished: write it, start the next
            const int sp = ctile % spatial, cb = ctile / spatial;
            const int tw_i = sp % p.tiles_w;
            const int th_i = (sp / p.tiles_w) % p.tiles_h;
            const int n = sp / (p.tiles_w * p.tiles_h);
            const int col = cb * BN + l31;
            if (col < p.Co) {
                const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int i = (r & 3) + 8 * (r >> 2) + 4 * lh; // pixel inside the wave's 2 x 16 strip
                    int h = th_i * TH + 2 * wid + (i >> 4), w = tw_i * TW + (i & 15);
                    if (h < p.H && w < p.W) p.y[(((int64_t)n * p.H + h) * p.W + w) * p.Co + col] = acc[r] + bv;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            cchunk = 0;
            ctile += gridDim.x;
        }
        if (!have_next) break;
        stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
}

}  // namespace c3

// 3x3 stride-1 pad-1 only; returns -1 if not eligible
int advmix_conv3x3_lds_dispatch(int flip, const float* x, const float* w, const float* bias, float* y, int N, int H,
                                int W, int Ci, int Co, hipStream_t st) {
    static int attr_ok = -1;
    if (attr_ok < 0)
        attr_ok = hipFuncSetAttribute((const void*)c3::conv3x3_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      2 * (c3::XS + c3::WS) * (int)sizeof(float)) == hipSuccess ? 1 : 0;
    const AdvmixOpts& o = advmix_opts();
    const int enabled = attr_ok && o.conv3, grid_cap = o.conv3_grid;
    if (!enabled || Ci % 32 != 0) return -1;
    // tiles are 8 x 16: skip maps where the padding waste would exceed ~1/3
    const int th = (H + c3::TH - 1) / c3::TH, tw = (W + c3::TW - 1) / c3::TW;
    if ((int64_t)th * c3::TH * tw * c3::TW * 2 > (int64_t)H * W * 3) return -1;
    const int64_t xb = (int64_t)N * H * W * Ci * 4, wb = (int64_t)Co * 9 * Ci * 4;
    if (xb >= 0x7fffffffLL || wb >= 0x7fffffffLL) return -1;
    const int cob = (Co + 31) / 32;
    c3::P3 p{x, w, bias, y, N, H, W, Ci, Co, th, tw, cob, (int)xb, (int)wb, flip};
    const int64_t tiles = (int64_t)N * th * tw * cob;
    if (tiles * (Ci / 32) < o.conv3_min_items) return -1;   // too little work to pipeline: conv_direct (+ split-K)
    const int grid = (int)(tiles < grid_cap ? tiles : grid_cap);
    hipLaunchKernelGGL(c3::conv3x3_lds, dim3(grid), dim3(256), 2 * (c3::XS + c3::WS) * sizeof(float), st, p);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}
