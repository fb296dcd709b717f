// Weight gradient of a 3x3 / stride 1 / pad 1 convolution with 32 input and 32 output channels (HRNet-W32's
// high-resolution branch: the most frequent weight gradient of the step) with both operands staged ONCE in LDS.
//
// dW[co][kh][kw][ci] = sum over pixels of dY[p][co] * X[p + (kh-1, kw-1)][ci].  The nine taps read the SAME input
// patch shifted by one pixel; wgrad_direct loads each tap's operand from L1/L2 again (1.3 vector-memory instructions
// per MFMA, 38 us for 11.5 us of MFMA work at B = 32 @64x48).  Here a workgroup owns ROWS image rows of one image:
//   phase 1  all 768 threads copy dY[ROWS][W][32] and the zero-padded patch X[ROWS+2][W+2][32] into LDS with 16-byte
//            loads (every input element is fetched once per workgroup; out-of-image halo = zeros);
//   phase 2  12 waves = 4 row groups x 3 kernel rows.  A wave walks its rows two pixels per MFMA step: one
//            ds_read_b32 for the dY fragment, three for the patch at kw = 0, 1, 2 (constant LDS offsets), three
//            v_mfma_f32_32x32x2_f32 into three accumulators - 1.33 LDS reads per MFMA, no global memory, no barriers;
//   phase 3  the row groups' partial tiles are added in a FIXED order through LDS (three passes); the workgroup's
//            32 x 288 partial goes to LDS in dW's layout and all 12 waves add it to dW with fp32 atomics - or, for the
//            deterministic mode, store it to the slab's slice of ``part`` (summed in slab order by the caller's launch).
// One workgroup per CU at ROWS = 8 (256 workgroups at B = 32 @64x48): as few partial tiles as CUs, because merging them
// is what a K split across workgroups costs.  Measured at that shape (tools/variants/wgrad_lds_dbg.patch, -DWL_DBG=...):
// launch 3.8 us + staging 1.6 + MFMA phase 13.5 (11.7 at the matrix peak) + in-workgroup merge 1 + atomics 12 = 32 us
// (wgrad_direct: 40).  The 2.4 M atomics run at 0.8 TB/s whatever their order (walking the tile from a different
// offset per workgroup: no change); storing the tiles and summing them with tail workgroups of the same launch (counter
// hand-off, write-through stores, device-scope loads) costs the same 12 us - two memory round trips between XCDs -
// and a device-scope release fence per workgroup 75 us (32 L2 write-backs queue on each XCD), so the simple form stays.
#include "common.h"
#include <stdio.h>

namespace wgl {

struct LP {
    const float* dy;   // [N, H, W, 32]
    const float* x;    // [N, H, W, 32]
    float* dw;         // [32][9][32]
    float* part;       // [slabs][32*9*32] or null
    int N, H, W, Wp;   // Wp = W rounded up to even
    int rows;          // image rows per slab (multiple of 4)
    int bytes;         // of dy and of x (same shape)
    // round 4: a workgroup walks ``nslab`` consecutive slabs (stage, multiply, stage, multiply ... one accumulator set) before
    // it merges - 1 / nslab of the partial tiles - and one launch serves up to 8 problems of one geometry (the eight 3x3
    // 32 -> 32 convs of an HRNet branch: weight gradients have no consumer before the optimizer step).  Workgroup b belongs
    // to problem b / wgs and walks slabs (b % wgs) * nslab ... of it.  Single problems: nslab = 1, n = 1, pointers in [0].
    int nslab, wgs, n;
    const float* dyv[64];
    const float* xv[64];
    float* dwv[64];
};

constexpr int C = 32;
constexpr int THREADS = 768;
constexpr int TILE = 1024;                     // one 32x32 accumulator tile

constexpr unsigned OOB = 0x80000000u;          // beyond any tensor this kernel accepts: the buffer load returns 0

__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

__global__ __launch_bounds__(THREADS) void wgrad3x3_c32(LP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int rg = wid / 3, kh = wid - rg * 3;
    constexpr int TOT = C * 9 * C;                     // 9216 outputs = 12 x 768
    const int slabs_per_img = p.H / p.rows;
    const int prob = blockIdx.x / p.wgs;
    const int wg = blockIdx.x - prob * p.wgs;
    const float* const pdy = p.dyv[prob];
    const float* const px = p.xv[prob];
    float* const pdw = p.dwv[prob];
    const int Wp = p.Wp, Wx = p.Wp + 2;

    float* ldy = lds;                                  // [rows][Wp][32]
    float* lx = lds + p.rows * Wp * C;                 // [rows+2][Wp+2][32]

    f32x16 acc[3];                                     // kw = 0, 1, 2 of this wave's kernel row, over ALL the slabs it walks
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    int slab = wg * p.nslab;
  for (int it = 0; it < p.nslab; ++it, ++slab) {
    const int n = slab / slabs_per_img;
    const int h0 = (slab - n * slabs_per_img) * p.rows;
    if (it > 0) __syncthreads();                       // the previous slab's multiplies are done with the LDS image

    // ---- phase 1: global -> LDS, 16 bytes per thread and load.  Branch-free: invalid elements (halo outside the image,
    // padding column of an odd width, beyond the tile) get an out-of-range buffer offset and come back as zeros, so a
    // batch of loads is in flight at once; (row, column) advance incrementally (one division per tensor and thread).
    {
        const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)pdy, 0, p.bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)px, 0, p.bytes, 0x00020000);
        const int q4 = (tid & 7) * 16;                 // byte offset of this thread's 4 channels
        constexpr int PSTEP = THREADS / 8;             // pixels between a thread's consecutive loads
        {
            const int ndy = p.rows * Wp;               // pixels
            int pix = tid >> 3;
            int r = pix / Wp, w = pix - r * Wp;
            const int dr = PSTEP / Wp, dwv = PSTEP - dr * Wp;
            const unsigned base = (unsigned)(((n * p.H + h0) * p.W) * C * 4 + q4);
            for (; pix < ndy; ) {
                f32x4 v[4];
                int pp = pix;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = pp < ndy && w < p.W;
                    v[u] = bload4(rdy, ok ? base + (unsigned)((r * p.W + w) * (C * 4)) : OOB);
                    pp += PSTEP; w += dwv; r += dr;
                    if (w >= Wp) { w -= Wp; ++r; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (pix < ndy) *reinterpret_cast<f32x4*>(ldy + pix * C + (tid & 7) * 4) = v[u];
                    pix += PSTEP;
                }
            }
        }
        {
            const int nx = (p.rows + 2) * Wx;
            int pix = tid >> 3;
            int pr = pix / Wx, pc = pix - pr * Wx;
            const int dr = PSTEP / Wx, dc = PSTEP - dr * Wx;
            const int base = ((n * p.H + h0 - 1) * p.W - 1) * C * 4 + q4;      // patch (0, 0) = image (h0 - 1, -1)
            for (; pix < nx; ) {
                f32x4 v[8];
                int pp = pix;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool ok = pp < nx && (unsigned)(h0 + pr - 1) < (unsigned)p.H && (unsigned)(pc - 1) < (unsigned)p.W;
                    v[u] = bload4(rx, ok ? (unsigned)(base + (pr * p.W + pc) * (C * 4)) : OOB);
                    pp += PSTEP; pc += dc; pr += dr;
                    if (pc >= Wx) { pc -= Wx; ++pr; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (pix < nx) *reinterpret_cast<f32x4*>(lx + pix * C + (tid & 7) * 4) = v[u];
                    pix += PSTEP;
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 2: 3 accumulators (kw = 0, 1, 2) per wave, operands from LDS -----------------------------------------
    const int rpg = p.rows >> 2;                       // rows per row group
    for (int rr = 0; rr < rpg; ++rr) {
        const int r = rg * rpg + rr;
        const float* ap = ldy + (r * Wp + lh) * C + l31;
        const float* bp = lx + ((r + kh) * Wx + lh) * C + l31;
        int w = 0;
        for (; w + 8 <= Wp; w += 8) {                  // four MFMA steps (8 pixels) of LDS reads ahead of their MFMAs
            float a[4], b[4][3];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a[s] = ap[(w + 2 * s) * C];
#pragma unroll
                for (int t = 0; t < 3; ++t) b[s][t] = bp[(w + 2 * s + t) * C];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s][t], acc[t], 0, 0, 0);
        }
        for (; w < Wp; w += 2) {
            const float a1 = ap[w * C];
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bp[(w + t) * C], acc[t], 0, 0, 0);
        }
    }
  }                                                    // next slab of this workgroup

    // ---- phase 3: row groups 1..3 are added to row group 0 in that order, then one partial per workgroup ------------
    float* red = lds;                                  // 3 waves x 3 tiles
    for (int g = 1; g < 4; ++g) {
        __syncthreads();                               // patch (g = 1) / previous pass consumed
        if (rg == g) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(kh * 3 + t) * TILE + r * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (rg == 0) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += red[(kh * 3 + t) * TILE + r * 64 + lane];
        }
    }
    // the workgroup's tile -> LDS ([co][tap][ci], dW's own layout) -> global with ALL waves, 256 bytes per wave and
    // instruction.  Every workgroup adds to the same 36 KB: each starts at a different offset (the atomic unit serialises
    // same-line updates; workgroups walking the tile in lock step would queue on one line while the others idle).
    __syncthreads();
    if (rg == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * lh;
                red[co * (9 * C) + (kh * 3 + t) * C + l31] = acc[t][r];
            }
    }
    __syncthreads();
    {
        if (p.part) {
            const __amdgpu_buffer_rsrc_t ro =
                __builtin_amdgcn_make_buffer_rsrc((void*)(p.part + (int64_t)blockIdx.x * TOT), 0, TOT * 4, 0x00020000);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int j = 0; j < TOT / 4 / THREADS; ++j) {
                const int i4 = (tid + j * THREADS) * 4;
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(red + i4), ro, i4 * 4, 0, 0);
            }
        } else {
            int idx = tid + ((int)blockIdx.x * 37 % (TOT / 64)) * 64;
#pragma unroll
            for (int j = 0; j < TOT / THREADS; ++j) {
                if (idx >= TOT) idx -= TOT;
                atomicAdd(pdw + idx, red[idx]);
                idx += THREADS;
            }
        }
    }
}

}  // namespace wgl

// The dynamic-LDS limit of wgrad3x3_c32, shared by BOTH dispatchers and only ever raised (each kept a tracker of its
// own, so a smaller grouped launch after a larger single one would have lowered the attribute under the other's feet).
static bool raise_lds_limit(int lds) {
    static int attr_lds = 0;
    if (lds <= attr_lds) return true;
    if (hipFuncSetAttribute((const void*)wgl::wgrad3x3_c32, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return false;
    attr_lds = lds;
    return true;
}


// -1 = not eligible (caller falls back).  part == null: fp32 atomics into dw; else the workgroups store their partial
// tiles and the CALLER sums ``*nslices`` slices in slab order.
int advmix_wgrad_lds_dispatch(const float* a, const float* b, float* dw, int N, int Ha, int Wa, int Ca, int Hb, int Wb,
                              int Cb, int R, int S, int stride, int pad, float* part, int64_t part_floats, int* nslices,
                              hipStream_t st) {
    const int mode = advmix_opts().wgrad_lds;
    if (!mode || Ca != 32 || Cb != 32 || R != 3 || S != 3 || stride != 1 || pad != 1 || Ha != Hb || Wa != Wb) return -1;
    const int64_t bytes = (int64_t)N * Ha * Wa * 32 * 4;
    if (bytes >= 0x7fffffffLL) return -1;
    const int Wp = (Wa + 1) & ~1;
    int rows = 0;
    for (int r : {8, 4}) {                              // one workgroup per CU if the batch has that many slabs
        if (Ha % r) continue;
        const int64_t lds = ((int64_t)r * Wp + (int64_t)(r + 2) * (Wp + 2)) * 32 * 4;
        if (lds > 150 * 1024) continue;
        if ((int64_t)N * (Ha / r) >= 192 || mode == 2) { rows = r; break; }
    }
    if (!rows) return -1;
    const int slabs = N * (Ha / rows);
    if (part && (int64_t)slabs * 32 * 9 * 32 > part_floats) return -1;
    int lds = (rows * Wp + (rows + 2) * (Wp + 2)) * 32 * 4;
    if (lds < 9 * wgl::TILE * 4) lds = 9 * wgl::TILE * 4;
    if (!raise_lds_limit(lds)) return -1;
    wgl::LP p{a, b, dw, part, N, Ha, Wa, Wp, rows, (int)bytes, 1, slabs, 1, {a}, {b}, {dw}};
    dim3 g(slabs);
    hipLaunchKernelGGL(wgl::wgrad3x3_c32, g, dim3(wgl::THREADS), lds, st, p);
    if (advmix_opts().trace_shapes)
        advmix_trace_launch("wgrad3x3_c32", g, "wgrad", N, Hb, Wb, Cb, Ha, Wa, Ca, R, S, stride,
                            2.0 * N * (double)Ha * Wa * Ca * Cb * R * S);
    if (nslices) *nslices = part ? slabs : 0;
    return hipGetLastError() == hipSuccess ? ADVMIX_OK : ADVMIX_ELAUNCH;
}

// n problems of one geometry in one launch, each workgroup walking ``nslab`` slabs before it merges (see wgl::LP).  -1 = not
// eligible (nothing launched).
int advmix_wgrad_lds_group_dispatch(int n, const float* const* a, const float* const* b, float* const* dw, int N, int Ha, int Wa,
                                    int Ca, int Hb, int Wb, int Cb, int R, int S, int stride, int pad, hipStream_t st) {
    const int mode = advmix_opts().wgrad_lds;
    if (!mode || n < 2 || n > 64 || Ca != 32 || Cb != 32 || R != 3 || S != 3 || stride != 1 || pad != 1 || Ha != Hb || Wa != Wb)
        return -1;
    const int64_t bytes = (int64_t)N * Ha * Wa * 32 * 4;
    if (bytes >= 0x7fffffffLL) return -1;
    const int Wp = (Wa + 1) & ~1;
    int rows = 0;
    for (int r : {8, 4}) {
        if (Ha % r) continue;
        const int64_t lds = ((int64_t)r * Wp + (int64_t)(r + 2) * (Wp + 2)) * 32 * 4;
        if (lds > 150 * 1024) continue;
        rows = r;
        break;
    }
    if (!rows) return -1;
    const int slabs = N * (Ha / rows);
    // as many workgroups as CUs over ALL the problems: the smallest divisor of the slab count that leaves <= 256 / n
    // workgroups per problem (B = 32 @64x48, eight problems: 8 slabs per workgroup, 32 workgroups each)
    static const int target = [] { const char* e = getenv("ADVMIX_WGRAD_GROUP_WGS"); int t = e ? atoi(e) : 256; return t > 0 ? t : 256; }();
    int nslab = 0;
    for (int d = 1; d <= slabs; ++d)
        if (slabs % d == 0 && (int64_t)n * (slabs / d) <= target) { nslab = d; break; }
    if (!nslab) return -1;
    int lds = (rows * Wp + (rows + 2) * (Wp + 2)) * 32 * 4;
    if (lds < 9 * wgl::TILE * 4) lds = 9 * wgl::TILE * 4;
    if (!raise_lds_limit(lds)) return -1;
    wgl::LP p{nullptr, nullptr, nullptr, nullptr, N, Ha, Wa, Wp, rows, (int)bytes, nslab, slabs / nslab, n, {}, {}, {}};
    for (int i = 0; i < 64; ++i) {
        p.dyv[i] = i < n ? a[i] : nullptr;
        p.xv[i] = i < n ? b[i] : nullptr;
        p.dwv[i] = i < n ? dw[i] : nullptr;
    }
    dim3 g(n * (slabs / nslab));
    hipLaunchKernelGGL(wgl::wgrad3x3_c32, g, dim3(wgl::THREADS), lds, st, p);
    if (advmix_opts().trace_shapes) {
        char kd[24];
        snprintf(kd, sizeof kd, "wgrad x%d", n);
        advmix_trace_launch("wgrad3x3_c32", g, kd, N, Hb, Wb, Cb, Ha, Wa, Ca, R, S, stride,
                            2.0 * n * N * (double)Ha * Wa * Ca * Cb * R * S);
    }
    return hipGetLastError() == hipSuccess ? ADVMIX_OK : ADVMIX_ELAUNCH;
}

// This translation unit's share of advmix_build_flags(): 0 in the shipped library (the measurement switches of this file live
// in tools/variants/wgrad_lds_dbg.patch, which makes this return 32).
int advmix_wgrad_lds_build_flags(void) { return 0; }
