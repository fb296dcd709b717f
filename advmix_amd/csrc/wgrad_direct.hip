// Weight gradient with both MFMA operands loaded straight from HBM/L2 in fragment layout.
//
// dW[co][tap][ci] = sum_p dY[p][co] * X[p (+) tap][ci]: pixels are the reduction axis and BOTH
// operands are channel-contiguous per pixel, which is exactly the fp32 32x32x2 fragment shape:
// lane (l&31, l>>5) of the A operand needs dY[p0 + (l>>5)][co0 + (l&31)] - one dword per lane, and a
// wave's 64 dwords are two fully coalesced 128-byte pixel rows.  So no LDS staging, no transposes
// and no barriers in the main loop (the first-generation conv_wgrad stages both operands through
// LDS with two barriers per 16 pixels and reads them back with one ds_read_b32 per MFMA operand).
// Out-of-image taps get an out-of-range buffer offset -> the hardware returns 0.
//
// A workgroup = 4 waves on the SAME (co, tap*ci) tile, each summing a quarter of the workgroup's
// pixel slice; the four partial tiles are reduced through LDS and added to dW with ONE set of fp32
// atomics (the atomic pipe runs at ~1.3 TB/s chip-wide, so partial sums must be merged on chip).
#include "common.h"
#include <stdio.h>

namespace wgd {

struct WP {
    const float* a;   // [P, Ca]   (conv: dY)
    const float* b;   // [N, Hb, Wb, Cb] (conv: X)
    float* dw;        // [Ca][R*S][Cb]
    int N, Ha, Wa, Ca, Hb, Wb, Cb;
    int R, S, stride, pad;
    int chunk;        // pixels per workgroup (multiple of 8)
    int abytes, bbytes;
};

constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

template <int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_direct(WP p) {
    extern __shared__ __attribute__((aligned(16))) float red[];        // [3][TM*TN*1024]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int jtiles = (p.R * p.S * p.Cb) / (32 * TN);
    const int tile = blockIdx.x;
    const int co0 = (tile / jtiles) * 32 * TM;
    const int j0 = (tile % jtiles) * 32 * TN;
    const int P = p.N * p.Ha * p.Wa;
    const int HWa = p.Ha * p.Wa;

    // this wave's pixel range: a quarter of the workgroup's slice, in units of 2 pixels
    const int blo = blockIdx.y * p.chunk;
    const int per = p.chunk / 4;
    const int plo = blo + wid * per;
    const int phi = min(P, plo + per);

    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, p.bbytes, 0x00020000);

    // per-tile constants
    unsigned a_col[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) a_col[t] = (co0 + 32 * t + l31 < p.Ca) ? (unsigned)((co0 + 32 * t + l31) * 4) : OOB;
    int b_dh[TN], b_dw[TN];
    unsigned b_col[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        int j = j0 + 32 * u;
        int tap = j / p.Cb, ci0 = j - tap * p.Cb;
        int r = tap / p.S;
        b_dh[u] = r - p.pad;
        b_dw[u] = tap - r * p.S - p.pad;
        b_col[u] = (unsigned)((ci0 + l31) * 4);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    if (plo < phi) {
        // this lane's pixel (plo + lh), advanced by 2 per MFMA k step
        int pix = plo + lh;
        int n = pix / HWa;
        int rem = pix - n * HWa;
        int ha = rem / p.Wa, wa = rem - ha * p.Wa;
        for (int p0 = plo; p0 < phi; p0 += 8) {
            float av[4][TM], bv[4][TN];
#pragma unroll
            for (int s = 0; s < 4; ++s) {                  // four k steps (8 pixels) of loads in flight
                const bool in = pix < phi;
                const unsigned arow = in ? (unsigned)pix * (unsigned)(p.Ca * 4) : OOB;
#pragma unroll
                for (int t = 0; t < TM; ++t)
                    av[s][t] = bload1(ar, (arow == OOB || a_col[t] == OOB) ? OOB : arow + a_col[t]);
#pragma unroll
                for (int u = 0; u < TN; ++u) {
                    int hb = ha * p.stride + b_dh[u], wb = wa * p.stride + b_dw[u];
                    bool ok = in && (unsigned)hb < (unsigned)p.Hb && (unsigned)wb < (unsigned)p.Wb;
                    bv[s][u] = bload1(br, ok ? (unsigned)(((n * p.Hb + hb) * p.Wb + wb) * p.Cb) * 4u + b_col[u] : OOB);
                }
                pix += 2;
                wa += 2;
                while (wa >= p.Wa) { wa -= p.Wa; if (++ha == p.Ha) { ha = 0; ++n; } }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < TN; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][t], bv[s][u], acc[t][u], 0, 0, 0);
        }
    }

    // ---- merge the four waves' partial tiles in LDS, then one set of atomics --------------------
    constexpr int TILE = TM * TN * 1024;
    if (wid > 0) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(wid - 1) * TILE + ((t * TN + u) * 16 + r) * 64 + lane] = acc[t][u][r];
    }
    __syncthreads();
    if (wid == 0) {
        const int Ntot = p.R * p.S * p.Cb;
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u) {
                const int j = j0 + 32 * u + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int idx = ((t * TN + u) * 16 + r) * 64 + lane;
                    float v = acc[t][u][r] + red[idx] + red[TILE + idx] + red[2 * TILE + idx];
                    int co = co0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (co < p.Ca) atomicAdd(p.dw + (int64_t)co * Ntot + j, v);
                }
            }
    }
}

template <int TM, int TN>
int launch(WP& p, hipStream_t st) {
    static bool attr_done = false;
    const int lds = 3 * TM * TN * 1024 * (int)sizeof(float);
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)wgrad_direct<TM, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess)
            return -1;
        attr_done = true;
    }
    const int64_t P = (int64_t)p.N * p.Ha * p.Wa;
    const int tiles = cdiv(p.Ca, 32 * TM) * ((p.R * p.S * p.Cb) / (32 * TN));
    int64_t ns = 768 / tiles;                               // ~3 workgroups per CU
    if (ns < 1) ns = 1;
    int64_t maxs = (P + advmix_wgrad_min_pix() - 1) / advmix_wgrad_min_pix();   // (>= 16 pixels = 2 load batches per wave)
    if (ns > maxs) ns = maxs;
    int64_t chunk = ((P + ns - 1) / ns + 31) / 32 * 32;     // multiple of 32: a multiple of 8 per wave
    p.chunk = (int)chunk;
    dim3 g(tiles, cdiv(P, chunk));
    hipLaunchKernelGGL((wgrad_direct<TM, TN>), g, dim3(256), lds, st, p);
    if (advmix_opts().trace_shapes) {
        char nm[64];
        snprintf(nm, sizeof nm, "wgrad_direct<%d, %d>", TM, TN);
        advmix_trace_launch(nm, g, "wgrad", p.N, p.Hb, p.Wb, p.Cb, p.Ha, p.Wa, p.Ca, p.R, p.S, p.stride,
                            2.0 * p.N * (double)p.Ha * p.Wa * p.Ca * p.Cb * p.R * p.S);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADVMIX_OK : ADVMIX_ELAUNCH;
}

}  // namespace wgd

// returns -1 when not eligible (caller falls back to conv_wgrad)
int advmix_wgrad_direct_dispatch(const float* a, const float* b, float* dw, int N, int Ha, int Wa, int Ca, int Hb,
                                 int Wb, int Cb, int R, int S, int stride, int pad, hipStream_t st) {
    // Measured (B=32): wins where the output-channel tile is a single 32 (40.8 vs 46.2 us on 3x3 32->32,
    // 19.4 vs 24.7 us on 3x3/s2 32->64 seen from its 32-wide side), loses ~10 % on wider tiles where the
    // LDS-staged kernel re-uses each staged operand across a 64x64 workgroup tile.  wgrad_direct = 2 forces it.
    const int mode = advmix_opts().wgrad_direct;
    if (!mode || Cb % 32 != 0 || (mode == 1 && Ca > 32)) return -1;
    const int64_t ab = (int64_t)N * Ha * Wa * Ca * 4, bb = (int64_t)N * Hb * Wb * Cb * 4;
    if (ab >= 0x7fffffffLL || bb >= 0x7fffffffLL) return -1;
    wgd::WP p{a, b, dw, N, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad, 0, (int)ab, (int)bb};
    const int jt32 = (R * S * Cb) / 32;
    if (Ca <= 32) {
        if (jt32 % 3 == 0) return wgd::launch<1, 3>(p, st);
        if (jt32 % 2 == 0) return wgd::launch<1, 2>(p, st);
        return wgd::launch<1, 1>(p, st);
    }
    if (jt32 % 2 == 0) return wgd::launch<2, 2>(p, st);
    return wgd::launch<2, 1>(p, st);
}
