// 1x1 convolution 64 -> 256 channels over many pixels - a streaming kernel (round 5).
//
// HRNet's / ResNet's first stage (pose_hrnet.py:59-98, Bottleneck) runs 1x1 convs between 64 and 256 channels on the
// 64 x 48 map: at B = 32 that is 98,304 pixels x 256 channels = 100 MB per tensor against 3.2 GFLOP - as much HBM time
// (21 us) as MFMA time (20.5 us).  On conv_direct these are 3,072 short workgroups (128 x 64 tiles, two K chunks) at
// 2.3-2.9 TB/s: 53 us forward + BatchNorm sums, 60 us eval, 81 us input gradient + BatchNorm backward.  Here, for the
// shape K = 64 -> N = 256 (the block's last conv and the shortcut forward; the input gradient of its first conv):
//   * a workgroup = 128 pixels x ALL 256 output channels, four waves of 32 pixels: 768 workgroups = one round of three per
//     CU; the 128 x 64 input tile is loaded ONCE with 16-byte coalesced loads, staged through LDS into each lane's MFMA A
//     fragments (whole K = 32 registers), and never touched again;
//   * filters come pre-laid in MFMA B-fragment order (pw_weights, beside the Winograd images): 64 KB per conv, the same for
//     every wave - L2 / L1 resident - a pass ahead in registers, never staged in LDS; no workgroup barrier after the first two;
//   * the 256 channels are produced in eight passes of 32 (one 32 x 32 accumulator at a time - 16 registers instead of 128,
//     three waves per SIMD): after each pass the wave writes its 32 x 32 result into a PRIVATE LDS image and runs the fused
//     epilogue in the natural layout - lane = (pixel row, 4 channels), 128 contiguous bytes per pixel and instruction, 16-byte
//     operand loads and stores (conv_smap.hip's epilogue arithmetic) - while the other waves of the SIMD multiply.
#include "common.h"
#include <stdio.h>

namespace pw {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;      // >= any buffer size accepted -> loads return 0, stores are dropped
constexpr int STORE_AUX = 16;              // sc1 (write-through), as conv_direct's epilogue
constexpr int K = 64, N = 256, BM = 128, NT = 256;
constexpr int AP = K + 4;                  // pitch of a staged pixel (floats): 16 lanes of a ds_read_b128 fall on 16 different slots
constexpr int IP = 32 + 4;                 // pitch of a row of a wave's epilogue image (32 channels per pass)
static_assert(4 * 32 * IP <= BM * AP, "the four waves' epilogue images fit the input tile's region");

struct PP {
    const float* x;
    const float* u;           // filters in fragment order (see pw_weights)
    float* y;
    int M;                    // pixels (B * H * W)
    int xbytes, ybytes, ubytes;
    // role 0 (forward): column sums of the raw output and / or eval-mode BatchNorm, residual, activation
    const float *bn_gamma, *bn_beta, *bn_rm, *bn_rv, *res;
    float bn_eps;
    int act;
    double* stats;            // [2][stats_nbg][N] fp64 slots (slot-major), zero on entry
    int stats_nbg;
    // role 1 (input gradient): ``res`` is the addend; with bnb_c the epilogue is the BatchNorm-backward one
    const unsigned char* bnb_mask;
    const float *bnb_c, *bnb_mean, *bnb_invstd, *bnb_gamma, *bnb_beta;
    int bnb_act;
};

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ int acc_row(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }   // v_mfma_f32_32x32x2 D layout

// VAR: which epilogue is compiled in (the SIMD issues VALU and MFMA instructions one after the other - a launch takes the SUM
// of the two, not the larger - so every instruction of an epilogue the launch does not use is paid for: the first version
// decided everything at run time, ~350 VALU instructions per pass and wave, 43 us = 25 us of MFMA section + 21 us of the rest):
//   0  forward + BatchNorm column sums of the raw output (taken in the accumulator layout: 34 instructions per pass)
//   1  forward + eval-mode BatchNorm (+ residual) + activation
//   2  input gradient (+ addend) + BatchNorm-backward epilogue (activation slope from the bit mask or from c, sums of g, g xhat)
//   3  plain: forward or input gradient (+ residual / addend), nothing else
template <int VAR>
__global__ __launch_bounds__(NT, 3) void conv_pw(const PP p) {
    __shared__ __attribute__((aligned(16))) float L[BM * AP];
    __shared__ float sred[2 * 4 * N];
    __shared__ __attribute__((aligned(16))) float prm[(VAR == 1 || VAR == 2) ? 4 * N : 4];   // per-channel epilogue parameters

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int m0 = (int)blockIdx.x * BM;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.u, 0, p.ubytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.y), 0, p.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bnb_c ? p.bnb_c : p.y), 0, p.ybytes, 0x00020000);
    // filter fragment 8 nt + g (column tile nt of 32, 8-channel group g): 1 KB each, the same 64 for every wave.  A pass's
    // eight are requested right after the previous pass's MFMAs - BEFORE that pass's epilogue stores (gfx950 counts loads
    // and stores in ONE in-order counter: a wait for a load issued after a store also waits for the store's round trip).
    const unsigned bo = (unsigned)(lane * 16);
    f32x4 bq[8];
    auto issue_b = [&](int pass) {
#pragma unroll
        for (int g = 0; g < 8; ++g) bq[g] = bload(ur, bo + (unsigned)((8 * pass + g) * 1024));     // (pass 8: out of range -> zeros)
    };
    issue_b(0);

    // ---- the 128 x 64 input tile -> LDS with coalesced 16-byte loads (a pixel past the end: out-of-range offset, zeros) ----
    {
        f32x4 stg[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int s = tid + NT * it;
            const int row = s >> 4, cs = s & 15;
            stg[it] = bload(xr, m0 + row < p.M ? (unsigned)(((m0 + row) * K + cs * 4) * 4) : OOB);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int s = tid + NT * it;
            *reinterpret_cast<f32x4*>(&L[(s >> 4) * AP + (s & 15) * 4]) = stg[it];
        }
    }
    if constexpr (VAR == 1) {                               // (1 / sqrt(var + eps), gamma, beta, running mean) of channel tid
        prm[tid] = 1.0f / sqrtf(p.bn_rv[tid] + p.bn_eps);
        prm[N + tid] = p.bn_gamma[tid]; prm[2 * N + tid] = p.bn_beta[tid]; prm[3 * N + tid] = p.bn_rm[tid];
    }
    if constexpr (VAR == 2) {                               // (mean, 1 / std, gamma, beta) of the producer's BatchNorm
        prm[tid] = p.bnb_mean[tid]; prm[N + tid] = p.bnb_invstd[tid];
        prm[2 * N + tid] = p.bnb_gamma ? p.bnb_gamma[tid] : 0.f; prm[3 * N + tid] = p.bnb_beta ? p.bnb_beta[tid] : 0.f;
    }
    __syncthreads();
    // this lane's A fragments for the WHOLE K: pixel 32 wv + l31, channels 32 lh + 4 g + j (MFMA j of group g)
    f32x4 a[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) a[g] = *reinterpret_cast<const f32x4*>(&L[(32 * wv + l31) * AP + 32 * lh + 4 * g]);
    __syncthreads();                                        // every wave holds its pixels: the region becomes the waves' private images
    float* const Ti = L + wv * (32 * IP);
    auto wave_fence = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };

    const bool mask_on = VAR == 2 && p.bnb_mask != nullptr && p.bnb_act != ADVMIX_ACT_NONE;
    const bool recompute = VAR == 2 && !mask_on && p.bnb_act != ADVMIX_ACT_NONE;
    const bool has_res = p.res != nullptr;
    const float bb_slope = act_neg_slope(p.bnb_act);
    const int quad = lane & 7, rsub = lane >> 3;            // epilogue item i of a pass: pixel row rsub + 8 i of the wave, channels 4 quad ..
    // the four items' byte offsets for column tile 0 (a pass adds 128 h); a pixel past the end: out of range - loads give 0, stores are dropped
    unsigned yo0[4];
    bool live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 32 * wv + rsub + 8 * i;
        live[i] = m < p.M;
        yo0[i] = live[i] ? (unsigned)((m * N + 4 * quad) * 4) : OOB;
    }
    const float* const tr = &Ti[rsub * IP + 4 * quad];      // this lane's items in the image: + 8 i IP
    float* const tw = &Ti[(4 * lh) * IP + l31];             // ... and its accumulator rows: acc_row(r, lh) = (r & 3) + 8 (r >> 2) + 4 lh

#pragma unroll 1
    for (int h = 0; h < 8; ++h) {                           // 32 output channels per pass: column tile h
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][j], bq[g][j], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        issue_b(h + 1);                                     // the next pass's filters: in flight under this pass's epilogue, older than its stores
        __builtin_amdgcn_sched_barrier(0);
        const int col = 32 * h + 4 * quad;
        if constexpr (VAR == 0) {
            // BatchNorm column sums of the raw output where a lane holds ONE column: 16 rows here, 16 in lane + 32
            float s1 = 0.f, s2 = 0.f;
            const int mr = m0 + 32 * wv + 4 * lh;           // row of register r: mr + (r & 3) + 8 (r >> 2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float o = mr + (r & 3) + 8 * (r >> 2) < p.M ? acc[r] : 0.f;
                s1 += o;
                s2 = __builtin_fmaf(o, o, s2);
            }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lh == 0) {
                sred[wv * N + 32 * h + l31] = s1;
                sred[(4 + wv) * N + 32 * h + l31] = s2;
            }
        }
        // ---- the pass's 32 x 32 result -> the wave's image, then the epilogue in the natural layout --------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) tw[((r & 3) + 8 * (r >> 2)) * IP] = acc[r];
        wave_fence();
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        auto ld4 = [&](int k) { return *reinterpret_cast<const f32x4*>(&prm[k * N + col]); };
        f32x4 bn_is = z4, bn_g = z4, bn_b = z4, bn_m = z4, bb_mu = z4, bb_is = z4, bb_g = z4, bb_b = z4;
        if constexpr (VAR == 1) { bn_is = ld4(0); bn_g = ld4(1); bn_b = ld4(2); bn_m = ld4(3); }
        if constexpr (VAR == 2) {
            bb_mu = ld4(0); bb_is = ld4(1);
            if (recompute) { bb_g = ld4(2); bb_b = ld4(3); }
        }
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        constexpr int GI = VAR == 2 ? 2 : 4;                // items whose operands are in flight together (variant 2: three operands each)
#pragma unroll
        for (int grp = 0; grp < 4 / GI; ++grp) {
            f32x4 oa[GI], oc[GI];
            unsigned mb[GI];
#pragma unroll
            for (int i = 0; i < GI; ++i) {
                const int ii = GI * grp + i;
                const unsigned yo = yo0[ii] + (unsigned)(128 * h);
                oa[i] = z4; oc[i] = z4; mb[i] = 0u;
                if (VAR != 0 && has_res) oa[i] = bload(rr, yo);
                if constexpr (VAR == 2) {
                    oc[i] = bload(cr, yo);
                    if (mask_on && live[ii]) mb[i] = p.bnb_mask[yo >> 4];       // byte = (pixel * 256 + column) / 4; bit e: channel col + e
                }
            }
#pragma unroll
            for (int i = 0; i < GI; ++i) {
                const int ii = GI * grp + i;
                f32x4 v = *reinterpret_cast<const f32x4*>(tr + 8 * ii * IP);
                if constexpr (VAR != 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float o = v[e];
                        if constexpr (VAR == 1) {
                            o = (o - bn_m[e]) * bn_is[e] * bn_g[e] + bn_b[e];
                            o += oa[i][e];
                            o = act_fwd(o, p.act);
                        } else {
                            o += oa[i][e];
                            if constexpr (VAR == 2) {
                                const float xh = (oc[i][e] - bb_mu[e]) * bb_is[e];
                                if (mask_on) o = ((mb[i] >> e) & 1u) ? o : o * bb_slope;
                                else if (recompute) o = __builtin_fmaf(xh, bb_g[e], bb_b[e]) > 0.f ? o : o * bb_slope;
                                if (live[ii]) { s1[e] += o; s2[e] = __builtin_fmaf(o, xh, s2[e]); }
                            }
                        }
                        v[e] = o;
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, yo0[ii] + (unsigned)(128 * h), 0, STORE_AUX);
            }
        }
        if constexpr (VAR == 2) {
            // a wave's instruction = 8 pixel rows x 8 channel quads (lane = 8 rsub + quad): the rows add up by shuffles
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int d = 8; d < 64; d <<= 1) {
                    s1[e] += __shfl_xor(s1[e], d, 64);
                    s2[e] += __shfl_xor(s2[e], d, 64);
                }
            }
            if (lane < 8) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sred[wv * N + col + e] = s1[e];
                    sred[(4 + wv) * N + col + e] = s2[e];
                }
            }
        }
        wave_fence();                                       // the image is free for the next pass
    }
    if constexpr (VAR == 0 || VAR == 2) {
        __syncthreads();
        double d1 = 0.0, d2 = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d1 += (double)sred[k * N + tid];
            d2 += (double)sred[(4 + k) * N + tid];
        }
        const int sl = (int)blockIdx.x % p.stats_nbg;      // slot-major [2][slots][N]: consecutive doubles per workgroup
        atomicAdd(p.stats + (int64_t)sl * N + tid, d1);
        atomicAdd(p.stats + ((int64_t)p.stats_nbg + sl) * N + tid, d2);
    }
}

// ---- filter re-layout ---------------------------------------------------------------------------------------------------
// One block of 256 threads = one 1 KB fragment (column tile nt of 32, 8-channel group g) at index 8 nt + g; thread t = 4 lane + j
// holds B[k-lane lane / 32, MFMA j][column lane % 32] = filter(n = 32 nt + lane % 32, k = 32 (lane / 32) + 4 g + j).
// role 0: w[n][k] (forward of a 64 -> 256 conv: n = Cout, k = Cin); role 1: w[k][n] (input gradient of a 256 -> 64 conv:
// n = Cin, k = Cout).  Same record as wino::WinoEnt.
struct PEnt {
    const float* w;
    float* u;
    int Cn, Ck, role, blk0;
};

__global__ __launch_bounds__(256) void pw_weights(const PEnt* __restrict__ ents, const int* __restrict__ blk_ent) {
    const PEnt e = ents[blk_ent[blockIdx.x]];
    const int lb = (int)blockIdx.x - e.blk0;
    const int g = lb & 7, nt = lb >> 3;
    const int t = threadIdx.x, lane = t >> 2, j = t & 3;
    const int n = 32 * nt + (lane & 31), k = 32 * (lane >> 5) + 4 * g + j;
    e.u[(int64_t)lb * 256 + t] = e.role == 0 ? e.w[(int64_t)n * e.Ck + k] : e.w[(int64_t)k * e.Cn + n];
}

}  // namespace pw

// Which problems the kernel serves: a 1x1 / stride 1 conv reading 64 channels and writing 256.
static bool pw_shape_ok(int N, int H, int W, int Ci, int Co) {
    if (N <= 0 || H < 1 || W < 1 || Ci != pw::K || Co != pw::N) return false;
    if ((int64_t)N * H * W * pw::N * 4 >= 0x7fffffffLL) return false;
    return true;
}

// 0: not served; otherwise the number of workgroups of the launch (128 pixels each)
extern "C" int advmix_conv_pw_config(int N, int H, int W, int Ci, int Co) {
    if (!pw_shape_ok(N, H, W, Ci, Co)) return 0;
    return (int)cdiv((int64_t)N * H * W, pw::BM);
}

// floats of one re-laid image of a 1x1 filter bank between 64 and 256 channels
extern "C" int64_t advmix_pw_u_floats(int Co, int Ci) { return (int64_t)Co * Ci; }

// Re-lay the filters of several convs in one launch (records as advmix_wino_weights'; a record owns 64 workgroups;
// role 0 needs w[256][64], role 1 w[64][256]).
extern "C" int advmix_pw_weights(const void* ents, const int* blk_ent, int blocks, void* stream) {
    if (!ents || !blk_ent || blocks <= 0) return ADVMIX_EINVAL;
    hipLaunchKernelGGL(pw::pw_weights, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const pw::PEnt*)ents, blk_ent);
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

static int pw_fill(pw::PP& p, const float* x, const float* u, float* y, int N, int H, int W, int Ci, int Co) {
    if (!x || !u || !y || !pw_shape_ok(N, H, W, Ci, Co)) return ADVMIX_EINVAL;
    p = pw::PP{};
    p.x = x; p.u = u; p.y = y;
    p.M = N * H * W;
    p.xbytes = (int)((int64_t)p.M * pw::K * 4);
    p.ybytes = (int)((int64_t)p.M * pw::N * 4);
    p.ubytes = pw::K * pw::N * 4;
    return ADVMIX_OK;
}

static int pw_slots(const int* stats_ns) {
    int ns = stats_ns && *stats_ns > 0 ? *stats_ns : advmix_opts().stat_slots;
    if (ns <= 0 || ns > ADVMIX_STAT_SLOTS_MAX || (ns & (ns - 1))) ns = 16;
    return ns;
}

static int pw_launch(int role, const pw::PP& p, int N, int H, int W, hipStream_t st) {
    const dim3 g((unsigned)cdiv(p.M, pw::BM));
    // the epilogue variant: forward + sums / forward + eval BatchNorm / input gradient + BatchNorm backward / plain
    if (role == 0 && p.stats && !p.bn_gamma && !p.res && p.act == ADVMIX_ACT_NONE) hipLaunchKernelGGL(pw::conv_pw<0>, g, dim3(pw::NT), 0, st, p);
    else if (role == 0 && p.bn_gamma && !p.stats) hipLaunchKernelGGL(pw::conv_pw<1>, g, dim3(pw::NT), 0, st, p);
    else if (role == 1 && p.bnb_c) hipLaunchKernelGGL(pw::conv_pw<2>, g, dim3(pw::NT), 0, st, p);
    else if (!p.stats && !p.bn_gamma && (role == 1 || p.act == ADVMIX_ACT_NONE)) hipLaunchKernelGGL(pw::conv_pw<3>, g, dim3(pw::NT), 0, st, p);
    else return ADVMIX_EINVAL;                              // (e.g. sums AND an eval epilogue in one launch: not a combination the step uses)
    if (advmix_opts().trace_shapes) {
        char nm[32];
        snprintf(nm, sizeof nm, "conv_pw<%d>", role == 0 ? (p.stats ? 0 : (p.bn_gamma ? 1 : 3)) : (p.bnb_c ? 2 : 3));
        advmix_trace_launch(nm, g, role == 0 ? (p.stats ? "fwd+sums" : (p.bn_gamma ? "fwd+bn_eval" : "fwd")) : (p.bnb_c ? "dgrad+bnb" : "dgrad"),
                            N, H, W, pw::K, H, W, pw::N, 1, 1, 1, 2.0 * p.M * (double)pw::K * pw::N);
    }
    ADVMIX_CHECK_LAUNCH();
    return ADVMIX_OK;
}

// advmix_conv_fwd_ex (no bias) for a 1x1 / stride 1 conv 64 -> 256 from the role 0 image ``u`` of advmix_pw_weights: the
// arguments and epilogues of advmix_conv3x3_wino_fwd.  ADVMIX_EINVAL (nothing launched) when advmix_conv_pw_config is 0.
// Semantics: lib/models/pose_hrnet.py:59-98 (conv1x1 + BatchNorm2d (+ residual) + ReLU of a Bottleneck).
extern "C" int advmix_conv1x1_pw_fwd(const float* x, const float* u, float* y, int N, int H, int W, int Ci, int Co,
                                     const float* bn_gamma, const float* bn_beta, const float* bn_rm, const float* bn_rv,
                                     float bn_eps, const float* residual, int act, double* stats, int* stats_ns, void* stream) {
    if ((bn_gamma != nullptr) != (bn_beta && bn_rm && bn_rv)) return ADVMIX_EINVAL;
    if (stats && !stats_ns) return ADVMIX_EINVAL;
    if (advmix_opts().deterministic && stats) return ADVMIX_EINVAL;        // fp64 atomics: the ordered form is conv_direct's
    pw::PP p;
    int rc = pw_fill(p, x, u, y, N, H, W, Ci, Co);
    if (rc) return rc;
    p.bn_gamma = bn_gamma; p.bn_beta = bn_beta; p.bn_rm = bn_rm; p.bn_rv = bn_rv; p.bn_eps = bn_eps;
    p.res = residual; p.act = act; p.stats = stats;
    p.stats_nbg = pw_slots(stats_ns);
    rc = pw_launch(0, p, N, H, W, (hipStream_t)stream);
    if (rc == ADVMIX_OK && stats_ns) *stats_ns = p.stats_nbg;
    return rc;
}

// advmix_conv_tr_w_add / advmix_conv_tr_w_bnb for a 1x1 / stride 1 conv 256 -> 64: dx[N,H,W,256] from dy[N,H,W,64] and the
// role 1 image ``u``; the arguments and epilogues of advmix_conv3x3_wino_dgrad (Co = 64 channels read, Ci = 256 written).
extern "C" int advmix_conv1x1_pw_dgrad(const float* dy, const float* u, const float* addend, float* dx, int N, int H, int W,
                                       int Co, int Ci, const unsigned char* act_mask, const float* bn_c, const float* bn_mean,
                                       const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int act,
                                       double* stats, int* stats_ns, void* stream) {
    if (bn_c) {
        if (!bn_mean || !bn_invstd || !stats || !stats_ns) return ADVMIX_EINVAL;
        if (act != ADVMIX_ACT_NONE && !act_mask && !(bn_gamma && bn_beta)) return ADVMIX_EINVAL;
        if (advmix_opts().deterministic) return ADVMIX_EINVAL;
    } else if (stats) {
        return ADVMIX_EINVAL;
    }
    pw::PP p;
    int rc = pw_fill(p, dy, u, dx, N, H, W, Co, Ci);        // the gradient conv reads Co = 64 channels and writes Ci = 256
    if (rc) return rc;
    p.res = addend;
    if (bn_c) {
        p.stats = stats; p.stats_nbg = pw_slots(stats_ns);
        p.bnb_mask = act_mask; p.bnb_c = bn_c; p.bnb_mean = bn_mean; p.bnb_invstd = bn_invstd;
        p.bnb_gamma = bn_gamma; p.bnb_beta = bn_beta; p.bnb_act = act;
    }
    rc = pw_launch(1, p, N, H, W, (hipStream_t)stream);
    if (rc == ADVMIX_OK && bn_c) *stats_ns = p.stats_nbg;
    return rc;
}
