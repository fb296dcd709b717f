/* advmix_hip.h — C ABI of libadvmix_hip.so: the MI355X (gfx950) kernels behind the
 * AdvMix training step and the lib/nms post-process.
 *
 * Conventions
 *   - every entry point returns int: 0 ok, ADVMIX_EINVAL bad argument/unsupported
 *     shape, ADVMIX_ELAUNCH HIP launch/runtime error.  Nothing throws.
 *   - all tensor pointers are DEVICE pointers unless the name ends in _host.
 *   - activations are dense NHWC fp32 ("channels-last"); a row is one pixel.
 *   - no hidden allocation, no implicit synchronisation: work is enqueued on
 *     `stream` (a hipStream_t passed as void*) and returns at once, so every call
 *     is legal inside hipStreamBeginCapture/EndCapture (HIP graphs).
 *     The *_host NMS entry points are the exception (they mirror the reference's
 *     synchronous `_nms`).
 *   - weights: conv  [Cout][R][S][Cin]  (= torch OIHW tensor in channels_last memory)
 *              deconv[Cin][R][S][Cout]  (= torch IOHW tensor in channels_last memory)
 *
 * The reference has no op-level native ABI of its own (torch -> cuDNN/ATen is its
 * "FFI"); each group below cites the reference call site(s) it replaces.  The one
 * native symbol the reference does export, `_nms` (lib/nms/gpu_nms.hpp:1-2), is
 * reproduced as advmix_nms_host with the same argument list.
 */
#ifndef ADVMIX_HIP_H
#define ADVMIX_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ADVMIX_OK 0
#define ADVMIX_EINVAL 1
#define ADVMIX_ELAUNCH 2

#define ADVMIX_ACT_NONE 0
#define ADVMIX_ACT_RELU 1
#define ADVMIX_ACT_LEAKY02 2   /* LeakyReLU(0.2), lib/models/Unet_generator.py:42 */

int advmix_version(void);
/* 0 for the shipped library; non-zero bits name the measurement switches a variant build was compiled with
 * (tools/build_variant.sh, tools/variants/): such a library must never be benchmarked or shipped as the product. */
int advmix_build_flags(void);
/* dispatch knobs for A/B runs and tests: "direct" (0 = first-generation conv only), "wgrad_direct", "wgrad_lds"
 * (0 off, 1 when the batch fills the chip, 2 whenever eligible), "ksplit_wg" (K split inside the workgroup vs across the grid), "stat_slots" (fp64 slots per channel of the statistics
 * epilogues), "deterministic" (1: no K split across the grid), "trace_shapes" (measurement aid, see common.h).
 * Unknown name -> ADVMIX_EINVAL. */
int advmix_set_option(const char* name, int value);

/* ---- convolution family: replaces nn.Conv2d / nn.ConvTranspose2d forward+backward
 * (lib/models/pose_hrnet.py:22-25,65-71,200-232,323-349,411-417;
 *  lib/models/pose_resnet.py:22-27,111-112,140-141,179-186;
 *  lib/models/Unet_generator.py:40-41,48-50,58-60,66-68).  fp32 MFMA implicit GEMM. */

/* y[N,Ho,Wo,Co] = conv(x[N,Hi,Wi,Ci], w[Co][R][S][Ci]) (+bias[Co]); Ho = (Hi+2p-R)/s+1.
 * Also the input-gradient of a ConvTranspose2d. */
int advmix_conv_fwd(const float* x, const float* w, const float* bias, float* y,
                    int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                    int R, int S, int stride, int pad, void* stream);

/* Conv2d forward with a fused epilogue.  Eval-mode BatchNorm (bn_* all non-NULL or all NULL): y =
 * act((conv + bias - rm) / sqrt(rv + eps) * gamma + beta + residual).  stats != NULL: additionally writes
 * column sums of the RAW conv output, ACCUMULATED with fp64 atomics into stats[2][*stats_nbg][Co] (slot-major; the buffer
 * must be zero on entry; advmix_norm_finalize consumes it and leaves it zero), so a following train-mode
 * BatchNorm needs no statistics pass.  Returns
 * ADVMIX_EINVAL when the shape is served by a kernel without the fused epilogue (Cin % 16 != 0, K-split
 * small-M configurations): call advmix_conv_fwd and the separate norm kernels instead.
 * (pose_hrnet.py:41-57: conv -> bn -> (+residual) -> relu.) */
int advmix_conv_fwd_ex(const float* x, const float* w, const float* bias, float* y,
                       int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                       int R, int S, int stride, int pad,
                       const float* bn_gamma, const float* bn_beta, const float* bn_rm, const float* bn_rv,
                       float bn_eps, const float* residual, int act, double* stats, int* stats_nbg, void* stream);

/* "transposed gather": y[N,Hb,Wb,Cn] = sum_{r,s,c} x[N,(h+p-r)/s,(w+p-s)/s,c] * wt[Cn][R][S][Ck]
 * for the taps where the division is exact and in range (phase-decomposed: no zero work).
 * Conv2d input-gradient (x = dY, wt = transposed weight [Ci][R][S][Co]) and
 * ConvTranspose2d forward (wt = [Cout][R][S][Cin], + bias). x is [N,Hs,Ws,Ck]. */
int advmix_conv_tr(const float* x, const float* wt, const float* bias, float* y,
                   int N, int Hs, int Ws, int Ck, int Hb, int Wb, int Cn,
                   int R, int S, int stride, int pad, void* stream);

/* Same operation with the weight in its OWN layout, w[Ck][R][S][Cn] (a Conv2d's [Co][R][S][Ci] for its input
 * gradient, a ConvTranspose2d's [Ci][R][S][Co] for its forward): no advmix_transpose_w needed.  Supported
 * when Ck % 16 == 0 and Cn % 4 == 0 (returns ADVMIX_EINVAL otherwise: re-layout and call advmix_conv_tr). */
int advmix_conv_tr_w(const float* x, const float* w, const float* bias, float* y,
                     int N, int Hs, int Ws, int Ck, int Hb, int Wb, int Cn,
                     int R, int S, int stride, int pad, void* stream);

/* advmix_conv_tr_w plus an addend in the epilogue: y = conv_transpose(x, w) + addend (addend laid out like y; NULL =
 * none).  Used for the input gradient of a conv whose input has a second consumer. */
int advmix_conv_tr_w_add(const float* x, const float* w, const float* addend, float* y,
                         int N, int Hs, int Ws, int Ck, int Hb, int Wb, int Cn,
                         int R, int S, int stride, int pad, void* stream);

/* Input gradient of a conv whose INPUT is y = act(BN(c) + residual) of a train-mode BatchNorm
 * (pose_hrnet.py:41-57 backward): g = (conv_transpose(x, w) + addend) * act'(y) is written to g_out and the two
 * BatchNorm-backward channel sums are accumulated with fp64 atomics into stats[2][Cn][ns] (zero on entry):
 * stats[0] += sum g, stats[1] += sum g * (c - mean) * invstd.  The sign of y (act != ADVMIX_ACT_NONE): from act_mask, the
 * bit-per-element mask advmix_norm_apply_slots wrote ([rows][Cn / 4] bytes; Cn % 16 == 0), or - act_mask NULL - recomputed
 * from c as fmaf((c - mean) * invstd, gamma, beta) > 0, which is what advmix_norm_apply_slots evaluated when there was no
 * residual (round 4: the fp32 y was read for its sign alone - 12.6 of 65 MB per launch at 32 x 64 x 48 x 32).
 * *stats_ns: in = slots per channel (0 = library default, a power of two <= 64), out = the number used.
 * Returns ADVMIX_EINVAL without launching when the shape is not served (grid K split, Ck % 16 != 0, >= 2 GiB);
 * the caller then runs advmix_conv_tr_w_add + advmix_norm_bwd.  Replaces torch autograd's cudnn_batch_norm_backward
 * reduction pass. */
int advmix_conv_tr_w_bnb(const float* x, const float* w, const float* addend, float* g_out,
                         int N, int Hs, int Ws, int Ck, int Hb, int Wb, int Cn,
                         int R, int S, int stride, int pad,
                         const unsigned char* act_mask, const float* bn_c, const float* bn_mean, const float* bn_invstd,
                         const float* bn_gamma, const float* bn_beta, int act, double* stats, int* stats_ns, void* stream);

/* ---- Winograd F(2x2,3x3) path for the 3x3 / stride 1 / pad 1 C -> C convs of HRNet's high-resolution branches
 * (lib/models/pose_hrnet.py:22-57: BasicBlock conv3x3 + BatchNorm2d (+ residual) + ReLU; 32 -> 32 @64x48 and
 * 64 -> 64 @32x24 are 47 % of the forward FLOPs at 256x192).  16 multiplies per 2x2 outputs and channel pair instead of 36;
 * fp32 throughout.  The filters are transformed ONCE per forward pass into a side buffer (advmix_wino_weights), never
 * inside the conv; the conv entry points keep the fused epilogues of advmix_conv_fwd_ex / advmix_conv_tr_w_bnb. */

/* 0 when advmix_conv3x3_wino_fwd / _dgrad do not serve [N,H,W,Ci] -> [N,H,W,Co] (they do for H, W even, Ci in {32, 48, 64, 96, 128},
 * Co % 16 == 0); otherwise the number of workgroups the launch would have (blocks of 32 output tiles x column tiles of 32). */
int advmix_conv_wino_config(int N, int H, int W, int Ci, int Co);
/* floats of ONE transformed image (forward or input gradient) of a [Co][3][3][Ci] filter bank: 16 * ceil(Co / 32) * 32 * Ci
 * (the n dimension is padded to whole column tiles of 32 with zero filters). */
int64_t advmix_wino_u_floats(int Co, int Ci);
/* Transform the filters of several convs in one launch.  ents (device): records {const float* w; float* u; int Cn, Ck, role,
 * blk0;} - role 0: the forward image of w[Cn][3][3][Ck] (n = Cout, k = Cin), role 1: the input-gradient image of
 * w[Ck][3][3][Cn] (n = Cin, k = Cout, taps rotated by 180 degrees); a record owns the ceil(Cn / 32) * (Ck / 8) workgroups from
 * blk0 on.  blk_ent (device): record index of each of the `blocks` workgroups.  u is written in the order the conv's lanes
 * read it: u[n / 32][xi][k / 8][32 * ((k % 8) / 4) + n % 32][k % 4], xi = 4 * row + column of G g G^T. */
int advmix_wino_weights(const void* ents, const int* blk_ent, int blocks, void* stream);
/* advmix_conv_fwd_ex (no bias) from the role 0 image u: same epilogue arguments, same results to rounding.  ADVMIX_EINVAL
 * (nothing launched) when advmix_conv_wino_config is 0, or in deterministic mode with stats. */
int advmix_conv3x3_wino_fwd(const float* x, const float* u, float* y, int N, int H, int W, int Ci, int Co,
                            const float* bn_gamma, const float* bn_beta, const float* bn_rm, const float* bn_rv,
                            float bn_eps, const float* residual, int act, double* stats, int* stats_ns, void* stream);
/* advmix_conv3x3_wino_fwd's train role (raw output y + its column sums into stats) on an input that is the RAW output c_in of
 * the preceding conv: that conv's train-mode BatchNorm + ReLU is applied while the input patch is staged, so
 * y = conv(relu(BN_train(c_in))) without the advmix_norm_apply_slots launch and without the activation tensor.  in_slots: the
 * column (sum, sum of squares) of c_in as the producer's epilogue left them, [2][in_ns][Ci], in_ns a power of two <= 16.  The
 * launch derives mean / invstd (biased variance, in_eps), writes them to in_mean / in_invstd (saved for the backward pass),
 * updates in_rmean / in_rvar (in_momentum, unbiased variance; both may be NULL) and increments *in_nbt (may be NULL).  The
 * activation is spelled as norm_apply_slots spells it - fma((c - mean) * invstd, gamma, beta) - so the backward pass's
 * "sign from c" epilogue (advmix_conv3x3_wino_dgrad with bn_gamma / bn_beta) agrees with it bit for bit.  Replaces the
 * BatchNorm2d + ReLU between conv1 and conv2 of lib/models/pose_hrnet.py:41-57 (BasicBlock) and :77-88 (Bottleneck).
 * ADVMIX_EINVAL (nothing launched) where advmix_conv3x3_wino_fwd refuses, for in_ns > 16, in deterministic mode. */
int advmix_conv3x3_wino_fwd_inbn(const float* c_in, const float* u, float* y, int N, int H, int W, int Ci, int Co,
                                 const double* in_slots, int in_ns, const float* in_gamma, const float* in_beta,
                                 float in_eps, float* in_mean, float* in_invstd, float* in_rmean, float* in_rvar,
                                 long long* in_nbt, float in_momentum, double* stats, int* stats_ns, void* stream);
/* advmix_conv_tr_w_add (bn_c NULL) / advmix_conv_tr_w_bnb (bn_c given) from the role 1 image u: dx[N,H,W,Ci] from
 * dy[N,H,W,Co]; addend, bn_c and act_mask are laid out like dx. */
int advmix_conv3x3_wino_dgrad(const float* dy, const float* u, const float* addend, float* dx, int N, int H, int W,
                              int Co, int Ci, const unsigned char* act_mask, const float* bn_c, const float* bn_mean,
                              const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int act,
                              double* stats, int* stats_ns, void* stream);

/* Small maps with 256 input channels (csrc/conv_smap.hip, round 5): HRNet's lowest-resolution branch (pose_hrnet.py:22-57 at
 * 256x192: 3x3 256 -> 256 @8x6) - one workgroup per image and 32 output channels, the padded image staged in LDS once, K split
 * over the workgroup's eight waves, filters read in MFMA fragment order from a side buffer (advmix_smap_weights, once per
 * forward pass like the Winograd images).  Direct convolution (no transform): same sums as advmix_conv_fwd_ex in another order.
 * advmix_conv_smap_config: 0 when not served (served: Ci == 256, Co % 32 == 0, H * W <= 48, (H + 2) * (W + 2) <= 80), else the
 * number of workgroups (N * Co / 32).  advmix_smap_u_floats: floats of ONE image of a [Co][3][3][Ci] bank (9 * Co * Ci).
 * advmix_smap_weights: records as advmix_wino_weights' (role 0: forward image of w[Cn][3][3][Ck]; role 1: input-gradient
 * image of w[Ck][3][3][Cn], taps rotated); a record owns (Cn / 32) * (Ck / 32) * 36 workgroups; u is written as
 * u[n / 32][k / 32][tap][(k % 32) / 16][(n % 32) / 16][16 * ((k % 16) / 4) + n % 16][k % 4].
 * advmix_conv3x3_smap_fwd / _dgrad: the arguments and epilogues of advmix_conv3x3_wino_fwd / _dgrad. */
int advmix_conv_smap_config(int N, int H, int W, int Ci, int Co);
int64_t advmix_smap_u_floats(int Co, int Ci);
int advmix_smap_weights(const void* ents, const int* blk_ent, int blocks, void* stream);
int advmix_conv3x3_smap_fwd(const float* x, const float* u, float* y, int N, int H, int W, int Ci, int Co,
                            const float* bn_gamma, const float* bn_beta, const float* bn_rm, const float* bn_rv,
                            float bn_eps, const float* residual, int act, double* stats, int* stats_ns, void* stream);
int advmix_conv3x3_smap_dgrad(const float* dy, const float* u, const float* addend, float* dx, int N, int H, int W,
                              int Co, int Ci, const unsigned char* act_mask, const float* bn_c, const float* bn_mean,
                              const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int act,
                              double* stats, int* stats_ns, void* stream);

/* The same workgroup shape with Winograd F(2x2,3x3) inside (conv_smapw): H and W even as well (an 8x6 map = 12 output tiles =
 * one MFMA row tile per position of the transformed patch; each of the eight waves multiplies two positions over all 256
 * channels; the input transform happens on the way from LDS to the MFMA).  Images: 16 * Co * Ci floats (advmix_smapw_u_floats),
 * written by advmix_smapw_weights (records as above, (Cn / 32) * 32 workgroups each, Ck == 256) as
 * u[n / 32][xi / 2][k / 16][xi % 2][(n % 32) / 16][16 * ((k % 16) / 4) + n % 16][k % 4], xi = 4 * row + column of G g G^T. */
int advmix_conv_smapw_config(int N, int H, int W, int Ci, int Co);
int64_t advmix_smapw_u_floats(int Co, int Ci);
int advmix_smapw_weights(const void* ents, const int* blk_ent, int blocks, void* stream);
int advmix_conv3x3_smapw_fwd(const float* x, const float* u, float* y, int N, int H, int W, int Ci, int Co,
                             const float* bn_gamma, const float* bn_beta, const float* bn_rm, const float* bn_rv,
                             float bn_eps, const float* residual, int act, double* stats, int* stats_ns, void* stream);
int advmix_conv3x3_smapw_dgrad(const float* dy, const float* u, const float* addend, float* dx, int N, int H, int W,
                               int Co, int Ci, const unsigned char* act_mask, const float* bn_c, const float* bn_mean,
                               const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int act,
                               double* stats, int* stats_ns, void* stream);

/* 1x1 / stride 1 convs reading 64 channels and writing 256 over many pixels (csrc/conv_pw.hip, round 5): the last conv and
 * the shortcut of a Bottleneck forward, the input gradient of its first conv (pose_hrnet.py:59-98 on the 64x48 map: as much
 * HBM time as MFMA time).  A workgroup = 128 pixels x all 256 channels, the input tile loaded once, filters read in MFMA
 * fragment order from a side buffer (advmix_pw_weights, once per forward pass like the Winograd images), eight passes of 32
 * channels with the fused epilogue in the natural layout after each.  advmix_conv_pw_config: 0 when not served (served: Ci ==
 * 64 read, Co == 256 written), else the number of workgroups.  advmix_pw_u_floats: floats of one image (Co * Ci).
 * advmix_pw_weights: records as advmix_wino_weights' (role 0: image of w[256][1][1][64] for the forward; role 1: image of
 * w[64][1][1][256] for the input gradient), 64 workgroups each; u[n / 32][(k % 32) / 4][32 * (k / 32) + n % 32][k % 4].
 * advmix_conv1x1_pw_fwd / _dgrad: the arguments and epilogues of advmix_conv3x3_wino_fwd / _dgrad. */
int advmix_conv_pw_config(int N, int H, int W, int Ci, int Co);
int64_t advmix_pw_u_floats(int Co, int Ci);
int advmix_pw_weights(const void* ents, const int* blk_ent, int blocks, void* stream);
int advmix_conv1x1_pw_fwd(const float* x, const float* u, float* y, int N, int H, int W, int Ci, int Co,
                          const float* bn_gamma, const float* bn_beta, const float* bn_rm, const float* bn_rv,
                          float bn_eps, const float* residual, int act, double* stats, int* stats_ns, void* stream);
int advmix_conv1x1_pw_dgrad(const float* dy, const float* u, const float* addend, float* dx, int N, int H, int W,
                            int Co, int Ci, const unsigned char* act_mask, const float* bn_c, const float* bn_mean,
                            const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int act,
                            double* stats, int* stats_ns, void* stream);

/* 4x4 / stride 2 / pad 1 convolutions as Winograd F(3x3, 2x2) per input phase (csrc/conv_wino4.hip, round 5): the U-Net
 * generator's down convs (lib/models/Unet_generator.py:60-62) and the input gradients of its transposed convs (:63-65, :74-76,
 * :84-86).  A 4x4 / stride-2 conv is four 2x2 / stride-1 convs on the input's parity phases; F(3x3, 2x2) multiplies 16 times per
 * 3x3 output tile, phase and channel pair instead of 36.  Not fused: an input transform x -> V[16][tiles][4 Ci], the 16 GEMMs
 * V[xi] . U[xi]^T in one launch of the direct kernel, an output transform M -> y.
 * advmix_wino4_u_floats: floats of the transformed filters of w[Co][4][4][Ci] (16 * Co * 4 * Ci), u[xi][co][(p, q), ci].
 * advmix_w4_weights: records as advmix_wino_weights' (Cn = Co, Ck = Ci, role 0); a record owns Co * 4 * Ci / 256 workgroups.
 * advmix_conv4x4s2_wino_ws_floats: floats of scratch a launch needs (0: shape not served - H, W even, Ci % 8 == 0, Co % 4 == 0,
 * every buffer below 2 GiB).  advmix_conv4x4s2_wino_fwd: y[N][H/2][W/2][Co] = conv(x[N][H][W][Ci]) + bias; ADVMIX_EINVAL for
 * unserved shapes (the caller runs advmix_conv_fwd).
 * advmix_conv4x4s2_wino_wgrad: the weight gradient of such a conv (filters [Cl][4][4][Ch]) ACCUMULATED into dw, from its input
 * hi[N][H][W][Ch] and its output gradient lo[N][H/2][W/2][Cl] (a transposed conv: output gradient / input): the adjoint output
 * transform of lo, the input transform of hi (or ``v``: the one a preceding advmix_conv4x4s2_wino_fwd(hi, ...) left at the start
 * of its scratch), 16 weight-gradient GEMMs as one grouped launch, the adjoint filter transform.  Served: the forward's shapes
 * with Cl % 64 == 0 and Ch % 32 == 0; not in deterministic mode.  _wgrad_ws_floats: its scratch (0 = not served).
 *
 * The transposed form (ConvTranspose2d(k 4, s 2, p 1) forward, Unet_generator.py:63-65,74-76,84-86; the input gradient of the down
 * convs): tile t of floor(Hl / 3) + 1 per axis owns output rows 6 t - 1 ... 6 t + 4, one 4x4 low-resolution patch serves the four
 * output phases, 16 GEMMs [tiles x Cl] . [Cl x 4 Ch].  advmix_w4t_weights: records as advmix_w4_weights with role 1 (a record
 * owns (Cl / 32) * (Ch / 32) * 4 workgroups), u'[xi][(P, Q), ch][cl].  advmix_deconv4x4s2_wino_fwd: y[N][2 Hl][2 Wl][Ch] =
 * conv_transpose(x[N][Hl][Wl][Cl]) + bias + addend with filters [Cl][4][4][Ch]; served: Cl, Ch multiples of 32. */
int64_t advmix_wino4_u_floats(int Co, int Ci);
int advmix_w4t_weights(const void* ents, const int* blk_ent, int blocks, void* stream);
int64_t advmix_deconv4x4s2_wino_ws_floats(int N, int Hl, int Wl, int Cl, int Ch);
int advmix_deconv4x4s2_wino_fwd(const float* x, const float* u, const float* bias, const float* addend, float* y, float* ws,
                                int64_t ws_floats, int N, int Hl, int Wl, int Cl, int Ch, void* stream);
int64_t advmix_conv4x4s2_wino_wgrad_ws_floats(int N, int H, int W, int Ch, int Cl, int have_v);
int advmix_conv4x4s2_wino_wgrad(const float* hi, const float* lo, float* dw, const float* v, float* ws, int64_t ws_floats,
                                int N, int H, int W, int Ch, int Cl, void* stream);
int advmix_w4_weights(const void* ents, const int* blk_ent, int blocks, void* stream);
int64_t advmix_conv4x4s2_wino_ws_floats(int N, int H, int W, int Ci, int Co);
int advmix_conv4x4s2_wino_fwd(const float* x, const float* u, const float* bias, float* y, float* ws, int64_t ws_floats,
                              int N, int H, int W, int Ci, int Co, void* stream);

/* Winograd weight gradient, F(3x3, 2x2) (csrc/wgrad_wino.hip, round 5): 16 multiplies per 2x2 tile of dy and channel pair
 * instead of 36.  advmix_wgrad_wino_config: 0 = not served (odd H / W, channels not multiples of 32 or > 256), else the
 * number of (32-tile block, 32 x 32 channel pair) units of one problem.  advmix_conv3x3_wgrad_wino_group: the weight
 * gradients of n (1-8) 3x3 / stride 1 / pad 1 convs of ONE geometry in one launch, ACCUMULATED (fp32 atomics) into
 * dw[i] ([Co][3][3][Ci]): dy[i] [N,H,W,Co], x[i] [N,H,W,Ci].  ADVMIX_EINVAL (nothing launched) for unserved shapes and in
 * deterministic mode.  Replaces autograd's cudnn convolution_backward weight path for pose_hrnet.py:22-57. */
int advmix_wgrad_wino_config(int N, int H, int W, int Ci, int Co);
int advmix_conv3x3_wgrad_wino_group(int n, const float* const* dy, const float* const* x, float* const* dw, int N, int H, int W,
                                    int Co, int Ci, void* stream);
/* The same with a per-problem BatchNorm on the x operand: bn_mean / bn_invstd / bn_gamma / bn_beta are arrays of n pointers
 * (an array, or single entries, may be NULL).  Where entry i is given, x[i] is the RAW output c of the conv preceding problem i
 * and relu(fma((c - mean) * invstd, gamma, beta)) - the saved batch statistics - is applied while x is staged: the activation
 * advmix_conv3x3_wino_fwd_inbn never wrote (weight gradient of conv2 in lib/models/pose_hrnet.py:41-57). */
int advmix_conv3x3_wgrad_wino_group_bn(int n, const float* const* dy, const float* const* x, float* const* dw,
                                       const float* const* bn_mean, const float* const* bn_invstd,
                                       const float* const* bn_gamma, const float* const* bn_beta, int N, int H, int W,
                                       int Co, int Ci, void* stream);

/* Transposed gather with <= 4 output channels: the input gradient of a network's FIRST conv (3 image channels; taken
 * when the images come from the generator - lib/core/function.py:146-160 back-propagates loss_G through the frozen
 * student into G).  Arguments as advmix_conv_tr_w without the bias; one thread per output pixel instead of 32 MFMA
 * columns for 3.  ADVMIX_EINVAL (nothing launched): Cn > 4, Ck not 64 or 128, R*S*Ck > 4096. */
int advmix_conv_tr_narrow(const float* x, const float* w, float* y, int N, int Hs, int Ws, int Ck, int Hb, int Wb, int Cn,
                          int R, int S, int stride, int pad, void* stream);

/* ConvTranspose2d(Cin, Cout <= 4, kernel 4, stride 2, padding 1) forward for NARROW outputs - the U-Net's last layer
 * (Unet_generator.py:51-57, 128 -> 3): a VALU kernel bound by reading x once instead of an MFMA tile 32 columns wide.
 * w in its own layout [Cin][4][4][Cout]; x [N,Hi,Wi,Cin] -> y [N,2Hi,2Wi,Cout].  ADVMIX_EINVAL for other shapes. */
int advmix_deconv4x4s2_narrow(const float* x, const float* w, const float* bias, float* y, int N, int Hi, int Wi,
                              int Ci, int Co, void* stream);

/* The same layer as ONE 1 x 1 transposed-weight convolution on the matrix pipe - every input pixel's 16 x Cout products into
 * ws (advmix_deconv4x4s2_narrow_ws_bytes bytes, caller-owned) - and a gather that sums the four products reaching an output
 * pixel (+ bias): x is read once instead of 16 times (395 -> ~125 us for the U-Net's 128 -> 3 tail at 256x192, B = 32).
 * ADVMIX_EINVAL (nothing launched): Cout > 4, Cin % 16, ws too small - call advmix_deconv4x4s2_narrow. */
int64_t advmix_deconv4x4s2_narrow_ws_bytes(int N, int Hi, int Wi, int Co);
int advmix_deconv4x4s2_narrow_gemm(const float* x, const float* w, const float* bias, float* y, float* ws, int64_t ws_bytes,
                                   int N, int Hi, int Wi, int Ci, int Co, void* stream);

/* Which tile configuration the second-generation conv kernel picks (introspection for tests / tuning):
 * 1 = 128x32, 2 = 128x64, 3 = 64x64, 4 = 64x64 + K split across the grid (atomics), 5 = 32x32 + K split between
 * the four waves of a workgroup, 6 = 64x32 + K split between two wave pairs, 7 = 64x32 with eight waves (two row tiles
 * sharing the weight staging, each K-split four ways); -1 = not served.  mode 0: forward, (Ho, Wo) = output size, Ci = reduction channels;
 * mode 1: input gradient / transposed conv, (Ho, Wo) = the LARGER (gradient) side, Ci = channels reduced over. */
int advmix_conv_direct_config(int mode, int N, int Ho, int Wo, int Ci, int Co, int R, int S, int stride);

/* dw[Ca][R][S][Cb] += sum_p a[p, Ca] * b[gather(p,r,s), Cb]   (fp32 atomics, split over pixels)
 * a: [N,Ha,Wa,Ca] at the conv's OUTPUT resolution, b: [N,Hb,Wb,Cb] at its INPUT resolution.
 * Conv2d: a = dY, b = X.   ConvTranspose2d: a = X, b = dY (gives [Cin][R][S][Cout]). */
int advmix_conv_wgrad(const float* a, const float* b, float* dw,
                      int N, int Ha, int Wa, int Ca, int Hb, int Wb, int Cb,
                      int R, int S, int stride, int pad, void* stream);

/* 2-64 weight gradients of ONE geometry (advmix_conv_wgrad's arguments; a / b / dw are HOST arrays of n device pointers) as
 * one launch.  Weight gradients have no consumer before optimizer.step() (lib/core/function.py:154-155), so the eight 3x3
 * C -> C convs of an HRNet branch (pose_hrnet.py:28-57, four BasicBlocks) are differentiated together at the end of the
 * branch's backward: an eighth of the pixel slices per problem to merge, the 128 x 128 tile.  Same sums as n single calls, in
 * another order (with enough problems that every tile is one pixel slice, the owning workgroup adds its tile with plain
 * accesses: no atomics, run-to-run reproducible - unless two problems share a dw buffer).  Served:
 * Ca % 64 == 0 with Cb % 4 == 0, and 3x3 / stride 1 / 32 -> 32.  ADVMIX_EINVAL without launching otherwise (and in
 * deterministic mode): call advmix_conv_wgrad per problem. */
int advmix_conv_wgrad_group(int n, const float* const* a, const float* const* b, float* const* dw,
                            int N, int Ha, int Wa, int Ca, int Hb, int Wb, int Cb,
                            int R, int S, int stride, int pad, void* stream);

/* 1-16 weight gradients of ANY geometries as one launch (a / b / dw: HOST arrays of n device pointers; geoms: n x 11 ints,
 * advmix_conv_wgrad's (N, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad) per problem): the small strided 3x3 and 1x1 convs of
 * HRNet's fuse layers and transitions (pose_hrnet.py:172-247, 305-337) - 7-23 us of launch latency each alone.  Same sums as
 * n single calls, in another order (fp32 atomics).  Every channel count a multiple of 4.  0 = launched, 1 = not served
 * (nothing launched: call advmix_conv_wgrad per problem), ADVMIX_EINVAL for bad arguments and in deterministic mode. */
int advmix_conv_wgrad_multi(int n, const float* const* a, const float* const* b, float* const* dw,
                            const int* geoms, void* stream);

/* Deterministic variants (bit-reproducible run to run; ops.set_deterministic): the pixel slices / row blocks STORE
 * their partial results into ws and a second launch adds them in slice order - no fp32 atomics.
 * advmix_conv_wgrad_det needs 4 * slices * Ca * R * S * Cb bytes, never more than advmix_wgrad_det_ws_bytes();
 * advmix_bias_grad_det 4 * 1024 * C bytes.  ADVMIX_EINVAL when ws is too small. */
int advmix_conv_wgrad_det(const float* a, const float* b, float* dw,
                          int N, int Ha, int Wa, int Ca, int Hb, int Wb, int Cb,
                          int R, int S, int stride, int pad, void* ws, int64_t ws_bytes, void* stream);
int64_t advmix_wgrad_det_ws_bytes(int Ca, int Cb, int R, int S);
int advmix_bias_grad_det(const float* dy, float* db, int64_t rows, int C, void* ws, int64_t ws_bytes, void* stream);

/* Grouped launch: 2-4 convolution problems of ONE kind in one kernel launch (HRNet runs the same layer on 2-4
 * branches of different resolution - pose_hrnet.py:247-265 - each too small to fill the chip on its own).
 * kind 0: every problem is an advmix_conv_fwd_ex (x = input, y = output; residual / eval BatchNorm / act / stats as
 *         there; either every problem uses the fused epilogue fields or none does);
 * kind 1: every problem is an advmix_conv_tr_w_add (x = upstream gradient [N,Hx,Wx,Cx], y = input gradient, residual =
 *         addend) or every problem is an advmix_conv_tr_w_bnb (bnb_* and stats set).
 * Stride 1 only.  stats_ns: in = slots per channel to use (0 = library default), out = slots used.
 * ADVMIX_EINVAL: the group cannot be served as one launch (NOTHING was launched): launch the problems one by one.
 * (Measured in DESIGN.md section 3: 0.51 of the fp32 matrix peak on HRNet-W32's four 3x3 branch convs at B = 32, against
 * 0.42 for the four launches back to back; the step runner keeps its four launch lanes, which do as well.) */
typedef struct advmix_conv_problem {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    int N, Hx, Wx, Cx, Hy, Wy, Cy, R, S, stride, pad;
    const float *bn_gamma, *bn_beta, *bn_rm, *bn_rv;
    float bn_eps;
    const float* residual;
    int act;
    double* stats;
    int stats_ns;
    const unsigned char* bnb_mask;                        /* kind 1 + stats: as advmix_conv_tr_w_bnb's act_mask ... */
    const float *bnb_c, *bnb_mean, *bnb_invstd, *bnb_gamma, *bnb_beta;
    int bnb_act;
} advmix_conv_problem;
int advmix_conv_group(int kind, int n, advmix_conv_problem* problems, void* stream);

/* out[B][T][A] = in[A][T][B]  (weight re-layout for advmix_conv_tr) */
int advmix_transpose_w(const float* in, float* out, int A, int T, int B, void* stream);

/* db[c] += sum over rows of dy[rows, C] */
int advmix_bias_grad(const float* dy, float* db, int64_t rows, int C, void* stream);

/* ---- normalisation: replaces nn.BatchNorm2d (train / eval, pose_hrnet.py:34 etc.) and
 * nn.InstanceNorm2d(affine=False) (Unet_generator.py:19,43,45).  groups = 1 -> BatchNorm
 * over all N*H*W rows; groups = N -> InstanceNorm over H*W rows of each image.
 * Statistics are accumulated and reduced in fp64 (deterministic block partials, no atomics):
 * E[x^2]-E[x]^2 needs twice the input precision when |mean| >> std.
 * ws: workspace of advmix_norm_ws_bytes(groups, C) bytes. */
int64_t advmix_norm_ws_bytes(int groups, int C);
/* batch statistics -> mean[g,C], invstd[g,C]; if running_mean != NULL updates running stats
 * (momentum, unbiased var) and increments *num_batches_tracked (int64, may be NULL). */
int advmix_norm_stats(const float* x, int groups, int64_t rows_per_group, int C, float eps,
                      float* mean, float* invstd, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, float momentum, void* ws, void* stream);
/* groups == 1 statistics from the sums accumulated by advmix_conv_fwd_ex: partial[2][nbg][C] (slot-major); zeroes them. */
int advmix_norm_finalize(double* partial, int nbg, int64_t rows, int C, float eps, float* mean,
                         float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                         float momentum, void* stream);
/* Train-mode BatchNorm forward with the finalize folded in: reduces the slots[2][ns][C] sums left by
 * advmix_conv_fwd_ex (every workgroup for the channels it streams), writes mean / invstd (saved for backward), updates
 * the running statistics / num_batches_tracked (may be NULL) and applies y = act(BN(c) + residual) in ONE launch.
 * The slots are NOT re-zeroed (the caller zero-fills its per-layer slots once per network pass).
 * act_mask (may be NULL): out, [rows][C / 4] bytes - bit e of byte (row, ch / 4) says whether BN(c) + residual is positive
 * at channel 4 * (ch / 4) + e: all a ReLU / LeakyReLU backward needs of y (advmix_conv_tr_w_bnb reads it instead of y).
 * ADVMIX_EINVAL without launching when C % 4 != 0 or ns is not a power of two <= 64. */
int advmix_norm_apply_slots(const float* c, const double* slots, int ns, int64_t rows, int C, float eps,
                            const float* gamma, const float* beta, const float* residual, float* y, int act,
                            float* mean, float* invstd, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, float momentum, unsigned char* act_mask, void* stream);
/* BatchNorm backward from the slots advmix_conv_tr_w_bnb filled: dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)),
 * dgamma += sum g*xhat, dbeta += sum g (either may be NULL).  g is already multiplied by the activation's slope. */
int advmix_norm_bwd_apply_slots(const float* g, const float* c, const float* mean, const float* invstd,
                                const float* gamma, const double* slots, int ns, int64_t rows, int C,
                                float* dx, float* dgamma, float* dbeta, void* stream);

/* Deterministic statistics (bit-reproducible runs, ops.set_deterministic): call advmix_conv_fwd_ex / advmix_conv_tr_w_bnb
 * with *stats_nbg = -capacity; the epilogue then STORES one partial per row tile, stats[2][C][count] (no atomics), and
 * returns count in *stats_nbg (ADVMIX_EINVAL, nothing launched, if count > capacity).  advmix_stats_fold adds the partials
 * in a fixed order into slots_out[2][1][C], which advmix_norm_apply_slots / advmix_norm_bwd_apply_slots read with ns = 1. */
int advmix_stats_fold(const double* partials, int count, int C, double* slots_out, void* stream);
/* y = act((x - mean)*invstd*gamma + beta + residual); gamma/beta/residual may be NULL.
 * y rows have stride ldy floats (>= C) so the result can land in a channel slice. */
int advmix_norm_apply(const float* x, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, const float* residual,
                      float* y, int ldy, int groups, int64_t rows_per_group, int C, int act,
                      void* stream);
/* eval-mode BN: y = act(x*scale + shift + residual), scale/shift from running stats */
int advmix_bn_eval(const float* x, const float* gamma, const float* beta,
                   const float* running_mean, const float* running_var, float eps,
                   const float* residual, float* y, int64_t rows, int C, int act, void* stream);
/* backward of norm_apply(train): g = dy * act'(y); dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat));
 * dgamma += sum g*xhat, dbeta += sum g (NULL to skip: frozen / affine=False);
 * dres (NULL to skip) = g.  dy/y rows have stride ldy. */
int advmix_norm_bwd(const float* dy, const float* y, int ldy, const float* x,
                    const float* mean, const float* invstd, const float* gamma,
                    float* dx, float* dres, float* dgamma, float* dbeta,
                    int groups, int64_t rows_per_group, int C, int act, void* ws, void* stream);

/* ---- pointwise / data movement (ReLU, LeakyReLU, residual add, cat, nearest-upsample fuse,
 * max-pool): pose_hrnet.py:35,54-55,206,254-265; pose_resnet.py:115; Unet_generator.py:42,44,83 */
/* y[r, yoff + c] = act(x[r, xoff + c]) for c < C, row strides ldx / ldy */
int advmix_act_copy(const float* x, int ldx, float* y, int ldy, int64_t rows, int C, int act,
                    void* stream);
/* dx[r,c] = dy[r,c] * act'(y[r,c]) (y = saved OUTPUT of the activation) */
int advmix_act_bwd(const float* dy, int lddy, const float* y, int ldy, float* dx, int lddx,
                   int64_t rows, int C, int act, void* stream);
/* y = act(sum_j up_{2^shift_j}(in_j)); in_j is [N, H>>shift_j, W>>shift_j, C]; n_in <= 4 */
int advmix_fuse_sum(const float* const* ins_host, const int* shifts_host, int n_in, float* y,
                    int N, int H, int W, int C, int act, void* stream);
/* g = dy*act'(y) (written to g_out); for each j with shift_j>0: din_j = block-sum of g */
int advmix_fuse_sum_bwd(const float* dy, const float* y, float* g_out, float* const* dins_host,
                        const int* shifts_host, int n_in, int N, int H, int W, int C, int act,
                        void* stream);
/* The same in ONE launch, leaving the BatchNorm-backward channel sums (sum g_j, sum g_j * xhat_j) of the sources that
 * are outputs of a train-mode conv + BN (pose_hrnet.py:196-232: the fuse layers end in BN without activation) in
 * their fp64 slots [2][ns][C] (pre-zeroed, added to): advmix_norm_bwd_apply_slots then finishes that BatchNorm's
 * backward in one launch.  bnb_c[j] == NULL: source j has no such target.  ADVMIX_EINVAL = shape not served, nothing
 * launched (use advmix_fuse_sum_bwd). */
int advmix_fuse_sum_bwd_bnb(const float* dy, const float* y, float* g_out, float* const* dins_host,
                            const int* shifts_host, int n_in, int N, int H, int W, int C, int act,
                            const float* const* bnb_c_host, const float* const* bnb_mean_host,
                            const float* const* bnb_invstd_host, double* const* bnb_slots_host, int ns, void* stream);
int advmix_maxpool3x3s2(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C,
                        int Ho, int Wo, void* stream);
int advmix_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int N, int H, int W,
                            int C, int Ho, int Wo, void* stream);
/* y = x * (*s_dev) * s_host  (s_dev may be NULL): scales a gradient by a device-resident scalar */
int advmix_scale_dev(float* y, const float* x, const float* s_dev, float s_host, int64_t n, void* stream);
/* a += alpha * b (n floats) */
int advmix_axpy(float* a, const float* b, float alpha, int64_t n, void* stream);
/* out = a + b (n floats, no aliasing): gradient fan-in of a tensor with two consumers */
int advmix_add(const float* a, const float* b, float* out, int64_t n, void* stream);

/* ---- AdvMix glue: lib/core/function.py:137-144 (cat + softmax-mix), lib/core/loss.py:25-65,
 * lib/core/inference.py:22-49 (argmax), lib/utils/utils.py:89-92 (Adam) */
/* out[N,H,W,3K] (NHWC) = cat_k views[k] (each NCHW [N,3,H,W]) */
int advmix_cat_views(const float* v0, const float* v1, const float* v2, float* out,
                     int N, int H, int W, void* stream);
/* tmp[N,H,W,3] (NHWC) = sum_k softmax(logits[N,H,W,3])_k * view_k (NCHW) */
int advmix_mix_fwd(const float* v0, const float* v1, const float* v2, const float* logits,
                   float* tmp, int N, int H, int W, void* stream);
/* dlogits[N,H,W,3] from dtmp[N,H,W,3] */
int advmix_mix_bwd(const float* v0, const float* v1, const float* v2, const float* logits,
                   const float* dtmp, float* dlogits, int N, int H, int W, void* stream);
/* JointsMSELoss as constructed by the reference (= SmoothL1, loss.py:16-21):
 * loss_out[0] = (0.5/(J*B*HW)) * sum smoothl1(w*(p - t)); grad (NHWC, may be NULL) = d loss/d p.
 * pred is NHWC [B,HW,J]; target is NCHW [B,J,HW] or NHWC (target_nhwc); tw [B,J] or NULL.
 * mse != 0 selects the plain-MSE variant (smooth_L1=True in the reference's inverted flag).
 * loss_out must be zeroed by the caller (accumulated with one atomic per block). */
int advmix_joints_loss(const float* pred, const float* target, int target_nhwc, const float* tw,
                       float* loss_out, float* grad, float grad_scale, int B, int J, int HW,
                       int mse, void* stream);
/* deterministic variant: the <= 512 block sums are stored in ws (>= 4 KiB) and added in block order */
int advmix_joints_loss_det(const float* pred, const float* target, int target_nhwc, const float* tw,
                           float* loss_out, float* grad, float grad_scale, int B, int J, int HW,
                           int mse, void* ws, void* stream);
/* loss_out += scale_a * L(pred, target_a) + scale_b * L(pred, target_b); grad = the same blend of the two gradients; one pass
 * over pred.  target_b may be NULL (a scaled single loss).  ws != NULL (>= 4 KiB): block sums added in block order
 * (deterministic mode).  Replaces the torch arithmetic around the two criterion calls of lib/core/function.py:151-153
 * (loss_D = (1 - alpha) L(out, target) + alpha L(out, teacher)) and :161 (loss_G = -adv_loss_weight L(out, target)). */
int advmix_joints_loss_blend(const float* pred, const float* target_a, int a_nhwc, const float* target_b, int b_nhwc,
                             const float* tw, float* loss_out, float* grad, float scale_a, float scale_b, int B, int J,
                             int HW, int mse, void* ws, void* stream);
/* first-occurrence argmax over HW per (b, j) of an NHWC (nhwc=1) or NCHW heat-map;
 * idx_out[B*J] int32, max_out[B*J] */
int advmix_heatmap_argmax(const float* hm, int nhwc, int32_t* idx_out, float* max_out,
                          int B, int J, int HW, void* stream);
/* flat Adam (no weight decay). hyper (device): [lr, beta1, beta2, eps]; step (device int64) is
 * incremented by the kernel launch itself (graph-replay safe). */
int advmix_adam(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                int64_t* step, void* stream);
/* One torch.optim.SGD step (dampening 0) over a flat buffer (lib/utils/utils.py:80-88): hyper = {lr, momentum,
 * weight_decay, nesterov as 0 / 1} in device memory; buf = the momentum buffer (zero before the first step). */
int advmix_sgd(float* p, const float* g, float* buf, int64_t n, const float* hyper, void* stream);
int advmix_fill(float* p, float value, int64_t n, void* stream);

/* ---- three-view input pipeline on the device -------------------------------------------------
 * Replace tools/train.py:116-126 (ToTensor + Normalize), lib/dataset/advaug.py:111-170 (GridMask as
 * MixCombine calls it) and lib/dataset/JointsDataset.py:412-491 (gaussian target rendering). */
/* base, aug: uint8 [B][H][W][3] device crops (aug may be NULL -> v1 = v0); grid: int32 [B][4] = (d, l, st_h,
 * st_w) of grid_aug's draws, d <= 0 = this sample keeps its image, NULL = no GridMask at all; mean, std_: 3 HOST
 * floats each.  v0 (clean), v1 (AutoAugment), v2 (GridMask; v1 / v2 may be NULL): float32 NCHW [B][3][H][W]. */
int advmix_make_views(const uint8_t* base, const uint8_t* aug, const int32_t* grid, const float* mean,
                      const float* std_, float* v0, float* v1, float* v2, int B, int H, int W, void* stream);
/* joints, vis: float64 [B][J][3] in input-image pixels; g: the (2*tmp_size+1)^2 float32 gaussian patch;
 * grid (optional) applies GridMask's visibility rule first; joints_weight (optional) [J].
 * target [B][J][Hh][Wh], target_weight [B][J], vis_out (optional) [B][J][3]. */
int advmix_render_targets(const double* joints, const double* vis, const int32_t* grid, const float* g,
                          int tmp_size, const float* joints_weight, float* target, float* target_weight,
                          double* vis_out, int B, int J, int H, int W, int Hh, int Wh, void* stream);

/* The AutoAugment view on the device: ImageNetPolicy (lib/dataset/advaug.py:10-108; applied per sample by MixCombine,
 * advaug.py:180-187, from JointsDataset.py:124).  The policy table only reaches equalize / posterize / solarize / invert
 * (Pillow ImageOps look-up tables) and sharpness (ImageFilter.SMOOTH + Image.blend); results are bit-identical to
 * Pillow's.  base / out / tmp: uint8 [B,H,W,3]; ops: int32 [B][4] = {code1, param1, code2, param2}, code 0 = none,
 * 1 equalize, 2 posterize (param = bit mask), 3 solarize (param = float bits of the threshold), 4 invert,
 * 5 sharpness (param = float bits of the blend factor).  The worker draws (advmix_amd.dataset.advaug.autoaug_params). */
int advmix_autoaug(const uint8_t* base, const int32_t* ops, uint8_t* tmp, uint8_t* out, int B, int H, int W,
                   void* stream);

/* ---- validate(): flip test and final predictions ----------------------------------------
 * Replace the numpy round trips of lib/core/function.py:240-261,285-287, lib/utils/transforms.py:16-41,57-107
 * and lib/core/inference.py:52-95. */
/* y = x.flip(3).  x dense NCHW [B,C,H,W]; y dense NCHW (y_nhwc = 0) or NHWC (1). */
int advmix_flip_w(const float* x, float* y, int B, int C, int H, int W, int y_nhwc, void* stream);
/* F = flip_back(flipped, pairs) (W reversed, joint j <- partner[j]); if shift, F[..., 1:] = F[..., :-1];
 * y = out ? (out + F) * 0.5f : F.  All four [B,J,H,W] in the same dense layout (nhwc 0/1); partner[J] int32
 * on the device. */
int advmix_flip_merge(const float* out, const float* flipped, const int32_t* partner, float* y,
                      int B, int J, int H, int W, int nhwc, int shift, void* stream);
/* get_final_preds: per (b, j) first-occurrence argmax, maxvals[B*J], optional +-0.25 px shift (post_process),
 * heat-map coordinates coords[B*J*2] (optional, may be NULL) and image coordinates preds[B*J*2] through the
 * inverse crop transform of center[B*2] / scale[B*2] (device fp32, scale in units of 200 px). */
int advmix_final_preds(const float* hm, int nhwc, const float* center, const float* scale,
                       int B, int J, int H, int W, int post_process,
                       float* coords, float* preds, float* maxvals, void* stream);

/* ---- lib/nms: nms_kernel.cu:33-77 (bitmask) + :90-143 (host greedy) */
/* device bitmask only: boxes_dev [n,5] sorted by score desc -> mask_dev [n, ceil(n/64)] uint64 */
int advmix_nms_mask(const float* boxes_dev, int n, float thresh, uint64_t* mask_dev, void* stream);
/* drop-in for `_nms` (lib/nms/gpu_nms.hpp:1-2): host buffers in/out, synchronous. */
int advmix_nms_host(int* keep_out, int* num_out, const float* boxes_host, int boxes_num,
                    int boxes_dim, float nms_overlap_thresh, int device_id);
/* OKS matrix (float64, lib/nms/nms.py:75-94): ious[n,n] for kpts[n,17*3], areas[n] (device) */
int advmix_oks_matrix(const double* kpts, const double* areas, const double* sigmas, int n, int K,
                      double* ious, void* stream);
/* oks_iou itself (lib/nms/nms.py:75-94): ng persons (g) against nd detections (d) -> ious[ng, nd], all float64 on the
 * device.  use_vis != 0 is ``in_vis_thre`` (nms.py:90-92): joints of DETECTION j whose visibility d_kpts[j][3k+2] is not
 * above vis_thre leave the sum and the divisor (the reference's ``list(vg > t) and list(vd > t)`` is the detection's mask
 * alone); no joint left: 0.  ADVMIX_EINVAL for a NaN threshold. */
int advmix_oks_iou(const double* g_kpts, const double* g_areas, int ng, const double* d_kpts, const double* d_areas,
                   int nd, const double* sigmas, int K, int use_vis, double vis_thre, double* ious, void* stream);
/* The greedy pass of oks_nms (lib/nms/nms.py:97-125) on the device: ``order`` = candidate indices best-first (the host's
 * argsort of the scores: numpy's tie order is part of the reference's result); a candidate is kept unless an earlier
 * kept one has OKS > thresh with it.  keep_out: int32 [n], count_out: int32 [1].  Only the kept indices cross PCIe. */
int advmix_oks_greedy(const double* ious, const int* order, int n, double thresh, int* keep_out, int* count_out,
                      void* stream);
/* The rescoring loop of soft_oks_nms (lib/nms/nms.py:139-177) on the device: keep the head of the order, multiply the
 * remaining scores by exp(-oks^2 / thresh) (float64), re-sort them as ``scores.argsort()[::-1]`` does (among exactly
 * equal scores, whose numpy order depends on the numpy build: a stable ascending sort read backwards), at most max_dets
 * (the reference's 20) times.  order / scores_sorted: the host's initial argsort and
 * scores[order]; scratch_scores fp64 [2n], scratch_order int32 [2n]: caller-owned; keep_out int32 [max_dets], count_out
 * int32 [1].  ADVMIX_EINVAL for n > 8192, a zero or NaN threshold. */
int advmix_soft_oks_greedy(const double* ious, const int* order, const double* scores_sorted, int n, double thresh,
                           int max_dets, double* scratch_scores, int* scratch_order, int* keep_out, int* count_out,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif
