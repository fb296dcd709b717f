"""torch.autograd.Function wrappers over the C ABI (include/advmix_hip.h).

PyTorch is plumbing here: device memory (caching allocator), the current HIP
stream and the autograd tape.  All arithmetic runs in libadvmix_hip.so.

Tensors are logical NCHW with channels_last strides, i.e. dense NHWC in HBM.
Parameter gradients are ACCUMULATED by the kernels straight into ``param.grad``
(a view of the optimizer's flat gradient buffer once ``FlatAdam`` owns the
model); the Functions return None for them.  ``ctx.needs_input_grad`` carries
the reference's three gradient modes (full / input-only after
``set_require_grad(model, False)`` / none), lib/core/function.py:98-104,140,158.
"""
import ctypes

import torch

from ._lib import call, lib

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
_CL = torch.channels_last


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def nhwc(x):
    """Dense NHWC view of a logical-NCHW fp32 CUDA tensor (copies only if needed)."""
    if x.dtype != torch.float32 or not x.is_cuda:
        raise TypeError('advmix_amd ops need fp32 CUDA tensors (no CPU fallback), got %s on %s'
                        % (x.dtype, x.device))
    if x.dim() != 4:
        raise ValueError('expected a 4-D NCHW tensor')
    if not x.is_contiguous(memory_format=_CL):
        x = x.contiguous(memory_format=_CL)
    return x


def empty_nhwc(B, C, H, W, device):
    return torch.empty((B, H, W, C), device=device, dtype=torch.float32).permute(0, 3, 1, 2)


def _grad_buf(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
    return p.grad


_ws_cache = {}


def _workspace(device, nbytes):
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    w = _ws_cache.get(key)
    if w is None or w.numel() * 4 < nbytes:
        w = torch.empty((nbytes + 3) // 4 + 1024, device=device, dtype=torch.float32)
        _ws_cache[key] = w
    return w


def _wt(w, A, T, B):
    """[A][T][B] -> [B][T][A] weight re-layout for the transposed-gather kernel."""
    out = torch.empty(w.numel(), device=w.device, dtype=torch.float32)
    call('advmix_transpose_w', _p(w), _p(out), A, T, B, _st())
    return out


def _check_w(w):
    if not w.is_contiguous(memory_format=_CL):
        raise ValueError('conv weights must be channels_last ([O][R][S][I] in memory)')


# ---------------------------------------------------------------------------------------------
class ConvFn(torch.autograd.Function):
    """nn.Conv2d (square stride / padding, dilation 1, groups 1)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad):
        x = nhwc(x)
        _check_w(w)
        B, Ci, Hi, Wi = x.shape
        Co, _, R, S = w.shape
        Ho = (Hi + 2 * pad - R) // stride + 1
        Wo = (Wi + 2 * pad - S) // stride + 1
        y = empty_nhwc(B, Co, Ho, Wo, x.device)
        call('advmix_conv_fwd', _p(x), _p(w), _p(bias), _p(y), B, Hi, Wi, Ci, Ho, Wo, Co, R, S,
             stride, pad, _st())
        ctx.save_for_backward(x, w, bias)
        ctx.geom = (stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, bias = ctx.saved_tensors
        stride, pad = ctx.geom
        dy = nhwc(dy)
        B, Ci, Hi, Wi = x.shape
        Co, _, R, S = w.shape
        Ho, Wo = dy.shape[2], dy.shape[3]
        dx = None
        if ctx.needs_input_grad[0]:
            wt = _wt(w, Co, R * S, Ci)
            dx = empty_nhwc(B, Ci, Hi, Wi, x.device)
            call('advmix_conv_tr', _p(dy), _p(wt), None, _p(dx), B, Ho, Wo, Co, Hi, Wi, Ci, R, S,
                 stride, pad, _st())
        if ctx.needs_input_grad[1]:
            call('advmix_conv_wgrad', _p(dy), _p(x), _p(_grad_buf(w)), B, Ho, Wo, Co, Hi, Wi, Ci,
                 R, S, stride, pad, _st())
        if bias is not None and ctx.needs_input_grad[2]:
            call('advmix_bias_grad', _p(dy), _p(_grad_buf(bias)), B * Ho * Wo, Co, _st())
        return dx, None, None, None, None


class DeconvFn(torch.autograd.Function):
    """nn.ConvTranspose2d (output_padding 0); weight logical [Cin, Cout, R, S]."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad):
        x = nhwc(x)
        _check_w(w)
        B, Ci, Hi, Wi = x.shape
        _, Co, R, S = w.shape
        Ho = (Hi - 1) * stride - 2 * pad + R
        Wo = (Wi - 1) * stride - 2 * pad + S
        wt = _wt(w, Ci, R * S, Co)                        # [Co][R][S][Ci]
        y = empty_nhwc(B, Co, Ho, Wo, x.device)
        call('advmix_conv_tr', _p(x), _p(wt), _p(bias), _p(y), B, Hi, Wi, Ci, Ho, Wo, Co, R, S,
             stride, pad, _st())
        ctx.save_for_backward(x, w, bias)
        ctx.geom = (stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, bias = ctx.saved_tensors
        stride, pad = ctx.geom
        dy = nhwc(dy)
        B, Ci, Hi, Wi = x.shape
        _, Co, R, S = w.shape
        Ho, Wo = dy.shape[2], dy.shape[3]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = empty_nhwc(B, Ci, Hi, Wi, x.device)
            call('advmix_conv_fwd', _p(dy), _p(w), None, _p(dx), B, Ho, Wo, Co, Hi, Wi, Ci, R, S,
                 stride, pad, _st())
        if ctx.needs_input_grad[1]:
            call('advmix_conv_wgrad', _p(x), _p(dy), _p(_grad_buf(w)), B, Hi, Wi, Ci, Ho, Wo, Co,
                 R, S, stride, pad, _st())
        if bias is not None and ctx.needs_input_grad[2]:
            call('advmix_bias_grad', _p(dy), _p(_grad_buf(bias)), B * Ho * Wo, Co, _st())
        return dx, None, None, None, None


# ---------------------------------------------------------------------------------------------
class BatchNormFn(torch.autograd.Function):
    """y = act(BN(x) + residual).  training: batch stats + running-stat update
    (momentum, unbiased var, num_batches_tracked += 1); eval: running stats."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, nbt, residual, act, training, momentum, eps):
        x = nhwc(x)
        B, C, H, W = x.shape
        rows = B * H * W
        res = nhwc(residual) if residual is not None else None
        y = empty_nhwc(B, C, H, W, x.device)
        ctx.training = training
        if not training:
            call('advmix_bn_eval', _p(x), _p(gamma), _p(beta), _p(rmean), _p(rvar), eps, _p(res),
                 _p(y), rows, C, act, _st())
            return y
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        ws = _workspace(x.device, lib.advmix_norm_ws_bytes(1, C))
        call('advmix_norm_stats', _p(x), 1, rows, C, eps, _p(mean), _p(invstd), _p(rmean), _p(rvar),
             _p(nbt), momentum, _p(ws), _st())
        call('advmix_norm_apply', _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(res), _p(y),
             C, 1, rows, C, act, _st())
        ctx.save_for_backward(x, y, mean, invstd, gamma, beta)
        ctx.act = act
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError('advmix_amd: backward through eval-mode BatchNorm is not on the hot path')
        x, y, mean, invstd, gamma, beta = ctx.saved_tensors
        dy = nhwc(dy)
        B, C, H, W = x.shape
        rows = B * H * W
        act = ctx.act
        need_res = ctx.has_res and ctx.needs_input_grad[6]
        dx = empty_nhwc(B, C, H, W, x.device)
        dres = None
        if need_res:
            dres = dy if act == ACT_NONE else empty_nhwc(B, C, H, W, x.device)
        ws = _workspace(x.device, lib.advmix_norm_ws_bytes(1, C))
        dg = _grad_buf(gamma) if ctx.needs_input_grad[1] else None
        db = _grad_buf(beta) if ctx.needs_input_grad[2] else None
        call('advmix_norm_bwd', _p(dy), _p(y), C, _p(x), _p(mean), _p(invstd), _p(gamma), _p(dx),
             _p(dres) if (need_res and act != ACT_NONE) else None, _p(dg), _p(db), 1, rows, C, act,
             _p(ws), _st())
        return dx, None, None, None, None, None, dres, None, None, None, None


class InstanceNormFn(torch.autograd.Function):
    """y = act(InstanceNorm2d(x)) with affine=False, no running stats, eps 1e-5."""

    @staticmethod
    def forward(ctx, x, act, eps):
        x = nhwc(x)
        B, C, H, W = x.shape
        mean = torch.empty(B * C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(B * C, device=x.device, dtype=torch.float32)
        ws = _workspace(x.device, lib.advmix_norm_ws_bytes(B, C))
        y = empty_nhwc(B, C, H, W, x.device)
        call('advmix_norm_stats', _p(x), B, H * W, C, eps, _p(mean), _p(invstd), None, None, None,
             0.0, _p(ws), _st())
        call('advmix_norm_apply', _p(x), _p(mean), _p(invstd), None, None, None, _p(y), C, B, H * W,
             C, act, _st())
        ctx.save_for_backward(x, y, mean, invstd)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, invstd = ctx.saved_tensors
        dy = nhwc(dy)
        B, C, H, W = x.shape
        dx = empty_nhwc(B, C, H, W, x.device)
        ws = _workspace(x.device, lib.advmix_norm_ws_bytes(B, C))
        call('advmix_norm_bwd', _p(dy), _p(y), C, _p(x), _p(mean), _p(invstd), None, _p(dx), None,
             None, None, B, H * W, C, ctx.act, _p(ws), _st())
        return dx, None, None


# ---------------------------------------------------------------------------------------------
class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        x = nhwc(x)
        B, C, H, W = x.shape
        y = empty_nhwc(B, C, H, W, x.device)
        call('advmix_act_copy', _p(x), C, _p(y), C, B * H * W, C, act, _st())
        ctx.save_for_backward(y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = nhwc(dy)
        B, C, H, W = y.shape
        dx = empty_nhwc(B, C, H, W, y.device)
        call('advmix_act_bwd', _p(dy), C, _p(y), C, _p(dx), C, B * H * W, C, ctx.act, _st())
        return dx, None


class CatActFn(torch.autograd.Function):
    """y = act(cat([a, b], dim=1))  (Unet_generator.py:83 followed by the parent's uprelu)."""

    @staticmethod
    def forward(ctx, a, b, act):
        a, b = nhwc(a), nhwc(b)
        B, Ca, H, W = a.shape
        Cb = b.shape[1]
        C = Ca + Cb
        y = empty_nhwc(B, C, H, W, a.device)
        rows = B * H * W
        base = y.data_ptr()
        call('advmix_act_copy', _p(a), Ca, ctypes.c_void_p(base), C, rows, Ca, act, _st())
        call('advmix_act_copy', _p(b), Cb, ctypes.c_void_p(base + 4 * Ca), C, rows, Cb, act, _st())
        ctx.save_for_backward(y)
        ctx.meta = (Ca, Cb, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        Ca, Cb, act = ctx.meta
        dy = nhwc(dy)
        B, C, H, W = y.shape
        rows = B * H * W
        da = empty_nhwc(B, Ca, H, W, y.device) if ctx.needs_input_grad[0] else None
        db = empty_nhwc(B, Cb, H, W, y.device) if ctx.needs_input_grad[1] else None
        if da is not None:
            call('advmix_act_bwd', _p(dy), C, _p(y), C, _p(da), Ca, rows, Ca, act, _st())
        if db is not None:
            call('advmix_act_bwd', ctypes.c_void_p(dy.data_ptr() + 4 * Ca), C,
                 ctypes.c_void_p(y.data_ptr() + 4 * Ca), C, _p(db), Cb, rows, Cb, act, _st())
        return da, db, None


class FuseSumFn(torch.autograd.Function):
    """y = act(sum_j nearest_up_{2^shift_j}(in_j))  (pose_hrnet.py:206,254-265)."""

    @staticmethod
    def forward(ctx, act, shifts, *ins):
        ins = [nhwc(t) for t in ins]
        n = len(ins)
        j0 = shifts.index(0)
        B, C, H, W = ins[j0].shape
        y = empty_nhwc(B, C, H, W, ins[0].device)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ins])
        sh = (ctypes.c_int * n)(*shifts)
        call('advmix_fuse_sum', ptrs, sh, n, _p(y), B, H, W, C, act, _st())
        ctx.save_for_backward(y)
        ctx.meta = (act, tuple(shifts))
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        act, shifts = ctx.meta
        dy = nhwc(dy)
        B, C, H, W = y.shape
        n = len(shifts)
        g = empty_nhwc(B, C, H, W, y.device)
        outs = []
        for j, s in enumerate(shifts):
            if not ctx.needs_input_grad[2 + j]:
                outs.append(None)
            elif s == 0:
                outs.append(g)
            else:
                outs.append(empty_nhwc(B, C, H >> s, W >> s, y.device))
        ptrs = (ctypes.c_void_p * n)(*[(o.data_ptr() if (o is not None and s > 0) else None)
                                      for o, s in zip(outs, shifts)])
        sh = (ctypes.c_int * n)(*shifts)
        call('advmix_fuse_sum_bwd', _p(dy), _p(y), _p(g), ptrs, sh, n, B, H, W, C, act, _st())
        return (None, None) + tuple(outs)


class MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1)  (pose_resnet.py:115)."""

    @staticmethod
    def forward(ctx, x):
        x = nhwc(x)
        B, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = empty_nhwc(B, C, Ho, Wo, x.device)
        idx = torch.empty(B * Ho * Wo * C, device=x.device, dtype=torch.uint8)
        call('advmix_maxpool3x3s2', _p(x), _p(y), _p(idx), B, H, W, C, Ho, Wo, _st())
        ctx.save_for_backward(idx)
        ctx.shape = (B, C, H, W, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, C, H, W, Ho, Wo = ctx.shape
        dy = nhwc(dy)
        dx = empty_nhwc(B, C, H, W, dy.device)
        call('advmix_maxpool3x3s2_bwd', _p(dy), _p(idx), _p(dx), B, H, W, C, Ho, Wo, _st())
        return dx


# ---------------------------------------------------------------------------------------------
def _nchw3(v):
    if v.dim() != 4 or v.shape[1] != 3 or not v.is_contiguous() or v.dtype != torch.float32 or not v.is_cuda:
        raise ValueError('views must be contiguous NCHW fp32 CUDA tensors [B,3,H,W]')
    return v


def cat_views(views):
    """torch.cat(inputs, dim=1) for the 3 NCHW views -> NHWC [B,9,H,W] (function.py:137)."""
    v0, v1, v2 = (_nchw3(v) for v in views)
    B, _, H, W = v0.shape
    out = empty_nhwc(B, 9, H, W, v0.device)
    call('advmix_cat_views', _p(v0), _p(v1), _p(v2), _p(out), B, H, W, _st())
    return out


class MixFn(torch.autograd.Function):
    """tmp = sum_k views[k] * softmax(logits, 1)[:, k:k+1]  (function.py:138-144)."""

    @staticmethod
    def forward(ctx, logits, v0, v1, v2):
        logits = nhwc(logits)
        v0, v1, v2 = _nchw3(v0), _nchw3(v1), _nchw3(v2)
        B, _, H, W = v0.shape
        tmp = empty_nhwc(B, 3, H, W, v0.device)
        call('advmix_mix_fwd', _p(v0), _p(v1), _p(v2), _p(logits), _p(tmp), B, H, W, _st())
        ctx.save_for_backward(logits, v0, v1, v2)
        return tmp

    @staticmethod
    def backward(ctx, dtmp):
        logits, v0, v1, v2 = ctx.saved_tensors
        dtmp = nhwc(dtmp)
        B, _, H, W = v0.shape
        dl = empty_nhwc(B, 3, H, W, v0.device)
        call('advmix_mix_bwd', _p(v0), _p(v1), _p(v2), _p(logits), _p(dtmp), _p(dl), B, H, W, _st())
        return dl, None, None, None


class JointsLossFn(torch.autograd.Function):
    """JointsMSELoss.forward (lib/core/loss.py:25-65); fused forward + gradient."""

    @staticmethod
    def forward(ctx, pred, target, tw, use_tw, mse):
        pred = nhwc(pred)
        B, J, H, W = pred.shape
        if target.shape != pred.shape:
            raise ValueError('target shape %s != output shape %s' % (tuple(target.shape), tuple(pred.shape)))
        target = target.float()
        if target.is_contiguous():
            t_nhwc = 0
        elif target.is_contiguous(memory_format=_CL):
            t_nhwc = 1
        else:
            target, t_nhwc = target.contiguous(), 0
        w = None
        if use_tw:
            w = tw.float().reshape(B, J).contiguous()
        loss = torch.zeros((), device=pred.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        grad = empty_nhwc(B, J, H, W, pred.device) if need else None
        call('advmix_joints_loss', _p(pred), _p(target), t_nhwc, _p(w), _p(loss), _p(grad), 1.0,
             B, J, H * W, 1 if mse else 0, _st())
        if need:
            ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, dl):
        (grad,) = ctx.saved_tensors
        out = torch.empty_like(grad, memory_format=torch.preserve_format)
        dl = dl.float().contiguous()
        call('advmix_scale_dev', _p(out), _p(grad), _p(dl), 1.0, grad.numel(), _st())
        return out, None, None, None, None


def heatmap_argmax(hm):
    """(idx int32 [B,J], max [B,J]) of a [B,J,H,W] heat-map; first occurrence (numpy.argmax)."""
    if not hm.is_cuda or hm.dtype != torch.float32:
        raise TypeError('heatmap_argmax needs an fp32 CUDA tensor')
    B, J, H, W = hm.shape
    if hm.is_contiguous():
        fmt = 0
    else:
        hm, fmt = nhwc(hm), 1
    idx = torch.empty((B, J), device=hm.device, dtype=torch.int32)
    mx = torch.empty((B, J), device=hm.device, dtype=torch.float32)
    call('advmix_heatmap_argmax', _p(hm), fmt, _p(idx), _p(mx), B, J, H * W, _st())
    return idx, mx


# functional spellings used by the model mirror
def conv2d(x, w, bias=None, stride=1, pad=0):
    return ConvFn.apply(x, w, bias, stride, pad)


def conv_transpose2d(x, w, bias=None, stride=2, pad=1):
    return DeconvFn.apply(x, w, bias, stride, pad)


def batch_norm(x, gamma, beta, rmean, rvar, nbt, residual=None, act=ACT_NONE, training=True,
               momentum=0.1, eps=1e-5):
    return BatchNormFn.apply(x, gamma, beta, rmean, rvar, nbt, residual, act, training, momentum, eps)


def instance_norm(x, act=ACT_NONE, eps=1e-5):
    return InstanceNormFn.apply(x, act, eps)


def activation(x, act):
    return ActFn.apply(x, act)


def cat_act(a, b, act=ACT_NONE):
    return CatActFn.apply(a, b, act)


def fuse_sum(ins, shifts, act=ACT_RELU):
    return FuseSumFn.apply(act, list(shifts), *ins)


def max_pool3x3s2(x):
    return MaxPoolFn.apply(x)


def softmax_mix(logits, views):
    return MixFn.apply(logits, views[0], views[1], views[2])


def joints_loss(pred, target, tw, use_target_weight=True, mse=False):
    return JointsLossFn.apply(pred, target, tw, use_target_weight, mse)
