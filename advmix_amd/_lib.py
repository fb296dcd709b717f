"""ctypes binding of libadvmix_hip.so (the C ABI in include/advmix_hip.h).

The product path has NO CPU fallback: if the HIP library is missing this module
raises at import, and every op raises on a non-zero status.
"""
import ctypes
import os

import torch  # noqa: F401  - FIRST: torch carries its own libamdhip64; loading ours before it would bind
#               the kernels to a second HIP runtime instance and every launch on a torch stream fails

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get('ADVMIX_SO') or os.path.join(_HERE, 'libadvmix_hip.so')   # ADVMIX_SO: debug builds (tools/)

if not os.path.exists(SO_PATH):
    raise ImportError('advmix_amd: %s not found - build it with `python -m advmix_amd.build` '
                      '(hipcc --offload-arch=gfx950); there is no CPU fallback.' % SO_PATH)
lib = ctypes.CDLL(SO_PATH)

_p, _i, _l, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

# name -> argtypes, mirrors include/advmix_hip.h line by line
SIGNATURES = {
    'advmix_version': [],
    'advmix_build_flags': [],
    'advmix_set_option': [ctypes.c_char_p, _i],
    'advmix_conv_fwd': [_p, _p, _p, _p] + [_i] * 11 + [_p],
    'advmix_conv_fwd_ex': [_p, _p, _p, _p] + [_i] * 11 + [_p, _p, _p, _p, _f, _p, _i, _p, _p, _p],
    'advmix_norm_finalize': [_p, _i, _l, _i, _f, _p, _p, _p, _p, _p, _f, _p],
    'advmix_conv_tr': [_p, _p, _p, _p] + [_i] * 11 + [_p],
    'advmix_conv_tr_w': [_p, _p, _p, _p] + [_i] * 11 + [_p],
    'advmix_conv_direct_config': [_i] * 9,
    'advmix_deconv4x4s2_narrow': [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    'advmix_deconv4x4s2_narrow_gemm': [_p, _p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _p],
    'advmix_conv_tr_narrow': [_p, _p, _p] + [_i] * 11 + [_p],
    'advmix_conv_tr_w_add': [_p, _p, _p, _p] + [_i] * 11 + [_p],
    'advmix_conv_tr_w_bnb': [_p, _p, _p, _p] + [_i] * 11 + [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p],
    'advmix_conv_wino_config': [_i] * 5,
    'advmix_wino_weights': [_p, _p, _i, _p],
    'advmix_conv3x3_wino_fwd': [_p, _p, _p] + [_i] * 5 + [_p, _p, _p, _p, _f, _p, _i, _p, _p, _p],
    'advmix_conv3x3_wino_fwd_inbn': [_p, _p, _p] + [_i] * 5 + [_p, _i, _p, _p, _f, _p, _p, _p, _p, _p, _f, _p, _p, _p],
    'advmix_conv3x3_wino_dgrad': [_p, _p, _p, _p] + [_i] * 5 + [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p],
    'advmix_conv_pw_config': [_i] * 5,
    'advmix_pw_weights': [_p, _p, _i, _p],
    'advmix_conv1x1_pw_fwd': [_p, _p, _p] + [_i] * 5 + [_p, _p, _p, _p, _f, _p, _i, _p, _p, _p],
    'advmix_conv1x1_pw_dgrad': [_p, _p, _p, _p] + [_i] * 5 + [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p],
    'advmix_conv_smap_config': [_i] * 5,
    'advmix_conv_smapw_config': [_i] * 5,
    'advmix_smapw_weights': [_p, _p, _i, _p],
    'advmix_conv3x3_smapw_fwd': [_p, _p, _p] + [_i] * 5 + [_p, _p, _p, _p, _f, _p, _i, _p, _p, _p],
    'advmix_conv3x3_smapw_dgrad': [_p, _p, _p, _p] + [_i] * 5 + [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p],
    'advmix_smap_weights': [_p, _p, _i, _p],
    'advmix_conv3x3_smap_fwd': [_p, _p, _p] + [_i] * 5 + [_p, _p, _p, _p, _f, _p, _i, _p, _p, _p],
    'advmix_conv3x3_smap_dgrad': [_p, _p, _p, _p] + [_i] * 5 + [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p],
    'advmix_w4_weights': [_p, _p, _i, _p],
    'advmix_conv4x4s2_wino_fwd': [_p, _p, _p, _p, _p, _l] + [_i] * 5 + [_p],
    'advmix_conv4x4s2_wino_wgrad': [_p, _p, _p, _p, _p, _l] + [_i] * 5 + [_p],
    'advmix_w4t_weights': [_p, _p, _i, _p],
    'advmix_deconv4x4s2_wino_fwd': [_p, _p, _p, _p, _p, _p, _l] + [_i] * 5 + [_p],
    'advmix_wgrad_wino_config': [_i] * 5,
    'advmix_conv3x3_wgrad_wino_group': [_i, _p, _p, _p] + [_i] * 5 + [_p],
    'advmix_conv3x3_wgrad_wino_group_bn': [_i, _p, _p, _p, _p, _p, _p, _p] + [_i] * 5 + [_p],
    'advmix_conv_wgrad': [_p, _p, _p] + [_i] * 11 + [_p],
    'advmix_conv_wgrad_group': [_i, _p, _p, _p] + [_i] * 11 + [_p],
    'advmix_conv_wgrad_multi': [_i, _p, _p, _p, _p, _p],
    'advmix_conv_wgrad_det': [_p, _p, _p] + [_i] * 11 + [_p, _l, _p],
    'advmix_bias_grad_det': [_p, _p, _l, _i, _p, _l, _p],
    'advmix_transpose_w': [_p, _p, _i, _i, _i, _p],
    'advmix_bias_grad': [_p, _p, _l, _i, _p],
    'advmix_norm_stats': [_p, _i, _l, _i, _f, _p, _p, _p, _p, _p, _f, _p, _p],
    'advmix_norm_apply': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _l, _i, _i, _p],
    'advmix_norm_apply_slots': [_p, _p, _i, _l, _i, _f, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _f, _p, _p],
    'advmix_norm_bwd_apply_slots': [_p, _p, _p, _p, _p, _p, _i, _l, _i, _p, _p, _p, _p],
    'advmix_stats_fold': [_p, _i, _i, _p, _p],
    'advmix_bn_eval': [_p, _p, _p, _p, _p, _f, _p, _p, _l, _i, _i, _p],
    'advmix_norm_bwd': [_p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _p, _p],
    'advmix_act_copy': [_p, _i, _p, _i, _l, _i, _i, _p],
    'advmix_act_bwd': [_p, _i, _p, _i, _p, _i, _l, _i, _i, _p],
    'advmix_fuse_sum': [_p, _p, _i, _p, _i, _i, _i, _i, _i, _p],
    'advmix_fuse_sum_bwd': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'advmix_fuse_sum_bwd_bnb': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p],
    'advmix_maxpool3x3s2': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'advmix_maxpool3x3s2_bwd': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'advmix_scale_dev': [_p, _p, _p, _f, _l, _p],
    'advmix_axpy': [_p, _p, _f, _l, _p],
    'advmix_add': [_p, _p, _p, _l, _p],
    'advmix_cat_views': [_p, _p, _p, _p, _i, _i, _i, _p],
    'advmix_mix_fwd': [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    'advmix_mix_bwd': [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'advmix_joints_loss': [_p, _p, _i, _p, _p, _p, _f, _i, _i, _i, _i, _p],
    'advmix_joints_loss_det': [_p, _p, _i, _p, _p, _p, _f, _i, _i, _i, _i, _p, _p],
    'advmix_joints_loss_blend': [_p, _p, _i, _p, _i, _p, _p, _p, _f, _f, _i, _i, _i, _i, _p, _p],
    'advmix_heatmap_argmax': [_p, _i, _p, _p, _i, _i, _i, _p],
    'advmix_adam': [_p, _p, _p, _p, _l, _p, _p, _p],
    'advmix_sgd': [_p, _p, _p, _l, _p, _p],
    'advmix_fill': [_p, _f, _l, _p],
    'advmix_make_views': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'advmix_render_targets': [_p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'advmix_autoaug': [_p, _p, _p, _p, _i, _i, _i, _p],
    'advmix_flip_w': [_p, _p, _i, _i, _i, _i, _i, _p],
    'advmix_flip_merge': [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'advmix_final_preds': [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    'advmix_nms_mask': [_p, _i, _f, _p, _p],
    'advmix_nms_host': [_p, _p, _p, _i, _i, _f, _i],
    'advmix_oks_matrix': [_p, _p, _p, _i, _i, _p, _p],
    'advmix_oks_iou': [_p, _p, _i, _p, _p, _i, _p, _i, _i, ctypes.c_double, _p, _p],
    'advmix_oks_greedy': [_p, _p, _i, ctypes.c_double, _p, _p, _p],
    'advmix_soft_oks_greedy': [_p, _p, _p, _i, ctypes.c_double, _i, _p, _p, _p, _p, _p],
}
for _name, _args in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here == header/library drift
    _fn.argtypes = _args
    _fn.restype = ctypes.c_int


class ConvProblem(ctypes.Structure):
    """advmix_conv_problem (include/advmix_hip.h)."""
    _fields_ = [('x', _p), ('w', _p), ('bias', _p), ('y', _p)] + \
               [(k, _i) for k in ('N', 'Hx', 'Wx', 'Cx', 'Hy', 'Wy', 'Cy', 'R', 'S', 'stride', 'pad')] + \
               [('bn_gamma', _p), ('bn_beta', _p), ('bn_rm', _p), ('bn_rv', _p), ('bn_eps', _f), ('residual', _p),
                ('act', _i), ('stats', _p), ('stats_ns', _i),
                ('bnb_mask', _p), ('bnb_c', _p), ('bnb_mean', _p), ('bnb_invstd', _p), ('bnb_gamma', _p), ('bnb_beta', _p),
                ('bnb_act', _i)]


lib.advmix_conv_group.argtypes = [_i, _i, ctypes.POINTER(ConvProblem), _p]
lib.advmix_conv_group.restype = ctypes.c_int
lib.advmix_norm_ws_bytes.argtypes = [_i, _i]
lib.advmix_norm_ws_bytes.restype = ctypes.c_int64
lib.advmix_wino_u_floats.argtypes = [_i, _i]
lib.advmix_wino_u_floats.restype = ctypes.c_int64
lib.advmix_smapw_u_floats.argtypes = [_i, _i]
lib.advmix_smapw_u_floats.restype = ctypes.c_int64
lib.advmix_pw_u_floats.argtypes = [_i, _i]
lib.advmix_pw_u_floats.restype = ctypes.c_int64
lib.advmix_wino4_u_floats.argtypes = [_i, _i]
lib.advmix_wino4_u_floats.restype = ctypes.c_int64
lib.advmix_conv4x4s2_wino_ws_floats.argtypes = [_i] * 5
lib.advmix_conv4x4s2_wino_ws_floats.restype = ctypes.c_int64
lib.advmix_conv4x4s2_wino_wgrad_ws_floats.argtypes = [_i] * 6
lib.advmix_conv4x4s2_wino_wgrad_ws_floats.restype = ctypes.c_int64
lib.advmix_deconv4x4s2_wino_ws_floats.argtypes = [_i] * 5
lib.advmix_deconv4x4s2_wino_ws_floats.restype = ctypes.c_int64
lib.advmix_smap_u_floats.argtypes = [_i, _i]
lib.advmix_smap_u_floats.restype = ctypes.c_int64
lib.advmix_wgrad_det_ws_bytes.argtypes = [_i, _i, _i, _i]
lib.advmix_wgrad_det_ws_bytes.restype = ctypes.c_int64
lib.advmix_deconv4x4s2_narrow_ws_bytes.argtypes = [_i, _i, _i, _i]
lib.advmix_deconv4x4s2_narrow_ws_bytes.restype = ctypes.c_int64


class AdvmixHipError(RuntimeError):
    pass


_ERR = {1: 'ADVMIX_EINVAL (bad argument / unsupported shape)', 2: 'ADVMIX_ELAUNCH (HIP runtime error)'}


def call(name, *args):
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise AdvmixHipError('%s failed: %s' % (name, _ERR.get(rc, rc)))
