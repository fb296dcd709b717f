"""Shared GPU-side helpers + the smoke step: one tiny AdvMix step on cuda:0 through the HIP
path, checked against the CPU oracle.  Used by __graft_entry__.smoke() and the GPU tests."""
import types

import os
import numpy as np
import torch


def product_models(net, extra, J, D_sd, T_sd, G_sd, downs=6, device='cuda:0', lr=1e-3):
    from advmix_amd import models
    from advmix_amd.config import CfgNode
    cfg = CfgNode({'MODEL': {'NAME': net, 'EXTRA': extra, 'NUM_JOINTS': J, 'INIT_WEIGHTS': False,
                             'PRETRAINED': ''}, 'TRAIN': {'OPTIMIZER': 'adam', 'LR': lr}})
    mod = getattr(models, net)
    D = mod.get_pose_net(cfg, is_train=False)
    T = mod.get_pose_net(cfg, is_train=False)
    G = models.Unet_generator.UnetGenerator(9, 3, downs)
    D.load_state_dict({k: v.detach() for k, v in D_sd.items()}, strict=True)
    T.load_state_dict({k: v.detach() for k, v in T_sd.items()}, strict=True)
    G.load_state_dict({k: v.detach() for k, v in G_sd.items()}, strict=True)
    return cfg, D.to(device), G.to(device), T.to(device)


def _np(t):
    return np.asarray(t.detach().double().cpu() if torch.is_tensor(t) else t, dtype=np.float64)


def assert_close(name, got, ref, tol=1e-3, report=None):
    """Element-wise |got - ref| <= tol + tol * |ref|: the north-star bound (heat-maps / loss within 1e-3 fp32)
    as an absolute AND a relative term per element - no scaling by the tensor's maximum."""
    got, ref = _np(got), _np(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = np.abs(got - ref)
    bound = tol + tol * np.abs(ref)
    ratio = float((err / bound).max()) if err.size else 0.0
    if report is not None:
        report[name] = max(report.get(name, 0.0), ratio)
    bad = err > bound
    assert not bad.any(), '%s: %d/%d elements outside %.0e + %.0e*|ref| (worst err %.3e at |ref| %.3e; max |ref| %.3e)' % (
        name, int(bad.sum()), bad.size, tol, tol, float(err[bad].max()), float(np.abs(ref)[bad][np.argmax(err[bad])]),
        float(np.abs(ref).max()))


def grad_stats(names, got, ref32, ref64):
    """Per-tensor max error relative to the tensor's max, for the HIP path and for the fp32
    oracle, both against the fp64 oracle.  Deep ReLU/BN nets at B=2 are ill-conditioned: a
    pre-activation within rounding of 0 flips its mask and moves a whole branch's gradients by
    O(10%) in ANY fp32 implementation (torch-CPU and torch-ROCm show the same isolated
    outliers), so the criterion is statistical: as accurate as the fp32 oracle."""
    eh, eo = [], []
    for k in names:
        sc = float(ref64[k].abs().max()) + 1e-30
        eh.append(float((got[k].double() - ref64[k]).abs().max()) / sc)
        eo.append(float((ref32[k].double() - ref64[k]).abs().max()) / sc)
    eh, eo = np.array(eh), np.array(eo)
    outliers = int((eh > np.maximum(20 * eo, 1e-2)).sum())
    return float(np.median(eh)), float(np.median(eo)), outliers, eh, eo


def assert_grads(what, names, got, ref32, ref64, k=3):
    mh, mo, outliers, eh, eo = grad_stats(names, got, ref32, ref64)
    if os.environ.get('ADVMIX_TEST_GRAD_TABLE'):            # (diagnosis: the per-tensor errors behind the two medians)
        for k, a, b in zip(names, eh, eo):
            print('  %-44s hip %.2e  fp32-oracle %.2e' % (k, a, b))
    assert mh <= k * mo + 1e-4, '%s: median grad error %.3e vs fp32-oracle %.3e (k = %d)' % (what, mh, mo, k)
    assert outliers <= max(2, 0.03 * len(names)), '%s: %d/%d tensors far outside the fp32-oracle error' % (
        what, outliers, len(names))
    return mh, mo, outliers


def pull_params(model, P):
    """Teacher forcing: overwrite the oracle's parameters with the device path's."""
    with torch.no_grad():
        for k, v in model.named_parameters():
            P[k].copy_(v.detach().cpu())


def match_fraction(model, P, atol=1e-6):
    tot = bad = 0
    for k, v in model.named_parameters():
        d = (v.detach().cpu() - P[k].detach()).abs()
        tot += d.numel()
        bad += int((d > atol).sum())
    return 1.0 - bad / tot


def run_smoke(tag='hrnet_tiny', iters=1, verbose=True):
    from oracle import configs
    from oracle.posenet import calibrate, trainable
    from oracle.step import Adam, advmix_step as oracle_step
    from oracle.synth import synth_batch
    from tests.helpers import build_states
    from advmix_amd.core.function import advmix_step
    from advmix_amd.core.loss import JointsMSELoss
    from advmix_amd.core.evaluate import accuracy
    from advmix_amd.utils.utils import get_optimizer

    net, extra, J, B, H, W = {'hrnet_tiny': ('pose_hrnet', configs.HRNET_TINY, 5, 2, 64, 64),
                              'resnet18_tiny': ('pose_resnet', configs.RES18_TINY, 5, 2, 64, 64)}[tag]
    D, T, G = build_states(net, extra, J, salt=10)
    calib = synth_batch(tag + '.calib', B, J, H, W)[0][0]
    calibrate(net, T, calib, extra)
    calibrate(net, D, calib, extra)
    cfg, mD, mG, mT = product_models(net, extra, J, D, T, G)
    optD, optG = get_optimizer(cfg, mD), get_optimizer(cfg, mG)
    oD, oG = Adam(D, trainable(D)), Adam(G, list(G))
    crit = JointsMSELoss(True)
    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    mD.train(); mG.train(); mT.eval()
    for it in range(iters):
        v, t, w = synth_batch('%s.it%d' % (tag, it), B, J, H, W)
        loss_D, out = advmix_step(args, mD, mG, mT, crit, optD, optG,
                                  [x.cuda().contiguous() for x in v], t.cuda(), w.cuda())
        torch.cuda.synchronize()
        ref = oracle_step(net, extra, D, G, T, oD, oG, v, t, w, alpha=0.1,
                          after_D_step=lambda: pull_params(mD, D))
        pull_params(mG, G)
        assert_close('loss_D it%d' % it, loss_D, ref['loss_D'])
        assert_close('output it%d' % it, out, ref['out2'])
        _, avg_acc, cnt, _ = accuracy(out, t.cuda())
        assert cnt == ref['cnt'] and abs(avg_acc - ref['avg_acc']) < 1e-6, (avg_acc, ref['avg_acc'])
        if verbose:
            print('smoke %s it%d: loss_D %.6f (oracle %.6f) max|out-ref| %.2e (max|ref| %.1f)' % (
                tag, it, float(loss_D), float(ref['loss_D']),
                float((out.cpu() - ref['out2']).abs().max()), float(ref['out2'].abs().max())), flush=True)
    return mD, mG, D, G
