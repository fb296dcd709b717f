"""Whole-network and whole-step parity on a real MI355X: the HIP path (through the C ABI)
against the CPU oracle on the same seeded inputs, and against the reference-generated golden
vectors.  Tolerance 1e-3 abs + 1e-3 rel (north_star: heatmaps/loss within 1e-3 fp32)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import CASES, ALL_FORWARD, DOWNS, gold_files, gold_json, gold_npz, build_states, checksum_close   # noqa: E402

REPORT = {}      # worst err / bound ratio per checked quantity (printed by the tests)
# Gradient criteria of the step tests.
# * The two tiny B = 2 networks (round 6): ELEMENT-WISE against an fp64 evaluation of the same step whose
#   ReLU / LeakyReLU masks (and max-pool winners) are the ones the device's own forward passes produced (ops.SLOT_TAP ->
#   tests/plan_functional.py) - the criterion the pinned-mask tests below apply to whole networks.  It replaces the round-5
#   statistical criteria, which these nets could only meet with a floor for ONE flipped mask and up to five attempts (one
#   pre-activation within rounding of zero lands on the other side than in fp64 and moves every tensor of the tiny generator's
#   gradient by 1-6 %: profiles/r05w_g_step_outlier_rate.log).  No retry anywhere.
# * The three real networks: median per-tensor D-gradient error DERIVED per case and iteration from the fp32 oracle: the device's
#   and the fp32 oracle's D-step gradients are held against an fp64 evaluation of the same step from the same state, and the
#   device may be at most GRAD_K x as far from it as the fp32 oracle is - or as far as the fp32 oracle itself moves when its
#   input is perturbed by an ulp.  One attempt, no floor.
PINNED_STEP = ('hrnet_tiny', 'resnet18_tiny')
PINNED_TOL_STEP = 3e-4   # of the fp64 tensor's max (= PINNED_TOL_NET: train-mode BatchNorm over 8-32 samples per channel in the deepest
                         # maps of a 64x64 B = 2 net amplifies rounding like depth does in the whole HRNet); observed: see the printed worst
GRAD_K = 5       # (first run: hip / fp32-oracle = 1.5-2.6 on the three real networks; the device's own median moves by x1.5 run to run
                 #  at B = 2 - the order of its atomics - so 3 would sit on top of hrnet_w32's 2.6; the bound was 3 x the DEVICE's worst before)
from smoke_step import (product_models, assert_close, run_smoke, assert_grads,      # noqa: E402
                        pull_params, match_fraction)


def _f64_grads(net, extra, D, names, x, tgt, tw, B, J):
    """fp64 re-run of the oracle (loss restated in double; oracle.loss casts to float)."""
    import torch.nn.functional as F
    from oracle.posenet import posenet_forward
    P = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in D.items()}
    for k in names:
        P[k].requires_grad_(True)
    xx = x.double().clone().requires_grad_(True)
    y = posenet_forward(net, P, xx, extra, True)
    w_ = tw.double().reshape(B, J, 1, 1)
    d = y * w_ - tgt.double() * w_
    l = (F.smooth_l1_loss(d, torch.zeros_like(d), reduction='none').mean(dim=(0, 2, 3)) * 0.5).sum() / J
    g = torch.autograd.grad(l, [P[k] for k in names] + [xx])
    return dict(zip(names + ['x'], g))


def _loss_any_dtype(y, tgt, tw, B, J):
    """lib/core/loss.py:25-65 in the dtype of ``y`` (oracle.loss casts to float)."""
    import torch.nn.functional as F
    w_ = tw.to(y.dtype).reshape(B, J, 1, 1)
    d = y * w_ - tgt.to(y.dtype) * w_
    return (F.smooth_l1_loss(d, torch.zeros_like(d), reduction='none').mean(dim=(0, 2, 3)) * 0.5).sum() / J


def _d_step_grads(net, extra, D0, names, tmp, teacher, tgt, tw, B, J, alpha, dtype):
    """Gradient of loss_D = (1 - alpha) L(D(tmp), target) + alpha L(D(tmp), teacher) (function.py:146-153) w.r.t. D's trainable
    parameters, evaluated in ``dtype`` from the state ``D0``."""
    from oracle.posenet import posenet_forward
    P = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in D0.items()}
    for k in names:
        P[k].requires_grad_(True)
    y = posenet_forward(net, P, tmp.to(dtype), extra, True)
    l = (1 - alpha) * _loss_any_dtype(y, tgt, tw, B, J) + alpha * _loss_any_dtype(y, teacher, tw, B, J)
    return dict(zip(names, torch.autograd.grad(l, [P[k] for k in names], allow_unused=True)))


def _g_step_grads_f64(net, extra, D1, G0, views, tgt, tw, B, J, downs):
    """fp64 gradient of loss_G = -L(D'(mix(views, G(views))), target) w.r.t. G (function.py:160-163) through the frozen, already
    updated student D1."""
    from oracle.posenet import posenet_forward
    from oracle.unet import unet_forward
    from oracle.step import mix_views
    P = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in D1.items()}
    G64 = {k: v.detach().clone().double().requires_grad_(True) for k, v in G0.items()}
    v64 = [x.double() for x in views]
    tmp, _ = mix_views(v64, unet_forward(G64, torch.cat(v64, 1), num_downs=downs))
    l = -_loss_any_dtype(posenet_forward(net, P, tmp, extra, True), tgt, tw, B, J)
    return dict(zip(G0, torch.autograd.grad(l, list(G64.values()), allow_unused=True)))


@pytest.mark.parametrize('tag', list(ALL_FORWARD))
def test_forward_backward_vs_oracle_and_golden(tag):
    from oracle.posenet import posenet_forward, calibrate, trainable
    from oracle.unet import unet_forward
    from oracle.loss import joints_loss
    from oracle.synth import synth_batch, strided
    from oracle import detinit
    from advmix_amd import ops
    from advmix_amd.core.loss import JointsMSELoss
    net, extra, J, B, H, W, _ = ALL_FORWARD[tag]
    downs = DOWNS.get(tag, 6)
    g = gold_npz(gold_files(tag)[0])
    D, T, G = build_states(net, extra, J, unet_downs=downs)
    views, tgt, tw = synth_batch(tag, B, J, H, W)
    calibrate(net, D, views[2], extra)
    cfg, mD, mG, _ = product_models(net, extra, J, D, T, G, downs=downs)

    mD.eval()                                             # eval-mode (teacher-style) forward
    with torch.no_grad():
        ye = mD(views[0].cuda())
        ye_ref = posenet_forward(net, D, views[0], extra, False)
    assert_close('eval out', ye, ye_ref, report=REPORT)
    assert_close('eval out vs golden', strided(ye.cpu().contiguous()), g[tag + '.eval_out'], report=REPORT)

    mD.train()                                            # train-mode forward + full backward
    x = views[1].cuda().requires_grad_(True)
    yt = mD(x)
    loss = JointsMSELoss(True)(yt, tgt.cuda(), tw.cuda())
    loss.backward()
    names = trainable(D)
    g64 = _f64_grads(net, extra, D, names, views[1], tgt, tw, B, J)
    for k in names:
        D[k].requires_grad_(True)
    xr = views[1].clone().requires_grad_(True)
    yr = posenet_forward(net, D, xr, extra, True)
    lr = joints_loss(yr, tgt, tw, True)
    g32 = dict(zip(names + ['x'], torch.autograd.grad(lr, [D[k] for k in names] + [xr])))
    assert_close('train out', yt, yr, report=REPORT)
    assert_close('train out vs golden', strided(yt.detach().cpu().contiguous()), g[tag + '.train_out'], report=REPORT)
    assert_close('loss', loss.detach(), lr.detach(), 1e-4)
    assert_close('loss vs golden', [float(loss.detach())], g[tag + '.loss'], 1e-4)
    got = {k: p.grad.detach().cpu() for k, p in mD.named_parameters()}
    got['x'] = x.grad.cpu()
    stats = assert_grads('D grads', names + ['x'], got, g32, g64)
    print(tag, 'D-grad median rel err hip %.2e fp32-oracle %.2e outliers %d' % stats)
    print(tag, 'worst err/bound ratios', {k: round(v, 3) for k, v in REPORT.items()})
    sd = mD.state_dict()
    for k in D:
        if k.endswith(('running_mean', 'running_var')):
            assert_close(k, sd[k], D[k].detach())
        if k.endswith('num_batches_tracked'):
            assert int(sd[k]) == int(D[k])

    for k in G:                                           # generator forward/backward
        G[k].requires_grad_(True)
    gi = ops.cat_views([v.cuda().contiguous() for v in views])
    lg = mG(gi)
    lg_ref = unet_forward(G, torch.cat(views, 1), num_downs=downs)
    assert_close('unet out', lg, lg_ref, report=REPORT)
    assert_close('unet out vs golden', strided(lg.detach().cpu().contiguous()), g[tag + '.unet_out'], report=REPORT)
    proj = detinit.normal(tag + '.gproj', lg_ref.shape)
    (lg * proj.cuda()).sum().backward()
    gg32 = dict(zip(G, torch.autograd.grad((lg_ref * proj).sum(), list(G.values()))))
    G64 = {k: v.detach().double().requires_grad_(True) for k, v in G.items()}
    lg64 = unet_forward(G64, torch.cat(views, 1).double(), num_downs=downs)
    gg64 = dict(zip(G, torch.autograd.grad((lg64 * proj.double()).sum(), list(G64.values()))))
    gmax = max(float(v.abs().max()) for v in gg64.values())
    # conv biases that feed an InstanceNorm have an exactly-zero true gradient: compare those
    # against the network's gradient scale instead of their own (pure rounding noise)
    live = [k for k in G if float(gg64[k].abs().max()) > 1e-4 * gmax]
    gotG = {k: p.grad.detach().cpu() for k, p in mG.named_parameters()}
    # (k = GRAD_K like the step tests' medians, not 3: WHICH ReLU masks flip is a lottery over forward values that agree to 2e-6,
    #  not a noise level - the 6-down U-Net at 512x512 drew 1.5-3.8e-3 / 4.2-4.7e-3 without / with the transposed Winograd form
    #  against the fp32 oracle's 1.5e-3, and with the masks pinned both agree with fp64 to 3e-6 on every tensor:
    #  test_generator_gradients_with_pinned_masks is the criterion that measures kernels; profiles/EXPERIMENTS.md K2)
    stats = assert_grads('G grads', live, gotG, gg32, gg64, k=GRAD_K)
    for k in G:
        if k not in live:
            assert float(gotG[k].abs().max()) <= 1e-3 * gmax, k
    print(tag, 'G-grad median rel err hip %.2e fp32-oracle %.2e outliers %d' % stats)


def _pinned_rel_errors(got, want, names):
    """{name: max |got - want| / max |want|} over the tensors whose fp64 gradient is not identically ~0; the others (conv biases
    under an InstanceNorm, frozen inputs) must be rounding noise against the largest gradient."""
    gmax = max(float(want[k].abs().max()) for k in names if want[k] is not None)
    rel = {}
    for k in names:
        if want[k] is None or float(want[k].abs().max()) <= 1e-6 * gmax:
            assert float(got[k].abs().max()) <= 1e-4 * gmax, (k, float(got[k].abs().max()), gmax)
            continue
        rel[k] = float((got[k].double() - want[k]).abs().max() / want[k].abs().max())
    return rel


def _pinned_step_gradients(tag, it, mD, mG, rec, gD_dev, gG_dev, dn, views, tgt, tw, B, J):
    """Both gradients of one AdvMix iteration (function.py:146-163) element-wise: the fp64 evaluation of plan.Plan's steps
    (tests/plan_functional.py, held to the oracle on the CPU) from the DEVICE's parameters and inputs, every activation on the
    side of zero the device's own forward pass put it on.
    D step: d[(1 - alpha) L(D0(tmp), target) + alpha L(D0(tmp), teacher)] / d D0.  G step: d[-L(D1(mix(views, G0(views))), target)] / d G0
    through the frozen, already updated student D1 (train-mode BatchNorm)."""
    from oracle.step import mix_views
    from plan_functional import interpret, pins_of
    PD, PG = mD.plan, mG.plan
    # --- D step
    W = {k: x_.clone().requires_grad_(True) for k, x_ in rec['W_D0'].items()}
    pin, pool = pins_of(PD, rec['D_a'][0])
    y = interpret(PD, W, rec['tmp'].double(), pin=pin, pool_src=pool)[PD.out]
    l = 0.9 * _loss_any_dtype(y, tgt, tw, B, J) + 0.1 * _loss_any_dtype(y, rec['teacher'].double(), tw, B, J)
    g64 = dict(zip(dn, torch.autograd.grad(l, [W[k] for k in dn], allow_unused=True)))
    assert all(torch.equal(rec['gD'][k], gD_dev[k]) for k in dn)            # (phase b left D's gradient buffer alone)
    rel_D = _pinned_rel_errors(gD_dev, g64, dn)
    # --- G step
    WG = {k: x_.clone().requires_grad_(True) for k, x_ in rec['W_G0'].items()}
    pinG, poolG = pins_of(PG, rec['G_a'][0])
    v64 = [x_.double() for x_ in views]
    logits = interpret(PG, WG, torch.cat(v64, 1), pin=pinG, pool_src=poolG)[PG.out]
    tmp64, _ = mix_views(v64, logits)
    assert float((tmp64.detach() - rec['tmp'].double()).abs().max()) <= 1e-4       # the same mixed image, to fp32 rounding (observed 1e-5 at |tmp| ~ 2)
    pin1, pool1 = pins_of(PD, rec['D_b'][0])
    y1 = interpret(PD, rec['W_D1'], tmp64, pin=pin1, pool_src=pool1)[PD.out]
    gn = list(WG)
    gG64 = dict(zip(gn, torch.autograd.grad(-_loss_any_dtype(y1, tgt, tw, B, J), [WG[k] for k in gn], allow_unused=True)))
    rel_G = _pinned_rel_errors(gG_dev, gG64, gn)
    for what, rel in (('D-step', rel_D), ('G-step', rel_G)):
        top = sorted(rel.items(), key=lambda kv: -kv[1])[:3]
        print(tag, 'it', it, what, 'gradients vs fp64 with the device\'s masks: worst of %d tensors' % len(rel), [(k, '%.2e' % e) for k, e in top],
              'median %.2e' % float(np.median(list(rel.values()))))
        assert top[0][1] <= PINNED_TOL_STEP, (what, top)


@pytest.mark.parametrize('tag', list(CASES))
def test_advmix_and_plain_steps_vs_oracle_and_golden(tag):
    """Reference lr (1e-3), teacher-forced: after each device-side update the oracle adopts the
    device weights, so every compared quantity is computed from identical parameters.

    Outputs, losses, golden values, update match fractions and running statistics for all five cases; the gradients of the two
    tiny networks element-wise against fp64 with the device's own masks (PINNED_STEP), those of the three real networks
    statistically against the fp32 oracle's own error.  One attempt each."""
    from oracle.posenet import calibrate, trainable
    from oracle.step import Adam, advmix_step as ostep, plain_step as oplain
    from oracle.synth import synth_batch, strided
    from advmix_amd.core.function import advmix_step, plain_step
    from advmix_amd.core.loss import JointsMSELoss
    from advmix_amd.utils.utils import get_optimizer
    net, extra, J, B, H, W, iters = CASES[tag]
    downs = DOWNS.get(tag, 6)
    g = gold_npz(gold_files(tag)[1])
    meta = gold_json(gold_files(tag)[2])[tag]
    D, T, G = build_states(net, extra, J, unet_downs=downs, salt=10)
    calib = synth_batch(tag + '.calib', B, J, H, W)[0][0]
    calibrate(net, T, calib, extra)
    calibrate(net, D, calib, extra)
    cfg, mD, mG, mT = product_models(net, extra, J, D, T, G, downs=downs)
    optD, optG = get_optimizer(cfg, mD), get_optimizer(cfg, mG)
    oD, oG = Adam(D, trainable(D)), Adam(G, list(G))
    crit = JointsMSELoss(True)
    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    mD.train(); mG.train(); mT.eval()
    fracs, loose = [], []
    for it in range(iters):
        v, t, w = synth_batch('%s.it%d' % (tag, it), B, J, H, W)
        before = {k: p.detach().clone() for k, p in mD.named_parameters()}
        if tag in PINNED_STEP:
            # the same calls advmix_step makes on one rank (core/function.py), with the device's intermediate values recorded
            # from the step's OWN forward passes: tmp, the teacher's heat-maps, the parameters each pass ran on, and every
            # activation's value (ops.SLOT_TAP) - what the fp64 evaluations below pin their masks to
            from advmix_amd.core.function import advmix_phase_a, advmix_phase_b
            from plan_functional import SlotRecorder
            inputs_dev = [x.cuda().contiguous() for x in v]
            W_D0 = {k: p.detach().cpu().double() for k, p in mD.named_parameters()}
            W_G0 = {k: p.detach().cpu().double() for k, p in mG.named_parameters()}
            with SlotRecorder() as rec_a:
                loss_D, tmp_dev = advmix_phase_a(args, mD, mG, mT, crit, optD, inputs_dev, t.cuda(), w.cuda())
            gD_a = {k: p.grad.detach().cpu().clone() for k, p in mD.named_parameters()}
            with torch.no_grad():
                teacher_dev = mT(inputs_dev[0]).detach().cpu()           # eval mode: no statistics, no atomics - the value phase a used
            with SlotRecorder() as rec_b:
                out = advmix_phase_b(args, mD, crit, optD, optG, tmp_dev, t.cuda(), w.cuda())
            W_D1 = {k: p.detach().cpu().double() for k, p in mD.named_parameters()}
            optG.step()
            pinned = dict(tmp=tmp_dev.detach().cpu(), teacher=teacher_dev, W_D0=W_D0, W_D1=W_D1, W_G0=W_G0, gD=gD_a,
                          D_a=rec_a.of(mD), G_a=rec_a.of(mG), D_b=rec_b.of(mD))
            assert len(pinned['D_a']) == len(pinned['G_a']) == len(pinned['D_b']) == 1, [len(pinned[k]) for k in ('D_a', 'G_a', 'D_b')]
        else:
            loss_D, out = advmix_step(args, mD, mG, mT, crit, optD, optG,
                                      [x.cuda().contiguous() for x in v], t.cuda(), w.cuda())

        # Adam step 1 is lr*g/(|g| + 1e-8): identical to ~1e-6 where |g| >> eps, and sensitive to the gradient's own
        # rounding error where |g| ~ eps (the O(1)-heat-map fixtures have many such elements: d(update) =
        # lr * eps/(|g|+eps) * (dg/g) ~ 1e-5 at |g| = 1e-8, dg/g = 1e-2); later steps are lr*m/sqrt(v): continuous in
        # g, so they agree to (grad rel. error) * lr
        atol = 1e-5 if it == 0 else 1e-4

        def force():
            fracs.append(match_fraction(mD, D, atol))     # oracle's own update vs the device's
            loose.append(match_fraction(mD, D, 10 * atol))
            pull_params(mD, D)
        D0 = {k: x_.detach().clone() for k, x_ in D.items()}          # the state both D-step gradients were taken at
        G0 = {k: x_.detach().clone() for k, x_ in G.items()}
        ref = ostep(net, extra, D, G, T, oD, oG, v, t, w, alpha=0.1, after_D_step=force,
                    unet_kw={'num_downs': downs})
        fracs.append(match_fraction(mG, G, atol))
        gG_dev = {k: p.grad.detach().cpu().clone() for k, p in mG.named_parameters()}
        pull_params(mG, G)
        assert_close('loss_D', loss_D, ref['loss_D'], report=REPORT)
        assert_close('out2', out, ref['out2'], report=REPORT)
        if it == 0:                                       # golden: the reference's own numbers
            want = g[tag + '.losses'][0]
            assert_close('loss_D vs golden', [float(loss_D)], [0.9 * want[0] + 0.1 * want[1]])
            for k, p in mD.named_parameters():            # Adam step 1 moves every element by <= lr
                assert float((p.detach() - before[k]).abs().max()) <= 1e-3 * 1.001, k
        # gradients left in the flat buffers: D grads from the D step, G grads from the G step
        gD = {k: p.grad.detach().cpu() for k, p in mD.named_parameters()}
        dn = [k for k in ref['gD'] if ref['gD'][k] is not None]
        if tag in PINNED_STEP:
            _pinned_step_gradients(tag, it, mD, mG, pinned, gD, gG_dev, dn, v, t, w, B, J)
        else:
            g64 = _d_step_grads(net, extra, D0, dn, ref['tmp'], ref['teacher'], t, w, B, J, 0.1, torch.float64)
            from smoke_step import grad_stats
            mh, mo, outl, _eh, _eo = grad_stats(dn, gD, ref['gD'], g64)
            spread = 0.0                                      # the fp32 oracle against ITSELF with tmp moved by about an ulp, twice
            for seed in (1, 2):
                noise = 1.0 + 2.0 ** -23 * torch.randn(ref['tmp'].shape, generator=torch.Generator().manual_seed(seed))
                gp = _d_step_grads(net, extra, D0, dn, ref['tmp'] * noise, ref['teacher'], t, w, B, J, 0.1, torch.float32)
                spread = max(spread, grad_stats(dn, gp, ref['gD'], {k: ref['gD'][k].double() for k in dn})[0])
            bound = GRAD_K * max(mo, spread) + 1e-4
            print(tag, 'it', it, 'median D-grad error vs fp64: hip %.3e fp32-oracle %.3e, oracle under a 1-ulp perturbation %.3e, '
                  'bound %.3e, outliers %d' % (mh, mo, spread, bound, outl))
            assert mh <= bound, ('median D-grad error vs the fp64 oracle', mh, mo, spread, bound)
            # the G step's gradient, teacher-forced (round 5, VERDICT r4 item 6 b: this replaces the un-forced generator checksum
            # bound of 2 units): through the frozen student the device has just updated - which the oracle adopted - against fp64,
            # statistically no worse than the fp32 oracle's own G gradient
            gG64 = _g_step_grads_f64(net, extra, D, G0, v, t, w, B, J, downs)
            gmax = max(float(x_.abs().max()) for x_ in gG64.values())
            live = [k for k in G0 if float(gG64[k].abs().max()) > 1e-4 * gmax]       # (biases under an InstanceNorm: true gradient 0)
            gmh, gmo, gout, _e1, _e2 = grad_stats(live, gG_dev, ref['gG'], gG64)
            gbound = GRAD_K * gmo + 1e-4                                          # (same rule as D's gradients above)
            assert gmh <= gbound, ('G-step grads it%d' % it, gmh, gmo, gbound)
            assert gout <= max(2, 0.03 * len(live)), ('G-step grads it%d: tensors far outside the fp32-oracle error' % it, gout, float(_e1.max()))
            gstats = (gmh, gmo, gout)
            for k in G0:
                if k not in live:
                    assert float(gG_dev[k].abs().max()) <= 1e-3 * gmax, k
            print(tag, 'it', it, 'G-step grad median rel err vs fp64: hip %.2e fp32-oracle %.2e outliers %d' % gstats)
    print(tag, 'worst err/bound ratios', {k: round(v, 3) for k, v in REPORT.items()})
    sd = mD.state_dict()
    assert int(sd['bn1.num_batches_tracked']) == meta['nbt']       # calib + 2 forwards / iteration
    for k in D:
        if k.endswith(('running_mean', 'running_var')):
            assert_close(k, sd[k], D[k].detach())
    # >= 90 % of all elements within 1 % of lr of the oracle's own update (10 % of lr from the second update on),
    # >= 97 % within ten times that; no element moves by more than lr (asserted above)
    assert min(fracs) >= 0.9 and min(loose) >= 0.97, (fracs, loose)
    assert all(not p.requires_grad for p in mD.parameters())       # function.py:158 leaves D frozen
    print(tag, 'element match fractions after each update', ['%.4f' % f for f in fracs], ['%.4f' % f for f in loose])

    if tag + '.plain_losses' not in g.files:              # (C4 fixture: AdvMix loop only)
        return
    D, _, _ = build_states(net, extra, J, unet_downs=downs, salt=20)   # plain (non-AdvMix) loop, function.py:30-95
    calibrate(net, D, calib, extra)
    cfg, mD, _, _ = product_models(net, extra, J, D, T, G, downs=downs)
    optD, oD = get_optimizer(cfg, mD), Adam(D, trainable(D))
    mD.train()
    for it in range(2):
        v, t, w = synth_batch('%s.plain%d' % (tag, it), B, J, H, W)
        loss, out = plain_step(mD, crit, optD, v[0].cuda(), t.cuda(), w.cuda())
        ref = oplain(net, extra, D, oD, v[0], t, w)
        assert match_fraction(mD, D, 1e-5 if it == 0 else 1e-4) >= 0.9          # (atol: see the AdvMix loop above)
        pull_params(mD, D)
        assert_close('plain loss', loss, ref['loss'])
        assert_close('plain out', out, ref['out'])
        if it == 0:
            assert_close('plain loss vs golden', [float(loss)], [g[tag + '.plain_losses'][0]])


def test_network_parity_at_the_benchmarked_batch():
    """HRNet-W32 256x192 at B = 32, the batch bench.py times: conv_direct dispatches other tile configurations
    there than at B = 2 (128x32 / 128x64 tiles, the in-workgroup K split instead of the grid split).  Eval forward,
    train forward + loss against the CPU oracle (full tensors) and the vectors the REAL reference produced at B = 32;
    then one whole AdvMix step against the real train_advmix's first iteration."""
    from oracle import configs
    from oracle.posenet import posenet_forward, calibrate
    from oracle.loss import joints_loss
    from oracle.synth import synth_batch, strided
    from advmix_amd._lib import lib
    from advmix_amd.core.function import advmix_step
    from advmix_amd.core.loss import JointsMSELoss
    from advmix_amd.utils.utils import get_optimizer
    tag, net, extra, J, B, H, W = 'hrnet_w32_b32', 'pose_hrnet', configs.HRNET_W32, 17, 32, 256, 192
    # the tiles this batch reaches (1 = 128x32, 2 = 128x64, 5 = 32x32 + K split over four waves, 6 = 64x32 + K split over two
    # pairs, 7 = eight waves: two row tiles sharing the weight staging, each K-split four ways)
    cfgs = {lib.advmix_conv_direct_config(0, B, 64 >> i, 48 >> i, 32 << i, 32 << i, 3, 3, 1) for i in range(4)}
    cfgs.add(lib.advmix_conv_direct_config(0, B, 64, 48, 64, 256, 1, 1, 1))
    assert {1, 2, 5, 6, 7} <= cfgs, cfgs
    g = gold_npz('b32_forward.npz')
    D, T, G = build_states(net, extra, J)
    views, tgt, tw = synth_batch(tag, B, J, H, W)
    calibrate(net, D, views[2], extra)
    cfg, mD, mG, mT = product_models(net, extra, J, D, T, G)
    rep = {}
    # ... and the Winograd kernel (round 5): the 64 + 64 + 56 branch convs 32 -> 32 @64x48 / 64 -> 64 @32x24 / 128 -> 128 @16x12
    # and layer1's four 3x3 64 -> 64 @64x48, in all three roles - eval forward, train forward + column sums, input gradient + BatchNorm backward
    from advmix_amd import ops as _ops_
    assert _ops_.WINO and lib.advmix_conv_wino_config(B, 64, 48, 32, 32) == 768 and lib.advmix_conv_wino_config(B, 32, 24, 64, 64) == 384 \
        and lib.advmix_conv_wino_config(B, 16, 12, 128, 128) == 256
    w0 = _ops_.COUNTERS.get('wino', 0)
    s0 = _ops_.COUNTERS.get('smap', 0)
    p0 = _ops_.COUNTERS.get('pw', 0)
    mD.eval()
    with torch.no_grad():
        ye = mD(views[0].cuda())
        ye_ref = posenet_forward(net, D, views[0], extra, False)
    assert _ops_.COUNTERS.get('wino', 0) - w0 == 188, _ops_.COUNTERS
    # ... and stage 4's 24 convs 256 -> 256 @8x6 on the image-per-workgroup kernel (csrc/conv_smap.hip, 256 workgroups)
    assert _ops_.SMAP and lib.advmix_conv_smap_config(B, 8, 6, 256, 256) == 256 and _ops_.COUNTERS.get('smap', 0) - s0 == 24, _ops_.COUNTERS
    # ... and layer1's 1x1 convs 64 -> 256 (four blocks' last conv + the shortcut) on the streaming kernel (csrc/conv_pw.hip, 768 workgroups)
    assert _ops_.PW and lib.advmix_conv_pw_config(B, 64, 48, 64, 256) == 768 and _ops_.COUNTERS.get('pw', 0) - p0 == 5, _ops_.COUNTERS
    assert_close('eval out', ye, ye_ref, report=rep)
    assert_close('eval out vs golden', strided(ye.cpu().contiguous()), g[tag + '.eval_out'], report=rep)
    mD.train()                                            # train forward + FULL backward at the benchmarked tiles
    from oracle.posenet import trainable as _trainable
    x = views[1].cuda().requires_grad_(True)
    yt = mD(x)
    loss = JointsMSELoss(True)(yt, tgt.cuda(), tw.cuda())
    m0 = _ops_.COUNTERS.get('wgrad_multi', 0)
    loss.backward()
    assert _ops_.COUNTERS.get('wino', 0) - w0 == 3 * 188, _ops_.COUNTERS     # + train forward + input gradients
    assert _ops_.COUNTERS.get('smap', 0) - s0 == 3 * 24, _ops_.COUNTERS
    assert _ops_.COUNTERS.get('pw', 0) - p0 == 5 + 5 + 3, _ops_.COUNTERS       # + train forward + the input gradients of three 256 -> 64 convs
    # the small weight gradients of the fuse layers / transitions went out as mixed launches, none is left parked
    assert _ops_.COUNTERS.get('wgrad_multi', 0) - m0 >= 4 and not _ops_._WG_SMALL, (_ops_.COUNTERS, dict(_ops_._WG_SMALL))
    names = _trainable(D)
    for k in names:
        D[k].requires_grad_(True)
    xr = views[1].clone().requires_grad_(True)
    yr = posenet_forward(net, D, xr, extra, True)
    lr = joints_loss(yr, tgt, tw, True)
    g32 = dict(zip(names + ['x'], torch.autograd.grad(lr, [D[k] for k in names] + [xr])))
    for k in names:
        D[k].requires_grad_(False)
    assert_close('train out', yt, yr, report=rep)
    assert_close('train out vs golden', strided(yt.detach().cpu().contiguous()), g[tag + '.train_out'], report=rep)
    assert_close('loss', loss.detach(), lr.detach(), 1e-4, report=rep)
    assert_close('loss vs golden', [float(loss)], g[tag + '.loss'], 1e-4, report=rep)
    assert_close('dx vs golden', strided(x.grad.cpu().contiguous()), g[tag + '.dx'], 1e-6 + 2e-3 * float(np.abs(g[tag + '.dx']).max()))
    # D gradients at B = 32 against the fp64 oracle: statistically no worse than the fp32 oracle's own error, and the
    # reference's gradient checksums (sum / abs-sum of ten tensors) within 2e-3 of the abs-sum
    g64 = _f64_grads(net, extra, D, names, views[1], tgt, tw, B, J)
    got = {k: p.grad.detach().cpu() for k, p in mD.named_parameters()}
    got['x'] = x.grad.cpu()
    stats = assert_grads('D grads at B = 32', names + ['x'], got, g32, g64)
    print(tag, 'D-grad median rel err hip %.2e fp32-oracle %.2e outliers %d' % stats)
    for key in g.files:
        if key.startswith(tag + '.grad.'):
            k = key[len(tag) + 6:]
            s_, a_ = g[key]
            gs, ga = float(got[k].double().sum()), float(got[k].double().abs().sum())
            # (the oracle is held to 2e-3 of the abs-sum; the worst device value observed is 2.3e-3, on bn1.weight - the
            #  first layer's gradient carries the rounding of the whole network)
            assert abs(ga - a_) <= 5e-3 * a_ and abs(gs - s_) <= 5e-3 * a_, (k, gs, ga, s_, a_)
    del mD, mG, mT, ye, yt, g64, g32, got

    ga = gold_npz('b32_advmix_steps.npz')
    from oracle.posenet import trainable
    from oracle.step import Adam, advmix_step as ostep
    D, T, G = build_states(net, extra, J, salt=10)
    calib = synth_batch(tag + '.calib', B, J, H, W)[0][0]
    calibrate(net, T, calib, extra)
    calibrate(net, D, calib, extra)
    cfg, mD, mG, mT = product_models(net, extra, J, D, T, G)
    optD, optG = get_optimizer(cfg, mD), get_optimizer(cfg, mG)
    oD, oG = Adam(D, trainable(D)), Adam(G, list(G))
    mD.train(); mG.train(); mT.eval()
    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    v, t, w = synth_batch(tag + '.it0', B, J, H, W)
    f0, g0, t0 = _ops_.COUNTERS.get('w4', 0), _ops_.COUNTERS.get('w4_wgrad', 0), _ops_.COUNTERS.get('w4t', 0)
    loss_D, out = advmix_step(args, mD, mG, mT, JointsMSELoss(True), optD, optG,
                              [x.cuda().contiguous() for x in v], t.cuda(), w.cuda())
    # ... and the U-Net's 4x4 / stride-2 convs in the Winograd domain (csrc/conv_wino4.hip): in the generator's forward the four
    # down convs 64 -> 128 ... 512 -> 512 with >= 128 tiles, in its backward the input gradients of the four transposed convs
    # 1024 -> 512 ... 256 -> 64, and the six weight gradients with >= 512 tiles; the transposed form: the four transposed convs'
    # forward and the four down convs' input gradients with >= 192 tiles
    assert _ops_.WINO4 and _ops_.WINO4_WGRAD and _ops_.WINO4_T
    assert _ops_.COUNTERS.get('w4', 0) - f0 == 8 and _ops_.COUNTERS.get('w4_wgrad', 0) - g0 == 6 and _ops_.COUNTERS.get('w4t', 0) - t0 == 8, _ops_.COUNTERS
    want = ga[tag + '.losses'][0]
    assert_close('loss_D vs golden', [float(loss_D)], [0.9 * want[0] + 0.1 * want[1]], report=rep)
    # out2 = D(tmp) AFTER the first Adam update: teacher-forced against the oracle (it adopts the device's updated D,
    # see test_advmix_and_plain_steps_vs_oracle_and_golden), then the strict element-wise bound applies
    ref = ostep(net, extra, D, G, T, oD, oG, v, t, w, alpha=0.1, after_D_step=lambda: pull_params(mD, D))
    assert_close('loss_D', loss_D, ref['loss_D'], report=rep)
    assert_close('out2', out, ref['out2'], report=rep)
    print(tag, 'worst err/bound ratios', {k: round(x, 3) for k, x in rep.items()})


PINNED_TOL_NET = 3e-4   # the same through all ~290 convs of HRNet-W32 (rounding compounds with depth): observed 8.2e-5, beside the fp32 CPU
                        # arithmetic's own worst under the same masks (printed; profiles/r05zt_*); heat-map bounds are 1e-3
PINNED_TOL = 2e-5    # of the fp64 tensor's max; observed <= 2.8e-6 (512x512, B = 2) / see the printed worst ratio (profiles/r05zq_*)


@pytest.mark.parametrize('wino4_t', [False, True])
@pytest.mark.parametrize('case', [('512x512 B2', 512, 512, 2, 6, (6, 4, 6)), ('256x192 B32', 256, 192, 32, 6, (8, 6, 8))])   # (.., launches of csrc/conv_wino4.hip: forward form, weight gradients, transposed form)
def test_generator_gradients_with_pinned_masks(case, wino4_t, monkeypatch):
    """Round 5 (profiles/EXPERIMENTS.md K2): the generator's forward + backward, launch for launch, against fp64 - EVERY
    intermediate value, EVERY intermediate gradient and EVERY parameter gradient element-wise within 2e-5 of the tensor's
    scale (heat-map-style bounds are 1e-3), at the 512x512 / B = 2 shapes and at the benchmarked batch (where the Winograd-
    domain kernels of csrc/conv_wino4.hip take the 4x4 / stride-2 convs, forward form and weight gradients; with
    ``wino4_t`` the transposed form too; both settings, whatever the default).  The fp64 evaluation takes its ReLU / LeakyReLU
    MASKS from the device's activations (tests/unet_functional.py): an InstanceNorm output within rounding of zero takes
    either side of a ReLU in any fp32 evaluation and moves a weight gradient by 1e-2 of its scale when it carries a large
    gradient - the reason the un-pinned criteria above are statistical (a median against the fp32 oracle's own error).  Pinned,
    nothing statistical is left: this is the arithmetic of Conv2d / InstanceNorm / ConvTranspose2d / cat + ReLU and their
    backward kernels (Unet_generator.py:39-88)."""
    from oracle import detinit, configs
    from advmix_amd import ops
    from unet_functional import run
    name, H, W, B, downs, want = case
    monkeypatch.setattr(ops, 'WINO4_T', wino4_t)
    _, _, G = build_states('pose_hrnet', configs.HRNET_W32, 17, unet_downs=downs)
    x0 = torch.cat([detinit.normal('pinned.%s.view%d' % (name, k), (B, 3, H, W)) for k in range(3)], 1)
    proj = detinit.normal('pinned.%s.gproj' % name, (B, 3, H, W))
    c0 = dict(ops.COUNTERS)
    vh, gh, ph, order = run('hip', G, x0, proj, downs)
    took = {k: ops.COUNTERS.get(k, 0) - c0.get(k, 0) for k in ('w4', 'w4_wgrad', 'w4t')}
    assert (took['w4'], took['w4_wgrad'], took['w4t']) == (want[0], want[1], want[2] if wino4_t else 0), took
    v64, g64, p64, _ = run('f64', G, x0, proj, downs, pin={n: t.float() for n, t in vh.items()})
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    worst = {}
    for n in order:
        worst['value ' + n] = rel(vh[n], v64[n])
        worst['grad ' + n] = rel(gh[n], g64[n])
    gmax = max(float(g.abs().max()) for g in p64.values())
    for k in G:
        if float(p64[k].abs().max()) > 1e-6 * gmax:
            worst['param ' + k] = rel(ph[k], p64[k])
        else:                                               # a bias under an InstanceNorm: true gradient 0, the device's is rounding noise
            assert float(ph[k].abs().max()) <= 1e-4 * gmax, (k, float(ph[k].abs().max()), gmax)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print(name, 'wino4_t', wino4_t, 'launches', took, 'worst of %d tensors:' % len(worst), [(k, '%.2e' % v) for k, v in top])
    assert top[0][1] <= PINNED_TOL, top


@pytest.mark.parametrize('case', [(32, 32, 64, 48), (32, 64, 32, 24), (32, 128, 16, 12), (32, 256, 8, 6), (2, 32, 64, 48)])
def test_branch_of_basic_blocks_with_pinned_masks(case):
    """The dominant path of the step element-wise (round 5): one HRNet branch - four BasicBlocks, eight 3x3 convs each
    followed by a train-mode BatchNorm, residual adds and ReLUs (pose_hrnet.py:22-57, 247-265) - forward and backward through
    the functional ops at the benchmarked batch, against fp64 with the ReLU masks taken from the device's signs (see
    test_generator_gradients_with_pinned_masks): every block output, its gradient, and every weight / gamma / beta gradient
    within 2e-5 of the tensor's scale.  At B = 32 these are the launches of csrc/conv_wino.hip (32 / 64 / 128 channels) and
    csrc/conv_smap.hip (256 channels @8x6) with the BatchNorm column sums in the forward's epilogue, norm_apply_slots,
    the input gradient + BatchNorm-backward epilogues (mask / sign-from-c) and the weight gradients; the B = 2 case takes the
    direct kernels."""
    import torch.nn.functional as F
    from oracle import detinit
    from advmix_amd import ops
    B, C, H, W = case
    tag = 'branch.%d.%d.%d' % (B, C, H)
    x0 = detinit.normal(tag + '.x', (B, C, H, W))
    proj = detinit.normal(tag + '.proj', (B, C, H, W))
    Wt = [detinit.normal(tag + '.w%d' % i, (C, C, 3, 3), std=(9 * C) ** -0.5) for i in range(8)]
    Ga = [detinit.normal(tag + '.g%d' % i, (C,), std=0.2, mean=1.0) for i in range(8)]
    Be = [detinit.normal(tag + '.b%d' % i, (C,), std=0.2) for i in range(8)]

    def net(x, Wt, Ga, Be, conv_bn):
        outs = []
        for blk in range(4):
            t = conv_bn(x, Wt[2 * blk], Ga[2 * blk], Be[2 * blk], None, 'b%d.relu1' % blk)
            x = conv_bn(t, Wt[2 * blk + 1], Ga[2 * blk + 1], Be[2 * blk + 1], x, 'b%d.out' % blk)
            for n, v in (('b%d.relu1' % blk, t), ('b%d.out' % blk, x)):
                v.retain_grad()
                outs.append((n, v))
        return x, outs

    # the device
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)
    Wd = [torch.nn.Parameter(cl(w)) for w in Wt]
    Gd, Bd = [torch.nn.Parameter(g.cuda()) for g in Ga], [torch.nn.Parameter(b.cuda()) for b in Be]
    bank = ops.WinoBank(Wd)
    bank.refresh()
    stats = lambda: (torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros((), dtype=torch.int64, device='cuda'))
    c0 = dict(ops.COUNTERS)
    xd = cl(x0).requires_grad_(True)
    yd, rec_d = net(xd, Wd, Gd, Bd, lambda x, w, g, b, res, n: ops.conv_bn(x, w, g, b, *stats(), res, 1, 1, ops.ACT_RELU, True))
    (yd * proj.cuda()).sum().backward()
    torch.cuda.synchronize()
    took = {k: v - c0.get(k, 0) for k, v in ops.COUNTERS.items() if v != c0.get(k, 0)}
    pin = {n: v.detach().cpu() for n, v in rec_d}
    # fp64, masks pinned to the device's signs
    W6 = [w.double().requires_grad_(True) for w in Wt]
    G6, B6 = [g.double().requires_grad_(True) for g in Ga], [b.double().requires_grad_(True) for b in Be]
    x6 = x0.double().requires_grad_(True)

    def conv_bn64(x, w, g, b, res, n):
        pre = F.batch_norm(F.conv2d(x, w, None, 1, 1), None, None, g, b, True, 0.1, 1e-5)
        return (pre if res is None else pre + res) * (pin[n] > 0).double()
    y6, rec_6 = net(x6, W6, G6, B6, conv_bn64)
    (y6 * proj.double()).sum().backward()
    rel = lambda a, b: float((a.detach().double().cpu() - b).abs().max() / b.abs().max())
    worst = {'grad x': rel(xd.grad, x6.grad)}
    for (n, vd), (_, v6) in zip(rec_d, rec_6):
        worst['value ' + n] = rel(vd, v6.detach())
        worst['grad ' + n] = rel(vd.grad, v6.grad)
    for i in range(8):
        worst['dw%d' % i], worst['dgamma%d' % i], worst['dbeta%d' % i] = rel(Wd[i].grad, W6[i].grad), rel(Gd[i].grad, G6[i].grad), rel(Bd[i].grad, B6[i].grad)
    bank.release()
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print(case, 'launches', took, 'worst of %d tensors:' % len(worst), [(k, '%.2e' % v) for k, v in top])
    assert top[0][1] <= PINNED_TOL, top


def _activated_slots(P):
    """Slots of Plan ``P`` that an activation produces (their signs are what a pinned-mask evaluation needs)."""
    from advmix_amd import ops
    out = []
    for st in P.steps:
        k = st[0]
        act, d_ = {'bn': (st[5], st[3]) if k == 'bn' else None, 'fuse': (st[4], st[3]) if k == 'fuse' else None,
                   'inorm': (st[3], st[2]) if k == 'inorm' else None, 'act': (st[3], st[2]) if k == 'act' else None,
                   'catact': (st[4], st[3]) if k == 'catact' else None}.get(k) or (ops.ACT_NONE, None)
        if act != ops.ACT_NONE:
            out.append(d_)
    return out


def _pinned_plan_check(P, taps, shape, tag, fp32_too=False):
    """Runs Plan ``P`` through plan.PlanNet (train mode) on a seeded input of ``shape``, backward from a seeded projection of
    the output, and the SAME plan through a small fp64 interpreter of its steps (conv / bn / fuse with shift 0) whose ReLU masks
    are the device's signs at ``taps`` - which must be every slot a ReLU produces, each also reaching the output through fuse
    sums so that the executor hands it out.  Returns ({tensor: error relative to the fp64 tensor's max}, launch counters)."""
    import torch.nn.functional as F
    from oracle import detinit
    from advmix_amd import ops
    from advmix_amd.plan import PlanNet
    relu_slots = set(_activated_slots(P))
    assert relu_slots == set(taps) and all(st[0] in ('conv', 'deconv', 'bn', 'fuse', 'inorm', 'act', 'catact') for st in P.steps), \
        (sorted(relu_slots), sorted(taps))
    net = PlanNet(P)
    init = {}
    for n, p_ in net.named_parameters():
        if p_.dim() == 4:
            init[n] = detinit.normal(tag + '.' + n, tuple(p_.shape), std=(p_.shape[1] * p_.shape[2] * p_.shape[3]) ** -0.5)
        else:
            init[n] = detinit.normal(tag + '.' + n, tuple(p_.shape), std=0.2, mean=1.0 if n.endswith('.weight') else 0.0)
        with torch.no_grad():
            p_.copy_(init[n])
    net = net.cuda().train()
    x0 = detinit.normal(tag + '.x', shape)
    c0 = dict(ops.COUNTERS)
    xd = x0.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    run = net.begin(xd)
    seen = {}
    while not run.done:
        for s_ in taps:                                     # (chain outputs, alive until the sums have read them)
            if torch.is_tensor(run.slots[s_]) and s_ not in seen:
                seen[s_] = run.slots[s_]
        run.consume(ops.run_group(run.members()))
    assert len(seen) == len(taps), (sorted(seen), sorted(taps))
    yd = run.result
    proj = detinit.normal(tag + '.proj', tuple(yd.shape))
    pin = {s_: v.detach().cpu() for s_, v in seen.items()}
    (yd * proj.cuda()).sum().backward()
    torch.cuda.synchronize()
    took = {k: v - c0.get(k, 0) for k, v in ops.COUNTERS.items() if v != c0.get(k, 0)}
    # fp64 (and, for scale, the fp32 CPU arithmetic): the plan's steps, masks pinned to the device's signs
    def interpret(dt):
        Pm = {n: v.to(dt).requires_grad_(True) for n, v in init.items()}
        xm = x0.to(dt).requires_grad_(True)
        val = {0: xm}

        def A(pre, d_, act):                                  # the activation with the DEVICE's mask (ReLU; LeakyReLU(0.2))
            if act == ops.ACT_NONE:
                return pre
            return pre * (pin[d_] > 0).to(dt) if act == ops.ACT_RELU else torch.where(pin[d_] > 0, pre, pre * 0.2)
        for st in P.steps:
            if st[0] in ('conv', 'deconv'):
                _, name, s_, d_, stride, pad, hb = st
                val[d_] = (F.conv2d if st[0] == 'conv' else F.conv_transpose2d)(val[s_], Pm[name + '.weight'],
                                                                                 Pm[name + '.bias'] if hb else None, stride, pad)
            elif st[0] == 'bn':
                _, name, s_, d_, res, act = st
                pre = F.batch_norm(val[s_], None, None, Pm[name + '.weight'], Pm[name + '.bias'], True, 0.1, 1e-5)
                val[d_] = A(pre if res is None else pre + val[res], d_, act)
            elif st[0] == 'inorm':
                val[st[2]] = A(F.instance_norm(val[st[1]], eps=1e-5), st[2], st[3])
            elif st[0] == 'act':
                val[st[2]] = A(val[st[1]], st[2], st[3])
            elif st[0] == 'catact':
                val[st[3]] = A(torch.cat([val[st[1]], val[st[2]]], 1), st[3], st[4])
            else:
                _, xs, shifts, d_, act = st                   # pose_hrnet.py:254-265: nearest up-sampling by 2^shift, sum, ReLU
                pre = sum(val[s_] if sh == 0 else F.interpolate(val[s_], scale_factor=2 ** sh, mode='nearest') for s_, sh in zip(xs, shifts))
                val[d_] = A(pre, d_, act)
        out = val[P.out]
        (out * proj.to(dt)).sum().backward()
        return out.detach(), xm.grad, {n: v.grad for n, v in Pm.items()}
    y6, gx6, g6 = interpret(torch.float64)
    rel = lambda a, b: float((a.detach().double().cpu() - b).abs().max() / b.abs().max())
    worst = {'out': rel(yd, y6), 'grad x': rel(xd.grad, gx6)}
    gmax = max(float(g.abs().max()) for g in g6.values())
    live = [n for n in g6 if float(g6[n].abs().max()) > 1e-6 * gmax]
    for n, p_ in net.named_parameters():
        if n in live:
            worst['d ' + n] = rel(p_.grad, g6[n])
        else:                                               # a bias under an InstanceNorm: true gradient 0, the device's is rounding noise
            assert float(p_.grad.abs().max()) <= 1e-4 * gmax, (n, float(p_.grad.abs().max()), gmax)
    if fp32_too:
        y3, gx3, g3 = interpret(torch.float32)
        took['fp32 CPU arithmetic, same masks: worst'] = '%.2e' % max([rel(y3, y6), rel(gx3, gx6)] + [rel(g3[n], g6[n]) for n in live])
    return worst, took


@pytest.mark.parametrize('case', [(32, 32, 64, 48), (32, 64, 32, 24), (32, 128, 16, 12), (32, 256, 8, 6)])
def test_branch_through_the_plan_executor_with_pinned_masks(case):
    """The same branch of four BasicBlocks through the PRODUCT's executor (plan.PlanNet: the launch chain as one autograd node,
    the BatchNorm-backward sums in the input-gradient epilogues, a branch's eight weight gradients as one grouped launch)
    instead of op by op.  To see the activations whose signs pin the fp64 masks, every ReLU output is also summed into the
    network's output (three fuse sums), so each is a chain output - and receives a direct gradient besides the one through
    the blocks, which the fp64 graph mirrors.  Output, input gradient and all 24 parameter gradients within 2e-5 of scale."""
    from advmix_amd import ops
    from advmix_amd.plan import Plan
    B, C, H, W = case
    P = Plan(C)
    P.tag = 'branch'
    x, taps = 0, []
    for k in range(4):
        t = P.conv_bn(x, 'b%d.conv1' % k, 'b%d.bn1' % k, C, 3, 1, 1, ops.ACT_RELU)
        x = P.bn(P.conv(t, 'b%d.conv2' % k, C, 3, 1, 1), 'b%d.bn2' % k, ops.ACT_RELU, x)
        taps += [t, x]
    P.tag = None
    P.out = P.fuse([P.fuse(taps[:4], [0] * 4, ops.ACT_NONE), P.fuse(taps[4:], [0] * 4, ops.ACT_NONE)], [0, 0], ops.ACT_NONE)
    worst, took = _pinned_plan_check(P, taps, (B, C, H, W), 'planbranch.%d.%d.%d' % (B, C, H))
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print(case, 'launches', took, 'worst of %d tensors:' % len(worst), [(k, '%.2e' % v) for k, v in top])
    assert top[0][1] <= PINNED_TOL, top
    # ... and it was the product's path: Winograd / small-map kernels forward and backward, BatchNorm backward in the input-gradient
    # epilogues, the eight weight gradients as one grouped launch
    assert took.get('wino', 0) + took.get('smap', 0) == 16 and took.get('bnb', 0) >= 7, took
    assert took.get('wgrad_wino', 0) + took.get('wgrad_group', 0) >= 1, took


@pytest.mark.parametrize('case', [(32, 32, 64, 48), (32, 64, 32, 24), (32, 128, 16, 12), (16, 64, 64, 48), (32, 48, 24, 18)])
def test_batch_norm_applied_by_the_reading_conv_on_load(case, monkeypatch):
    """Round 6: for the inner edge conv1 -> bn1 -> relu -> conv2 of a BasicBlock (pose_hrnet.py:41-57) the
    product no longer launches norm_apply_slots nor writes the activation: conv2's Winograd kernel applies BatchNorm + ReLU while
    it stages its input (advmix_conv3x3_wino_fwd_inbn), derives and publishes bn1's batch statistics and running statistics on the
    way, conv2's weight gradient applies it again while staging (advmix_conv3x3_wgrad_wino_group_bn), bn1's backward takes the sign
    of the activation from c.  A branch of four BasicBlocks WITHOUT any tap in the graph (a second reader would switch the fusion
    off) through plan.PlanNet: (a) against the same network with ADVMIX_INBN off - the forward is the same arithmetic (output
    equal to rounding of the statistics' atomics), running statistics and counters equal, gradients to atomics' rounding; (b)
    element-wise against fp64 with the device's own masks read through ops.SLOT_TAP.  The last case is an HRNet-W48 width (48
    channels: the kernel's general staging path)."""
    from oracle import detinit
    from advmix_amd import ops
    from advmix_amd.plan import Plan, PlanNet
    from plan_functional import SlotRecorder, interpret, pins_of
    B, C, H, W = case
    P = Plan(C)
    P.tag = 'branch'
    x = 0
    for k in range(4):
        x = P.block('BASIC', x, 'b%d' % k, C)
    P.out = x
    tag = 'inbn.%d.%d.%d' % (B, C, H)
    x0 = detinit.normal(tag + '.x', (B, C, H, W))
    runs = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, 'INBN', fused)
        net = PlanNet(P)
        init = {}
        for n, p_ in net.named_parameters():
            init[n] = detinit.normal(tag + '.' + n, tuple(p_.shape), std=(p_.shape[1] * 9) ** -0.5) if p_.dim() == 4 else \
                detinit.normal(tag + '.' + n, tuple(p_.shape), std=0.2, mean=1.0 if n.endswith('.weight') else 0.0)
            with torch.no_grad():
                p_.copy_(init[n])
        net = net.cuda().train()
        c0 = dict(ops.COUNTERS)
        xd = x0.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        with SlotRecorder() as rec:
            yd = net(xd)
        proj = detinit.normal(tag + '.proj', tuple(yd.shape))
        (yd * proj.cuda()).sum().backward()
        torch.cuda.synchronize()
        took = {k: v - c0.get(k, 0) for k, v in ops.COUNTERS.items() if v != c0.get(k, 0)}
        runs[fused] = dict(y=yd.detach().cpu(), gx=xd.grad.detach().cpu(), g={n: p_.grad.detach().cpu() for n, p_ in net.named_parameters()},
                           buf={n: b_.detach().cpu().clone() for n, b_ in net.named_buffers()}, took=took, slots=rec.of(net), init=init)
    a, b = runs[True], runs[False]
    print(case, 'fused launches', a['took'], '| unfused', b['took'])
    if C % 32 == 0:
        assert a['took'].get('inbn', 0) == 4 and a['took'].get('wgrad_wino_bn', 0) == 4 and not a['took'].get('inbn_materialized'), a['took']
    else:       # 48 channels: no Winograd weight gradient (channel pairs of 32) - the backward pass materialises the four activations
        assert a['took'].get('inbn', 0) == 4 and not a['took'].get('wgrad_wino_bn') and a['took'].get('inbn_materialized') == 4, a['took']
    assert not b['took'].get('inbn') and a['took'].get('wino') == b['took'].get('wino') == 16, (a['took'], b['took'])
    # (a) the same network with and without the fusion
    sc = float(b['y'].abs().max())
    assert float((a['y'] - b['y']).abs().max()) <= 2e-6 * sc, float((a['y'] - b['y']).abs().max()) / sc
    for n in b['buf']:                                       # bn1's running statistics and counters came from conv2's workgroup (0, 0)
        assert torch.allclose(a['buf'][n].double(), b['buf'][n].double(), rtol=1e-6, atol=1e-7), n
    assert float((a['gx'] - b['gx']).abs().max()) <= 1e-4 * float(b['gx'].abs().max())
    for n in b['g']:
        assert float((a['g'][n] - b['g'][n]).abs().max()) <= 1e-4 * float(b['g'][n].abs().max()) + 1e-9, n
    # (b) the fused run element-wise against fp64 with its own masks
    (slots,) = a['slots']
    pin, pool = pins_of(P, slots)
    W64 = {n: v.double().requires_grad_(True) for n, v in a['init'].items()}
    x64 = x0.double().requires_grad_(True)
    y64 = interpret(P, W64, x64, pin=pin, pool_src=pool)[P.out]
    proj = detinit.normal(tag + '.proj', tuple(y64.shape)).double()
    names = list(W64)
    g64 = dict(zip(names + ['x'], torch.autograd.grad((y64 * proj).sum(), [W64[n] for n in names] + [x64])))
    got = dict(a['g'], x=a['gx'])
    rel = _pinned_rel_errors(got, g64, names + ['x'])
    rel['out'] = float((a['y'].double() - y64.detach()).abs().max() / y64.abs().max())
    top = sorted(rel.items(), key=lambda kv: -kv[1])[:3]
    print(case, 'fused run vs fp64 with its own masks: worst of %d tensors' % len(rel), [(k, '%.2e' % v) for k, v in top])
    assert top[0][1] <= PINNED_TOL, top


def test_layer1_bottlenecks_through_the_plan_executor_with_pinned_masks():
    """HRNet's layer1 (pose_hrnet.py:59-98, 286: four Bottlenecks 64 -> 256 @64x48, the first with its 1x1 shortcut conv) at the
    benchmarked batch the same way: the streaming 1x1 kernel of csrc/conv_pw.hip (64 -> 256: forward + sums, and the input
    gradients of the 256 -> 64 convs), the direct kernel's 256 -> 64 1x1 convs, the Winograd kernel's 64 -> 64 @64x48, BatchNorm
    over 256 channels.  The 64-channel ReLU outputs reach the output through one extra 1x1 conv (64 -> 256) on their sum."""
    from advmix_amd import ops
    from advmix_amd.plan import Plan
    B, H, W = 32, 64, 48
    P = Plan(64)
    P.tag = 'layer1'
    x, narrow, wide = 0, [], []
    for k in range(4):
        t1 = P.conv_bn(x, 'l%d.conv1' % k, 'l%d.bn1' % k, 64, 1, 1, 0, ops.ACT_RELU)
        t2 = P.conv_bn(t1, 'l%d.conv2' % k, 'l%d.bn2' % k, 64, 3, 1, 1, ops.ACT_RELU)
        o = P.conv(t2, 'l%d.conv3' % k, 256, 1, 1, 0)
        res = x if P.ch[x] == 256 else P.conv_bn(x, 'l%d.downsample.0' % k, 'l%d.downsample.1' % k, 256, 1, 1, 0, ops.ACT_NONE)
        x = P.bn(o, 'l%d.bn3' % k, ops.ACT_RELU, res)
        narrow += [t1, t2]
        wide.append(x)
    P.tag = None
    n64 = P.fuse([P.fuse(narrow[:4], [0] * 4, ops.ACT_NONE), P.fuse(narrow[4:], [0] * 4, ops.ACT_NONE)], [0, 0], ops.ACT_NONE)
    P.out = P.fuse([P.fuse(wide, [0] * 4, ops.ACT_NONE), P.conv(n64, 'tap', 256, 1, 1, 0)], [0, 0], ops.ACT_NONE)
    worst, took = _pinned_plan_check(P, narrow + wide, (B, 64, H, W), 'planlayer1')
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print('layer1', 'launches', took, 'worst of %d tensors:' % len(worst), [(k, '%.2e' % v) for k, v in top])
    assert top[0][1] <= PINNED_TOL, top
    assert took.get('pw', 0) >= 5 and took.get('wino', 0) >= 8 and took.get('bnb', 0) >= 8, took


def _tap_every_relu(P, out_ch=32):
    """Extends Plan ``P`` so that every slot a ReLU produces also reaches the output: per shape the slots are summed by trees of
    fuse sums (<= 4 inputs each), every shape's sum goes through a 1x1 'tap' conv to ``out_ch`` channels and is up-sampled
    (fuse shifts) to the finest shape, and the network's own output joins the same way.  Returns the tapped slots."""
    from advmix_amd import ops
    scale = {0: 0}                                          # log2 of the down-sampling of each slot
    for st in P.steps:
        if st[0] in ('conv', 'deconv'):
            scale[st[3]] = scale[st[2]] + (st[4] - 1) * (1 if st[0] == 'conv' else -1)
        elif st[0] == 'bn':
            scale[st[3]] = scale[st[2]]
        elif st[0] in ('inorm', 'act'):
            scale[st[2]] = scale[st[1]]
        elif st[0] == 'catact':
            scale[st[3]] = scale[st[1]]
        else:
            scale[st[3]] = scale[st[1][st[2].index(0)]]
    taps = _activated_slots(P)
    P.tag = None
    groups = {}
    for s_ in taps + [P.out]:
        groups.setdefault((P.ch[s_], scale[s_]), []).append(s_)

    def tree(xs, shifts):
        while len(xs) > 1:
            nx, ns = [], []
            for i in range(0, len(xs), 4):
                part, ps = xs[i:i + 4], shifts[i:i + 4]
                if len(part) == 1:
                    nx.append(part[0]); ns.append(ps[0])
                else:
                    base = min(ps)                          # (the sum lives at its finest member's shape)
                    order = sorted(range(len(part)), key=lambda q: ps[q])
                    nx.append(P.fuse([part[q] for q in order], [ps[q] - base for q in order], ops.ACT_NONE)); ns.append(base)
            xs, shifts = nx, ns
        return xs[0]
    finest = min(sc for _, sc in groups)
    tops, shifts = [], []
    for gi, ((c, sc), slots) in enumerate(sorted(groups.items())):
        tops.append(P.conv(tree(slots, [0] * len(slots)), 'tap%d' % gi, out_ch, 1, 1, 0))
        shifts.append(sc - finest)
    P.out = tree(tops, shifts)
    return taps


@pytest.mark.parametrize('case', [('hrnet_w32', 16, 256, 192), ('hrnet_w48', 8, 384, 288)])
def test_whole_hrnet_through_the_plan_executor_with_pinned_masks(case):
    """The whole pose network element-wise (round 5): HRNet-W32 256x192 (pose_hrnet.py:270-500; plan.hrnet_plan's 316 steps) at
    B = 16 - every kernel family of the benchmarked step is reached from that batch on (asserted) - and HRNet-W48 384x288 (C4) at
    B = 8, through plan.PlanNet in
    train mode, forward and backward, against an fp64 evaluation of the same plan whose ReLU masks are the device's signs
    (every ReLU output is tapped into the output, _tap_every_relu): the output, the input gradient and every one of the
    network's parameter gradients within 3e-4 of the tensor's scale (PINNED_TOL_NET)."""
    from oracle import configs
    from advmix_amd import ops
    from advmix_amd._lib import lib
    from advmix_amd.plan import hrnet_plan
    name, B, H, W = case
    w32 = name == 'hrnet_w32'
    assert not w32 or (lib.advmix_conv_wino_config(B, 16, 12, 128, 128) >= ops.WINO_MIN_WGS and lib.advmix_conv_pw_config(B, 64, 48, 64, 256) >= ops.WINO_MIN_WGS
                       and lib.advmix_conv_smapw_config(B, 8, 6, 256, 256) >= ops.WINO_MIN_WGS)
    P = hrnet_plan(configs.HRNET_W32 if w32 else configs.HRNET_W48, 17)
    n_own = len(P.params)
    taps = _tap_every_relu(P)
    worst, took = _pinned_plan_check(P, taps, (B, 3, H, W), 'plan' + name, fp32_too=True)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print(name, 'B', B, 'taps', len(taps), 'own parameters', n_own, 'launches', took, 'worst of %d tensors:' % len(worst),
          [(k, '%.2e' % v) for k, v in top])
    assert top[0][1] <= PINNED_TOL_NET, top
    assert took.get('wino', 0) >= (2 * 188 if w32 else 100) and took.get('bnb', 0) >= 200 and took.get('wgrad_wino', 0) >= 1, took
    assert not w32 or (took.get('smap', 0) >= 2 * 24 and took.get('pw', 0) >= 8 and took.get('wgrad_multi', 0) >= 1), took


def test_generator_through_the_plan_executor_with_pinned_masks():
    """The augmentation generator as the PRODUCT runs it (plan.unet_plan: the in-place-LeakyReLU skip as norm + leaky, the parent's
    in-place ReLU as cat + relu, three launch chains) at the benchmarked batch, every activation tapped into the output: output,
    input gradient and all parameter gradients element-wise against fp64 with the device's masks; the Winograd-domain launches of
    csrc/conv_wino4.hip (forward form, transposed form, weight gradients) asserted."""
    from advmix_amd.plan import unet_plan
    P = unet_plan(9, 3, 6)
    taps = _tap_every_relu(P)
    worst, took = _pinned_plan_check(P, taps, (32, 9, 256, 192), 'planunet', fp32_too=True)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print('unet B 32 taps', len(taps), 'launches', took, 'worst of %d tensors:' % len(worst), [(k, '%.2e' % v) for k, v in top])
    assert top[0][1] <= PINNED_TOL, top
    assert took.get('w4', 0) == 8 and took.get('w4_wgrad', 0) == 6 and took.get('w4t', 0) == 8, took


@pytest.mark.parametrize('tag', ['hrnet_tiny', 'resnet18_tiny'])
def test_gradient_fan_in_by_separate_add_gives_the_same_gradients(tag):
    """with ADVMIX_FANIN=0 the gradients pending for a slot are NOT folded into the consumer's input-gradient
    epilogue, so that epilogue must not carry the producer's BatchNorm backward either (it would multiply a partial
    gradient by act'(y) and sum it).  Full and input-only backward, fan-in fused vs separate: same gradients, and the
    BatchNorm-backward epilogue still taken wherever the gradient is complete."""
    from oracle.posenet import calibrate
    from oracle.synth import synth_batch
    from advmix_amd import ops
    from advmix_amd.core.loss import JointsMSELoss
    from advmix_amd.core.function import set_require_grad
    net, extra, J, B, H, W, _ = CASES[tag]
    D, T, G = build_states(net, extra, J)
    views, tgt, tw = synth_batch(tag, B, J, H, W)
    calibrate(net, D, views[2], extra)
    cfg, mD, _, _ = product_models(net, extra, J, D, T, G)
    mD.train()
    res = {}
    try:
        for fused in (True, False):
            ops.FANIN_FUSED = fused
            for frozen in (False, True):
                set_require_grad(mD, not frozen)
                for p in mD.parameters():
                    p.grad = None
                x = views[1].cuda().requires_grad_(True)
                n0 = ops.COUNTERS['bnb']
                JointsMSELoss(True)(mD(x), tgt.cuda(), tw.cuda()).backward()
                g = {'x': x.grad.detach().clone()}
                if not frozen:
                    g.update({k: p.grad.detach().clone() for k, p in mD.named_parameters()})
                res[fused, frozen] = (g, ops.COUNTERS['bnb'] - n0)
    finally:
        ops.FANIN_FUSED = True
        set_require_grad(mD, True)
    for frozen in (False, True):
        (ga, na), (gb, nb) = res[True, frozen], res[False, frozen]
        assert 0 <= nb <= na and (na > 0 or tag != 'hrnet_tiny'), (na, nb)     # (resnet18_tiny's narrow layers: no epilogue)
        for k in ga:
            sc = float(ga[k].abs().max()) + 1e-30
            assert float((ga[k] - gb[k]).abs().max()) <= 2e-4 * sc, (k, frozen, float((ga[k] - gb[k]).abs().max()), sc)


BENCH_TILE_CASES = {
    # tag: (net, extra, J, B, H, W, unet downs, the 3x3 / 1x1 stride-1 convs (C, H, W, k) whose tiles decide the case)
    'resnet50_b32': ('pose_resnet', 'RES50', 17, 32, 256, 192, 6,
                     [(64, 64, 48, 3), (128, 32, 24, 3), (256, 16, 12, 3), (512, 8, 6, 3), (256, 64, 48, 1), (2048, 8, 6, 1)]),
    'hrnet_w48_b16': ('pose_hrnet', 'HRNET_W48', 17, 16, 384, 288, 5,
                      [(48, 96, 72, 3), (96, 48, 36, 3), (192, 24, 18, 3), (384, 12, 9, 3)]),
    # C4 at its benchmarked batch itself (round 5, VERDICT r4 item 6 a): forward-only vectors from the real reference
    # (tests/golden/c4_b32_forward.npz, oracle/gen_golden.py::gen_c4b32 - its backward would not fit the build container)
    'hrnet_w48_b32': ('pose_hrnet', 'HRNET_W48', 17, 32, 384, 288, 5,
                      [(48, 96, 72, 3), (96, 48, 36, 3), (192, 24, 18, 3), (384, 12, 9, 3)]),
}


@pytest.mark.parametrize('tag', sorted(BENCH_TILE_CASES))
def test_every_benchmarked_network_at_its_benchmarked_tiles(tag):
    """ResNet-50 (C2) and HRNet-W48 384x288 (C4) are benchmarked at B = 32 but were parity-checked at B = 2,
    where conv_direct picks other tiles.  ResNet-50 at B = 32 itself; HRNet-W48 at B = 16, which reaches every tile
    configuration B = 32 does (asserted), and since round 5 at B = 32 itself - eval forward, train forward, loss and running
    statistics against the CPU oracle (full tensors) and the REAL reference's vectors at that batch
    (tests/golden/benchtiles_forward.npz, c4_b32_forward.npz)."""
    from oracle import configs
    from oracle.posenet import posenet_forward, calibrate
    from oracle.loss import joints_loss
    from oracle.synth import synth_batch, strided
    from advmix_amd._lib import lib
    from advmix_amd.core.loss import JointsMSELoss
    net, ename, J, B, H, W, downs, convs = BENCH_TILE_CASES[tag]
    extra = getattr(configs, ename)

    def tiles(b):
        return {lib.advmix_conv_direct_config(m, b, h, w, c, c, k, k, 1) for c, h, w, k in convs for m in (0, 1)}
    assert tiles(B) >= tiles(32) and -1 not in tiles(32), (tiles(B), tiles(32))
    g = gold_npz('c4_b32_forward.npz' if tag == 'hrnet_w48_b32' else 'benchtiles_forward.npz')
    D, T, G = build_states(net, extra, J, unet_downs=downs)
    views, tgt, tw = synth_batch(tag, B, J, H, W)
    calibrate(net, D, views[2], extra)
    cfg, mD, mG, mT = product_models(net, extra, J, D, T, G, downs=downs)
    rep = {}
    mD.eval()
    with torch.no_grad():
        ye = mD(views[0].cuda())
        ye_ref = posenet_forward(net, D, views[0], extra, False)
    assert_close('eval out', ye, ye_ref, report=rep)
    assert_close('eval out vs golden', strided(ye.cpu().contiguous()), g[tag + '.eval_out'], report=rep)
    mD.train()
    with torch.no_grad():
        yt = mD(views[1].cuda())
        loss = JointsMSELoss(True)(yt, tgt.cuda(), tw.cuda())
        yr = posenet_forward(net, D, views[1], extra, True)
        lr = joints_loss(yr, tgt, tw, True)
    assert_close('train out', yt, yr, report=rep)
    assert_close('train out vs golden', strided(yt.cpu().contiguous()), g[tag + '.train_out'], report=rep)
    assert_close('loss', loss, lr, 1e-4, report=rep)
    assert_close('loss vs golden', [float(loss)], g[tag + '.loss'], 1e-4, report=rep)
    sd = mD.state_dict()
    for key in g.files:
        if key.startswith(tag + '.bn.'):
            assert_close(key, sd[key[len(tag) + 4:]], g[key])
    print(tag, 'worst err/bound ratios', {k: round(x, 3) for k, x in rep.items()})


def test_c4_step_at_its_benchmarked_batch_against_the_real_reference():
    """C4's whole STEP was parity-tested at B = 2 only.  One iteration of the REAL train_advmix on
    HRNet-W48 384x288, B = 32, UnetGenerator(9, 3, 5) (tests/golden/c4_b32_advmix_steps.npz, oracle/gen_golden.py::gen_c4b32step:
    ~45 GB of host memory in the build container) against one advmix_step of the product from the same state and batch:
      * loss_D = 0.9 L(D(tmp), target) + 0.1 L(D(tmp), teacher) - a function of the generator's forward, the mix, the teacher and
        the student's first forward, all before any update - against the real reference's logged losses, to the heat-map bound;
      * out2 = D'(tmp), the student's output AFTER Adam's first step, and loss_D again, against the CPU oracle's step
        teacher-forced as in test_network_parity_at_the_benchmarked_batch (the oracle adopts the device's updated student: Adam's
        first step is lr * sign(g), so un-forced the two updated networks differ wherever a gradient is rounding noise - the
        reference's own out2 is 0.18 away at |ref| <= 1.6 - which measures that, not kernels).  The oracle's step needs ~45 GB
        of host memory: the GPU box's 300 GB limit has room (asserted up front, so that a smaller host skips instead of dying)."""
    import os
    from oracle import configs
    from oracle.posenet import calibrate, trainable
    from oracle.step import Adam, advmix_step as ostep
    from oracle.synth import synth_batch
    from advmix_amd.core.function import advmix_step
    from advmix_amd.core.loss import JointsMSELoss
    from advmix_amd.utils.utils import get_optimizer
    from advmix_amd import ops as _ops_
    try:
        with open('/sys/fs/cgroup/memory.max') as f:
            lim = f.read().strip()
        lim = int(lim) if lim.isdigit() else None
    except OSError:
        lim = None
    avail = os.sysconf('SC_PHYS_PAGES') * os.sysconf('SC_PAGE_SIZE')
    if min(avail, lim or avail) < 100 * (1 << 30):
        pytest.skip('the CPU oracle of this step needs ~45 GB of host memory; this host offers %.0f GB' % (min(avail, lim or avail) / 2 ** 30))
    tag, net, extra, J, B, H, W, downs = 'hrnet_w48_b32', 'pose_hrnet', configs.HRNET_W48, 17, 32, 384, 288, 5
    ga = gold_npz('c4_b32_advmix_steps.npz')
    D, T, G = build_states(net, extra, J, unet_downs=downs, salt=10)
    calib = synth_batch(tag + '.calib', B, J, H, W)[0][0]
    calibrate(net, T, calib, extra)
    calibrate(net, D, calib, extra)
    cfg, mD, mG, mT = product_models(net, extra, J, D, T, G, downs=downs)
    optD, optG = get_optimizer(cfg, mD), get_optimizer(cfg, mG)
    oD, oG = Adam(D, trainable(D)), Adam(G, list(G))
    mD.train(); mG.train(); mT.eval()
    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    v, t, w = synth_batch(tag + '.it0', B, J, H, W)
    c0 = dict(_ops_.COUNTERS)
    loss_D, out = advmix_step(args, mD, mG, mT, JointsMSELoss(True), optD, optG, [x.cuda().contiguous() for x in v], t.cuda(), w.cuda())
    torch.cuda.synchronize()
    took = {k: n - c0.get(k, 0) for k, n in _ops_.COUNTERS.items() if n != c0.get(k, 0)}
    print(tag, 'launch counters of the step', took)
    assert took.get('wino', 0) >= 600 and took.get('inbn', 0) >= 2 * 60, took      # the 48 / 96-channel branches on the Winograd kernel, BatchNorm on load
    want = ga[tag + '.losses'][0]                            # (heat-map loss, distillation loss, generator loss) of the iteration
    rep = {}
    assert_close('loss_D vs the real train_advmix', [float(loss_D)], [0.9 * want[0] + 0.1 * want[1]], report=rep)
    ref = ostep(net, extra, D, G, T, oD, oG, v, t, w, alpha=0.1, after_D_step=lambda: pull_params(mD, D), unet_kw={'num_downs': downs})
    assert_close('loss_D', loss_D, ref['loss_D'], report=rep)
    assert_close('out2', out, ref['out2'], report=rep)
    print(tag, 'loss_D %.6f (real reference %.6f, oracle %.6f); worst err/bound ratios' % (
        float(loss_D), 0.9 * want[0] + 0.1 * want[1], float(ref['loss_D'])), {k: round(x, 3) for k, x in rep.items()})


def _device_checksums(model, keys):
    sd = model.state_dict()
    return {k: [float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys}


def _checksum_units(got, want, numel, lr, updates):
    """Worst |difference| of the (sum, abs-sum) checksums in units of numel * lr * updates - half the largest difference an
    UN-forced run can show (every element moving the other way at every Adam update = 2) (Adam's first updates are
    ~lr * sign(g): elements whose gradient is rounding noise go either way)."""
    worst = 0.0
    for k, (s, a) in want.items():
        unit = numel[k] * lr * updates
        worst = max(worst, abs(got[k][0] - s) / unit, abs(got[k][1] - a) / unit)
    return worst


CHECKSUM_UNITS_OBSERVED = {'hrnet_tiny': 0.75, 'resnet18_tiny': 0.82, 'c1_resnet50_j16_b4': 0.082}


@pytest.mark.parametrize('tag', ['hrnet_tiny', 'resnet18_tiny'])
def test_unforced_advmix_loop_lands_on_the_reference_checksums(tag):
    """The post-step parameters of the HIP path, NOT teacher-forced, against the REAL reference's checksums after its
    2-3 train_advmix iterations (tests/golden/advmix_checksums.json).  What can hold un-forced: Adam's first updates
    are ~lr * sign(g), so an element whose gradient is rounding noise moves lr the other way in ANY other fp32
    implementation, and everything downstream (outputs, running statistics of later iterations) inherits that - the
    oracle reproduces the reference's checksums to 2e-3 only because it runs the same torch-CPU kernels.  The bound
    is therefore in units of numel * lr * updates (an element that steps -lr where the reference steps +lr differs by 2 lr
    per update: 2 = every element went the other way every time).  Observed (round
    3, three runs): D 0.04-0.27, C1's ResNet-50 0.08 - but the GENERATOR 0.03 in one run and 0.75-0.82 in the others
    (both tiny nets): its gradient through the frozen student is almost all rounding noise at these sizes, so un-forced
    its checksums CANNOT hold (VERDICT r2 item 4's alternative: this docstring says why).  Asserted: never more than the
    all-opposite bound of 2 - which still catches a doubled update or a wrong learning rate - and 3 x the observation
    where that is tighter (C1); the element-wise claims are the teacher-forced tests'."""
    from oracle.posenet import calibrate
    from oracle.synth import synth_batch
    from advmix_amd.core.function import advmix_step
    from advmix_amd.core.loss import JointsMSELoss
    from advmix_amd.utils.utils import get_optimizer
    net, extra, J, B, H, W, iters = CASES[tag]
    meta = gold_json('advmix_checksums.json')[tag]
    D, T, G = build_states(net, extra, J, salt=10)
    calib = synth_batch(tag + '.calib', B, J, H, W)[0][0]
    calibrate(net, T, calib, extra)
    calibrate(net, D, calib, extra)
    cfg, mD, mG, mT = product_models(net, extra, J, D, T, G)
    optD, optG = get_optimizer(cfg, mD), get_optimizer(cfg, mG)
    mD.train(); mG.train(); mT.eval()
    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    for it in range(iters):
        v, t, w = synth_batch('%s.it%d' % (tag, it), B, J, H, W)
        advmix_step(args, mD, mG, mT, JointsMSELoss(True), optD, optG, [x.cuda().contiguous() for x in v], t.cuda(), w.cuda())
    numel = {k: v.numel() for k, v in mD.state_dict().items()}
    params = {k: v for k, v in meta['D'].items() if 'running_' not in k}
    u = _checksum_units(_device_checksums(mD, params), params, numel, 1e-3, iters)
    numel_G = {k: v.numel() for k, v in mG.state_dict().items()}
    ug = _checksum_units(_device_checksums(mG, meta['G']), meta['G'], numel_G, 1e-3, iters)
    print(tag, 'un-forced checksum drift in units of numel*lr*updates: D %.4f G %.4f' % (u, ug))
    # D: never more than the all-opposite bound.  G (round 5): no un-forced bound at all - its gradient is held element-wise,
    # teacher-forced, against fp64 in test_advmix_and_plain_steps_vs_oracle_and_golden instead (the bound of 2 units that
    # stood here could only catch a skipped or doubled step); ug is printed for the record and only its sanity is asserted
    bound = min(2.0, 3 * CHECKSUM_UNITS_OBSERVED[tag])
    assert u <= bound and ug == ug and ug <= 2.0, (u, ug, bound)
    assert int(mD.state_dict()['bn1.num_batches_tracked']) == meta['nbt']


def test_c1_literally_plain_loop_j16_b4():
    """BASELINE.json configs[0] as written (pose_resnet50 256x192, MPII's 16 joints, batch 4, the plain ``train`` loop,
    function.py:30-95): iteration 0 against the REAL reference's loss and heat-maps, both iterations against the oracle
    (teacher-forced after each update; the oracle itself is pinned to both reference iterations in
    test_oracle_golden.py), then the reference's post-step parameter checksums in units of the Adam drift."""
    from oracle import configs
    from oracle.posenet import calibrate, trainable
    from oracle.step import Adam, plain_step as oplain
    from oracle.synth import synth_batch, strided
    from advmix_amd.core.function import plain_step
    from advmix_amd.core.loss import JointsMSELoss
    from advmix_amd.utils.utils import get_optimizer
    tag, net, extra, J, B, H, W = 'c1_resnet50_j16_b4', 'pose_resnet', configs.RES50, 16, 4, 256, 192
    g, meta = gold_npz('c1_plain_steps.npz'), gold_json('c1_plain_checksums.json')[tag]
    D, T, G = build_states(net, extra, J, salt=20)
    calib = synth_batch(tag + '.calib', B, J, H, W)[0][0]
    calibrate(net, D, calib, extra)
    cfg, mD, _, _ = product_models(net, extra, J, D, D, G)
    optD, oD = get_optimizer(cfg, mD), Adam(D, trainable(D))
    mD.train()
    crit = JointsMSELoss(True)
    for it in range(2):
        v, t, w = synth_batch('%s.plain%d' % (tag, it), B, J, H, W)
        loss, out = plain_step(mD, crit, optD, v[0].cuda(), t.cuda(), w.cuda())
        ref = oplain(net, extra, D, oD, v[0], t, w)
        frac = match_fraction(mD, D, 1e-5 if it == 0 else 1e-4)               # the oracle's own update vs the device's
        assert frac >= 0.9, frac
        pull_params(mD, D)
        assert_close('loss it%d' % it, loss, ref['loss'])
        assert_close('out it%d' % it, out, ref['out'])
        if it == 0:                                       # before any update: the reference's own numbers
            assert_close('loss vs golden', [float(loss)], [g[tag + '.plain_losses'][0]], 1e-4)
            assert_close('out vs golden', strided(out.cpu().contiguous(), 2048), g[tag + '.out.it0'])
    sd = mD.state_dict()
    assert int(sd['bn1.num_batches_tracked']) == meta['nbt']
    numel = {k: v.numel() for k, v in sd.items()}
    params = {k: v for k, v in meta['plain_D'].items() if 'running_' not in k}
    u = _checksum_units(_device_checksums(mD, params), params, numel, 1e-3, 2)
    print(tag, 'checksum drift vs the reference in units of numel*lr*updates: %.4f' % u)
    assert u <= min(2.0, 3 * CHECKSUM_UNITS_OBSERVED[tag]), u


def test_smoke_entry():
    run_smoke('hrnet_tiny', iters=2)
    run_smoke('resnet18_tiny', iters=1)


def test_nms_mirror_matches_reference_semantics():
    from advmix_amd.nms import nms as pn
    from oracle import nms as onms
    g = gold_json('nms.json')
    for name, c in g['box'].items():
        d = np.array(c['dets'], np.float32) if c['dets'] is not None else np.load(
            __import__('os').path.join(__import__('helpers').GOLD, 'nms_dets_%s.npy' % name))
        assert [int(i) for i in pn.nms(d, c['thresh'])] == c['keep'], name
        assert [int(i) for i in pn.gpu_nms(d, c['thresh'])] == c['keep'], name
        assert [int(i) for i in pn.cpu_nms(d, c['thresh'])] == c['keep'], name
    d = np.array([[0, 0, 9, 9, 0.9], [5, 0, 14, 9, 0.8]], np.float32)
    th = float(np.float32(50.0) / np.float32(150.0))
    for thr in (th, float(np.nextafter(th, 1.0)), float(np.nextafter(th, 0.0)), 0.3333, 0.5):   # Python floats
        assert [int(i) for i in pn.nms(d, thr)] == onms.py_nms(d, thr), thr
        assert [int(i) for i in pn.gpu_nms(d, thr)] == onms.gpu_nms(d, thr), thr
        assert [int(i) for i in pn.cpu_nms(d, thr)] == onms.cpu_nms(d, thr), thr
    assert pn.nms(d[:0], 0.5) == [] and pn.gpu_nms(d[:0], 0.5) == []
    for name, c in g['oks'].items():
        k = np.array(c['kpts'])
        db = [{'score': s, 'keypoints': kk, 'area': a} for s, kk, a in zip(c['score'], k, c['area'])]
        assert [int(i) for i in pn.oks_nms(db, c['thresh'])] == c['keep'], name
        assert [int(i) for i in pn.soft_oks_nms(db, c['thresh'])] == c['soft_keep'], name


# ---- validate() ------------------------------------------------------------------------------

VAL_CASES = {
    'hrnet_tiny': ('pose_hrnet', 'HRNET_TINY', 5, 3, 64, 64),
    'resnet18_tiny': ('pose_resnet', 'RES18_TINY', 5, 3, 64, 64),
    'hrnet_w32': ('pose_hrnet', 'HRNET_W32', 17, 2, 256, 192),
}


@pytest.mark.parametrize('tag', sorted(VAL_CASES))
@pytest.mark.parametrize('mode', ['plain', 'flip'])
def test_validate_loop_matches_reference_fixture(tag, mode):
    """The product ``validate`` (device flip test + device get_final_preds) on the batches the REAL
    reference's ``validate`` was run on: merged heat-maps / losses within 1e-3, all_boxes exact,
    maxvals within 1e-3, coordinates equal wherever the argmax did not flip on an fp32 near-tie, and
    - exactly - equal to the oracle's get_final_preds applied to the product's own heat-maps."""
    from oracle import configs, validate as oval
    from oracle.posenet import calibrate
    from oracle.synth import synth_batch, synth_boxes, strided
    from helpers import gold_npz, gold_json, build_states
    from smoke_step import product_models, assert_close
    from advmix_amd.core.function import validate
    from advmix_amd.core.loss import JointsMSELoss
    net, ename, J, B, H, W = VAL_CASES[tag]
    extra = getattr(configs, ename)
    g, meta = gold_npz('validate.npz'), gold_json('validate.json')
    key = '%s.%s' % (tag, mode)
    m = meta[key]
    flip = mode == 'flip'
    D_sd, T_sd, G_sd = build_states(net, extra, J, salt=30)
    calibrate(net, D_sd, synth_batch(tag + '.valcalib', B, J, H, W)[0][0], extra)
    cfg, D, _, _ = product_models(net, extra, J, D_sd, T_sd, G_sd)
    cfg.defrost() if hasattr(cfg, 'defrost') else None
    cfg['TEST'] = type(cfg)({'FLIP_TEST': flip, 'SHIFT_HEATMAP': flip, 'POST_PROCESS': flip})
    cfg['PRINT_FREQ'] = 10 ** 9
    batches, boxes = [], []
    for it in range(2):
        v, t, w = synth_batch('%s.val%d' % (tag, it), B, J, H, W)
        c, s, score = synth_boxes('%s.valbox%d' % (tag, it), B)
        names = ['img/%012d.jpg' % (100 + (it * B + k) // 2) for k in range(B)]
        batches.append((v[0], [t, t], w, {'center': torch.from_numpy(c), 'scale': torch.from_numpy(s),
                                          'score': torch.from_numpy(score), 'image': names}))
        boxes.append((c, s))
    seen, outs, losses = {}, [], []

    class DS:
        flip_pairs = m['pairs']

        def __len__(self):
            return 2 * B

        def evaluate(self, cfg_, preds, out_dir, all_boxes, img_path, *a, **k):
            seen.update(preds=preds.copy(), boxes=all_boxes.copy(), paths=list(img_path))
            return {'AP': 0.0}, 0.0

    crit = JointsMSELoss(True).cuda()

    def rec(o, t, w):
        outs.append(o.detach().clone())
        v = crit(o, t, w)
        losses.append(float(v))
        return v
    validate(cfg, None, batches, DS(), D, rec, '/tmp', '/tmp', None)
    assert seen['paths'] == m['paths']
    assert np.array_equal(seen['boxes'], g[key + '.all_boxes'])
    for it in range(2):
        assert_close('%s merged heat-map %d' % (key, it), strided(outs[it].contiguous(), 2048),
                     g['%s.out%d' % (key, it)])
    assert_close(key + ' losses', np.array(losses), g[key + '.losses'])
    assert abs(validate.last['loss'] - m['loss_avg']) <= 1e-3 * max(1.0, abs(m['loss_avg']))
    want = g[key + '.all_preds']
    assert_close(key + ' maxvals', seen['preds'][:, :, 2], want[:, :, 2])
    same = np.abs(seen['preds'][:, :, 0:2] - want[:, :, 0:2]).max(axis=2) <= 1e-3
    assert same.mean() >= 0.9, same.mean()
    # exact post-processing parity on the product's own heat-maps
    for it in range(2):
        c, s = boxes[it]
        p, mv, _ = oval.get_final_preds(outs[it].contiguous().cpu().numpy(), c, s, flip)
        got = seen['preds'][it * B:(it + 1) * B]
        assert np.array_equal(got[:, :, 2:3], mv)
        ulp = np.spacing(np.abs(p).astype(np.float32))
        assert (np.abs(got[:, :, 0:2].astype(np.float64) - p) <= ulp).all()


def test_coco_rescoring_and_oks_nms_match_reference_fixture():
    """advmix_amd.dataset.coco.rescore_and_nms (device OKS matrix) against COCODataset.evaluate's kept
    persons and scores recorded from the real reference."""
    from oracle import detinit
    from helpers import gold_npz, gold_json
    from advmix_amd.dataset.coco import rescore_and_nms
    g, meta = gold_npz('validate.npz'), gold_json('validate.json')
    for i in range(3):
        m = meta['oks%d' % i]
        N, per_img, J = m['N'], m['per_img'], 17
        base = detinit.uniform('val.oks%d.base' % i, (N // per_img, J, 2)).numpy() * 200 + 50
        jit = detinit.normal('val.oks%d.jit' % i, (N, J, 2), 3.0).numpy()
        kp = np.zeros((N, J, 3), dtype=np.float32)
        kp[:, :, 0:2] = base[np.arange(N) // per_img] + jit
        kp[:, :, 2] = detinit.uniform('val.oks%d.conf' % i, (N, J)).numpy()
        boxes = np.zeros((N, 6))
        boxes[:, 4] = detinit.uniform('val.oks%d.area' % i, (N,)).numpy().astype(np.float64) * 20000 + 5000
        boxes[:, 5] = detinit.uniform('val.oks%d.score' % i, (N,)).numpy().astype(np.float64)
        paths = ['img/%012d.jpg' % (7 + n // per_img) for n in range(N)]
        kept = rescore_and_nms(kp, boxes, paths, J, m['in_vis'], m['oks_thre'], m['soft'])
        flat = []
        for persons in kept:
            for person in persons:
                row = int(np.where((kp == person['keypoints']).all(axis=(1, 2)))[0][0])
                flat.append((int(person['image']), row, float(person['score'])))
        want = g['oks%d.kept' % i]
        assert [(a, b) for a, b, _ in flat] == [(int(a), int(b)) for a, b, _ in want], i
        assert np.allclose([f[2] for f in flat], want[:, 2], rtol=1e-12, atol=0)


def test_launch_chains_equal_the_level_schedule(monkeypatch):
    """Scheduling only: the chain schedule (ops.Chain members, 12 levels for an HRNet) and the one-step-per-member
    schedule launch the same kernels on the same data.  Forward results agree to rounding (not bit for bit: K-split
    convolutions and the BN column sums accumulate with atomics, whose order varies run to run in either schedule).
    Gradients: two separate forwards of a B = 2 net whose deepest maps are 2 x 2 now and then put one pre-activation on
    different sides of zero (profiles/r04_diag_chains_vs_levels.log: either schedule against ITSELF shows the same), so the
    two schedules are not compared with each other under a retry (round 5) but EACH against the fp64 evaluation of the plan
    with that run's own masks (ops.SLOT_TAP, tests/plan_functional.py): every parameter gradient and the input gradient
    element-wise, one attempt."""
    from oracle import configs
    from oracle.synth import synth_batch
    from advmix_amd import plan as plan_mod
    from advmix_amd.core.loss import JointsMSELoss
    from plan_functional import SlotRecorder, interpret, pins_of
    net, extra, J, B, H, W = 'pose_hrnet', configs.HRNET_TINY, 5, 2, 64, 64
    D_sd, T_sd, G_sd = build_states(net, extra, J, salt=40)
    v, t, w = synth_batch('chains.check', B, J, H, W)
    res = {}
    for chains in (True, False):
        monkeypatch.setattr(plan_mod, 'CHAINS', chains)
        cfg, D, G, _ = product_models(net, extra, J, D_sd, T_sd, G_sd)
        kinds = {st[0] for lv in D._levels for st in lv}
        assert ('chain' in kinds) == chains
        D.train()
        x = v[0].cuda().requires_grad_(True)
        with SlotRecorder() as rec:
            out = D(x)
        loss = JointsMSELoss(True).cuda()(out, t.cuda(), w.cuda())
        loss.backward()
        g_out = G(torch.cat(v, 1).cuda())
        got = {k: p.grad.detach().cpu() for k, p in D.named_parameters()}
        got['x'] = x.grad.detach().cpu()
        # fp64 with THIS run's masks
        (slots,) = rec.of(D)
        pin, pool = pins_of(D.plan, slots)
        W64 = {k: p.detach().cpu().double().requires_grad_(True) for k, p in D.named_parameters()}
        x64 = v[0].double().requires_grad_(True)
        y64 = interpret(D.plan, W64, x64, pin=pin, pool_src=pool)[D.plan.out]
        names = list(W64)
        g64 = dict(zip(names + ['x'], torch.autograd.grad(_loss_any_dtype(y64, t, w, B, J), [W64[k] for k in names] + [x64])))
        rel = _pinned_rel_errors(got, g64, names + ['x'])
        top = sorted(rel.items(), key=lambda kv: -kv[1])[:3]
        print('chains' if chains else 'levels', len(D._levels), 'levels; gradients vs fp64 with this run\'s masks: worst of %d' % len(rel),
              [(k, '%.2e' % e) for k, e in top])
        assert float((out.detach().cpu().double() - y64.detach()).abs().max()) <= 1e-5 * float(y64.abs().max())
        assert top[0][1] <= PINNED_TOL_STEP, top
        res[chains] = (out.detach().cpu(), g_out.detach().cpu(), {k: b_.detach().cpu().clone() for k, b_ in D.named_buffers()}, len(D._levels))
    a, b = res[True], res[False]
    assert a[3] < b[3] / 3                                  # far fewer joins
    for i in (0, 1):
        assert float((a[i] - b[i]).abs().max()) <= 1e-5 * max(1.0, float(b[i].abs().max())), 'output %d' % i
    for k in a[2]:                                          # BN running statistics (the deepest ones are variances over 8 rows)
        assert float((a[2][k].double() - b[2][k].double()).abs().max()) <= 1e-4 * max(1.0, float(b[2][k].double().abs().max())), k


def _tiny_setup(salt=10, lr=1e-3):
    from oracle import configs
    from oracle.posenet import calibrate
    from oracle.synth import synth_batch
    from advmix_amd.core.loss import JointsMSELoss
    from advmix_amd.utils.utils import get_optimizer
    net, extra, J, B, H, W = 'pose_hrnet', configs.HRNET_TINY, 5, 2, 64, 64
    D, T, G = build_states(net, extra, J, salt=salt)
    calib = synth_batch('hrnet_tiny.calib', B, J, H, W)[0][0]
    calibrate(net, T, calib, extra)
    calibrate(net, D, calib, extra)
    cfg, mD, mG, mT = product_models(net, extra, J, D, T, G, lr=lr)
    optD, optG = get_optimizer(cfg, mD), get_optimizer(cfg, mG)
    mD.train(); mG.train(); mT.eval()
    return cfg, mD, mG, mT, JointsMSELoss(True), optD, optG, (B, J, H, W)


def test_teacher_riding_in_the_students_launch_groups_changes_no_result(monkeypatch):
    """advmix_phase_a runs the frozen teacher's levels as members of the student's launch groups (default); the two
    separate forwards of function.py:146-149 (ADVMIX_PAIR_TEACHER=0) must give the same loss and the same D gradients
    from the same state - the pairing only changes WHICH launches share the chip."""
    from oracle.synth import synth_batch
    from advmix_amd import ops
    from advmix_amd.core import function as F_
    res = {}
    ops.set_deterministic(True)                                         # so that G's forward is the same bits in both runs
    try:
        for pair in (True, False):
            monkeypatch.setattr(F_, '_PAIR_TEACHER', pair)
            cfg, mD, mG, mT, crit, optD, optG, (B, J, H, W) = _tiny_setup()
            v, t, w = synth_batch('hrnet_tiny.it0', B, J, H, W)
            args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
            loss, tmp = F_.advmix_phase_a(args, mD, mG, mT, crit, optD, [x.cuda().contiguous() for x in v], t.cuda(), w.cuda())
            torch.cuda.synchronize()
            res[pair] = (float(loss), optD.flat_grads.detach().clone(), tmp.detach().clone(),
                         {k: b.detach().clone() for k, b in mT.named_buffers()})
            assert all(p.grad is None for p in mT.parameters())
    finally:
        ops.set_deterministic(False)
    (la, ga, ta, ba), (lb, gb, tb, bb) = res[True], res[False]
    assert torch.equal(ta, tb)                                          # G's forward is not touched
    assert abs(la - lb) <= 1e-6 * max(1.0, abs(lb)), (la, lb)
    scale = float(gb.abs().max())
    d = (ga - gb).abs()
    # same arithmetic, possibly another tile configuration where two members were fused into one grouped launch (another
    # summation order; a ReLU mask can flip at a value that is zero to rounding)
    assert float((d > 1e-4 * scale).float().mean()) <= 1e-3 and float(d.max()) <= 0.05 * scale, (float(d.max()), scale)
    for k in ba:                                                        # an eval-mode teacher: no buffer moves
        assert torch.equal(ba[k], bb[k]), k


def test_teacher_as_a_graph_of_its_own_beside_the_step():
    """ADVMIX_PAIR_TEACHER=2 (round 6, EXPERIMENTS M6: measured 0.8 % slower than the teacher riding in the student's launch
    groups, kept as the A/B switch): the frozen teacher's forward is a HIP graph of its own - own memory pool, own lane set -
    replayed on a side stream at the start of the step beside the generator's and the student's forward; phase a is captured in
    two halves and the loss half waits for it (eagerly: a stream and lane set of its own inside advmix_phase_a1).  The switch is
    read at import, so the tests that hold the graph runner, the loops and the smoke step to the reference run again in a
    child process with it set."""
    import os, subprocess, sys
    here = os.path.abspath(__file__)
    out = subprocess.run([sys.executable, '-m', 'pytest', here, '-q', '-x', '-k',
                          'graph_runner_matches or loops_graph or train_advmix_loop_first or smoke_entry or advmix_and_plain_steps_vs_oracle_and_golden and tiny'],
                         capture_output=True, text=True, timeout=1200, env=dict(os.environ, ADVMIX_PAIR_TEACHER='2'))
    assert out.returncode == 0 and ' passed' in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


def test_train_advmix_loop_first_iteration_matches_reference():
    """The loop mirror itself (function.py:107-197): batches in the reference loader's format, meters, the
    tensorboard counter; the first iteration's loss_D equals the number the REAL train_advmix recorded."""
    from oracle.synth import synth_batch
    from advmix_amd.core.function import train_advmix
    cfg, mD, mG, mT, crit, optD, optG, (B, J, H, W) = _tiny_setup()
    cfg['PRINT_FREQ'] = 2
    g = gold_npz('advmix_steps.npz')
    batches = []
    for it in range(3):
        v, t, w = synth_batch('hrnet_tiny.it%d' % it, B, J, H, W)
        batches.append((v, [t, t, t], [w, w, w], [{}, {}, {}]))
    seen = []
    writer = types.SimpleNamespace(add_scalar=lambda k, v, s: seen.append((k, float(v), s)))
    wd = {'writer': writer, 'train_global_steps': 0}
    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    train_advmix(cfg, args, batches, [mD, mG, mT], crit, [optD, optG], 0, '/tmp', '/tmp', wd)
    assert wd['train_global_steps'] == 2                     # i = 0 and i = 2
    want = g['hrnet_tiny.losses'][0]
    first = [v for k, v, s in seen if k == 'train_loss'][0]
    assert abs(first - (0.9 * want[0] + 0.1 * want[1])) <= 1e-3 * max(1.0, abs(want[0]))
    assert all(not p.requires_grad for p in mD.parameters())          # left frozen, like the reference
    assert int(mD.state_dict()['bn1.num_batches_tracked']) == 1 + 2 * 3   # calibration + two forwards per batch


@pytest.mark.parametrize('graph', [True, False])
def test_loops_graph_path_and_eager_path_match_reference(graph):
    """train_advmix / train capture the step on the first batch and replay it (core.function.GRAPH_EXEC); a batch of
    another shape (the ragged last batch) runs eagerly.  Both modes: first-iteration losses equal the numbers the REAL
    loops recorded, BatchNorm counters advance exactly twice (once) per batch - no trace of the capture warm-up."""
    from oracle.synth import synth_batch
    from advmix_amd.core import function as F_
    g = gold_npz('advmix_steps.npz')
    old = F_.GRAPH_EXEC
    F_.GRAPH_EXEC = graph
    F_.release_graphs()
    try:
        cfg, mD, mG, mT, crit, optD, optG, (B, J, H, W) = _tiny_setup()
        cfg['PRINT_FREQ'] = 1
        batches = []
        for it in range(3):
            v, t, w = synth_batch('hrnet_tiny.it%d' % it, B, J, H, W)
            batches.append((v, [t, t, t], [w, w, w], [{}, {}, {}]))
        v, t, w = synth_batch('hrnet_tiny.it0', B, J, H, W)                 # ragged last batch: B - 1 samples
        batches.append(([x[:1].clone() for x in v], [t[:1]] * 3, [w[:1]] * 3, [{}, {}, {}]))
        seen = []
        wd = {'writer': types.SimpleNamespace(add_scalar=lambda k, v_, s: seen.append((k, float(v_)))),
              'train_global_steps': 0}
        args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
        F_.train_advmix(cfg, args, batches, [mD, mG, mT], crit, [optD, optG], 0, '/tmp', '/tmp', wd)
        assert len(F_._RUNNERS) == (1 if graph else 0)
        want = g['hrnet_tiny.losses'][0]
        losses = [v_ for k, v_ in seen if k == 'train_loss']
        assert len(losses) == 4 and all(np.isfinite(losses))
        assert_close('first loss_D', [losses[0]], [0.9 * want[0] + 0.1 * want[1]])
        assert int(mD.state_dict()['bn1.num_batches_tracked']) == 1 + 2 * 4
        assert all(not p.requires_grad for p in mD.parameters())

        # the plain loop (function.py:30-95)
        from helpers import build_states
        from oracle.posenet import calibrate
        from oracle import configs
        D, T, G = build_states('pose_hrnet', configs.HRNET_TINY, J, salt=20)
        calibrate('pose_hrnet', D, synth_batch('hrnet_tiny.calib', B, J, H, W)[0][0], configs.HRNET_TINY)
        cfg2, pD, _, _ = product_models('pose_hrnet', configs.HRNET_TINY, J, D, T, G)
        cfg2['PRINT_FREQ'] = 1
        from advmix_amd.utils.utils import get_optimizer
        opt = get_optimizer(cfg2, pD)
        pb = []
        for it in range(2):
            v, t, w = synth_batch('hrnet_tiny.plain%d' % it, B, J, H, W)
            pb.append((v[0], [t, t], w, {}))
        seen.clear()
        F_.train(cfg2, None, pb, pD, crit, opt, 0, '/tmp', '/tmp', wd)
        pl = [v_ for k, v_ in seen if k == 'train_loss']
        assert_close('plain first loss', [pl[0]], [g['hrnet_tiny.plain_losses'][0]])
        assert int(pD.state_dict()['bn1.num_batches_tracked']) == 1 + 2
    finally:
        F_.GRAPH_EXEC = old
        F_.release_graphs()


def _dp_one_rank_worker():
    """The data-parallel step with ONE rank and real RCCL (GradSync(force=True), as ADVMIX_FORCE_SYNC=1 does in
    bench.py): the backward pass runs in pieces, every finished range of the flat gradient buffers is all-reduced on
    the side stream beside the next piece, the optimizers wait for them.  With one rank an all-reduce(mean) is the
    identity, so the result must equal the plain single-GPU step: gradients after phase a to rounding (fp32 atomics
    order), the whole step as closely as two eager runs agree.  Eager and HIP-graph execution (seven graphs)."""
    import copy
    import torch.distributed as dist
    from oracle.synth import synth_batch
    from advmix_amd.core.function import advmix_step, advmix_phase_a
    from advmix_amd.dp import GradSync
    from advmix_amd.graph import AdvMixGraphRunner
    if not dist.is_initialized():
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29631', rank=0, world_size=1,
                                device_id=torch.device('cuda:0'))
    try:
        args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
        B, J, H, W = 2, 5, 64, 64
        v, t, w = synth_batch('hrnet_tiny.it0', B, J, H, W)
        data = ([x.cuda().contiguous() for x in v], t.cuda(), w.cuda())
        sync = GradSync(force=True, bucket_mb=0.25)
        assert sync.active
        # (1) gradients of phase a: plain backward vs pieces + all-reduces.  In the deterministic mode (ordered partial
        # sums instead of fp32 atomics), so the comparison sees the piece-wise path and not the run-to-run noise of the
        # atomics (with them, 1e-4 of the largest gradient is exceeded once in a few dozen runs).
        from advmix_amd import ops as _o
        _o.set_deterministic(True)
        try:
            cfg, D1, G1, T1, crit, oD1, oG1, _ = _tiny_setup(lr=1e-4)
            cfg, D2, G2, T2, crit2, oD2, oG2, _ = _tiny_setup(lr=1e-4)
            sync.broadcast_state([D2, G2, T2], [oD2, oG2])
            l1, _ = advmix_phase_a(args, D1, G1, T1, crit, oD1, *data)
            cuts = sync.cuts_for(D2)
            assert len(cuts) == 2 and 0 < cuts[0][1] < cuts[1][1] < oD2.flat_grads.numel()
            l2, _, pieces, _pg = advmix_phase_a(args, D2, G2, T2, crit2, oD2, *data, (cuts, sync.cuts_for(G2)))
            done = []
            for piece in pieces:
                lo, hi = piece()
                sync.reduce_async(oD2.flat_grads, lo, hi)
                done.append((lo, hi))
            sync.finish()
            torch.cuda.synchronize()
        finally:
            _o.set_deterministic(False)
        assert [d[1] for d in done] == [oD2.flat_grads.numel(), cuts[1][1], cuts[0][1]] and done[-1][0] == 0
        assert abs(float(l1) - float(l2)) <= 1e-6 * max(1.0, abs(float(l1)))
        g1, g2 = oD1.flat_grads, oD2.flat_grads
        err, ref = float((g1 - g2).abs().max()), float(g1.abs().max())
        assert err <= 1e-5 * ref, (err, ref)
        assert float(g2.abs().max()) > 0
        # (2) the whole step, eager: synced pieces vs plain
        cfg, D1, G1, T1, crit, oD1, oG1, _ = _tiny_setup(lr=1e-4)
        cfg, D2, G2, T2, crit2, oD2, oG2, _ = _tiny_setup(lr=1e-4)
        la, oa = advmix_step(args, D1, G1, T1, crit, oD1, oG1, *data)
        lb, ob = advmix_step(args, D2, G2, T2, crit2, oD2, oG2, *data, sync)
        assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(la)))
        assert float((oa - ob).abs().max()) <= 0.05 * float(oa.abs().max())
        fa, fb = oG1.flat_params, oG2.flat_params
        assert float(((fa - fb).abs() > 2e-5).float().mean()) <= 0.1           # G really was updated identically
        assert float((oG2.flat_grads).abs().max()) > 0
        # (3) HIP graphs: seven segments, all-reduces between the replays
        cfg, D3, G3, T3, crit3, oD3, oG3, _ = _tiny_setup(lr=1e-4)
        runner = AdvMixGraphRunner(args, D3, G3, T3, crit3, oD3, oG3, *data, grad_sync=sync, warmup=1)
        assert len(runner.segments) == 7
        lg, og = runner.step()
        assert abs(float(lg) - float(la)) <= 1e-6 * max(1.0, abs(float(la)))
        assert float((og - oa).abs().max()) <= 0.05 * float(oa.abs().max())
        assert int(D3.state_dict()['bn1.num_batches_tracked']) == int(D1.state_dict()['bn1.num_batches_tracked'])
    finally:
        dist.destroy_process_group()


def _dp_two_rank_worker():
    """WORLD SIZE 2 on ONE GPU: both ranks use cuda:0 and exchange over gloo (which all-reduces device tensors through
    pinned host memory) - RCCL refuses two ranks on one device, and the build pool has no multi-GPU box.  Everything
    else is the real data-parallel path: ranks start from DIFFERENT weights, ``broadcast_state`` makes them equal, each
    rank steps on its own batch (eagerly, then through the seven-graph runner), the backward pass runs in pieces with
    every finished range of the flat gradient buffers averaged on the side stream.  Asserted: the averaged gradient is
    exactly the mean of the two ranks' own gradients (deterministic mode); parameters and Adam moments of D and G stay
    BIT-identical across the ranks after every step; BatchNorm statistics and losses stay per replica."""
    import os
    import torch.distributed as dist
    from oracle.synth import synth_batch
    from advmix_amd import ops as _o
    from advmix_amd.core.function import advmix_step, advmix_phase_a
    from advmix_amd.dp import GradSync
    from advmix_amd.graph import AdvMixGraphRunner
    rank = int(os.environ['RANK'])
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % os.environ['MASTER_PORT'], rank=rank, world_size=2)
    torch.cuda.set_device(0)
    # Everything below runs as the training loops run it: on a created stream (core.function._loop_stream), never the NULL
    # stream.  Round 4 (tools/dp_graph_repro.py): on a GPU shared by two ranks, NULL-stream traffic
    # between two graph replays - this test's own all_gathers and loss.item() reads in round 3 - makes the next replays
    # compute garbage (the runtime's captured-packet launches; 0 failures with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 or with
    # the NULL stream left alone).
    from advmix_amd.core.function import _loop_stream
    with _loop_stream():
        assert torch.cuda.current_stream().cuda_stream != 0
        _dp_two_rank_body(rank)
    dist.barrier()
    dist.destroy_process_group()


def _dp_two_rank_body(rank):
    import os
    import torch.distributed as dist
    from oracle.synth import synth_batch
    from advmix_amd import ops as _o
    from advmix_amd.core.function import advmix_step, advmix_phase_a
    from advmix_amd.dp import GradSync
    from advmix_amd.graph import AdvMixGraphRunner
    sync = GradSync(bucket_mb=0.25)
    assert sync.active and sync.world == 2

    def gathered(t):
        # NOT with the NULL stream current: round 3's "open bug" was exactly this - the test's own all_gathers between the
        # replays, issued on the NULL stream, made the NEXT graph replays compute garbage (DESIGN.md section 4, 28 of 29 runs;
        # 0 of 11 with the NULL stream left alone).  The product's collectives all go through off_null.
        got = [torch.zeros_like(t), torch.zeros_like(t)]
        with sync.off_null(t):
            dist.all_gather(got, t.contiguous())
        return got

    def same(t):
        a, b = gathered(t)
        return bool(torch.equal(a, b))
    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    B, J, H, W = 2, 5, 64, 64
    v, t, w = synth_batch('hrnet_tiny.it%d' % rank, B, J, H, W)             # a different shard per rank
    data = ([x.cuda().contiguous() for x in v], t.cuda(), w.cuda())
    cfg, D, G, T, crit, oD, oG, _ = _tiny_setup(salt=10 + 7 * rank, lr=1e-3)  # and different initial weights
    assert not same(oD.flat_params) and not same(oG.flat_params)
    sync.broadcast_state([D, G, T], [oD, oG])
    assert same(oD.flat_params) and same(oG.flat_params) and all(same(b.float()) for b in D.buffers())
    # (1) the exchanged gradient of the D step = the mean of the ranks' own gradients, bit for bit (ordered sums)
    _o.set_deterministic(True)
    try:
        snap = [x.clone() for x in oD.flat_state()] + [b.clone() for b in D.buffers()]
        l0, _ = advmix_phase_a(args, D, G, T, crit, oD, *data)             # local gradient, no exchange
        torch.cuda.synchronize()
        g0, g1 = gathered(oD.flat_grads.clone())
        want = (g0 + g1) * 0.5
        with torch.no_grad():
            for dst, src in zip(oD.flat_state() + list(D.buffers()), snap):
                dst.copy_(src)
        cuts = (sync.cuts_for(D), sync.cuts_for(G))
        l1, _, pieces, _pg = advmix_phase_a(args, D, G, T, crit, oD, *data, cuts)
        for piece in pieces:
            lo, hi = piece()
            sync.reduce_async(oD.flat_grads, lo, hi)
        sync.finish()
        torch.cuda.synchronize()
        assert float(l0) == float(l1)
        assert torch.equal(oD.flat_grads, want), float((oD.flat_grads - want).abs().max())
        assert not torch.equal(g0, g1)
        with torch.no_grad():
            for dst, src in zip(oD.flat_state() + list(D.buffers()), snap):
                dst.copy_(src)
    finally:
        _o.set_deterministic(False)
    # (2) whole steps: eager, then the seven-graph runner
    losses = []
    l, o = advmix_step(args, D, G, T, crit, oD, oG, *data, sync)
    torch.cuda.synchronize()
    losses.append(float(l))
    assert all(same(x) for x in oD.flat_state()) and all(same(x) for x in oG.flat_state())
    for _ in range(2):                                                      # two more steps from the moved state
        l, o = advmix_step(args, D, G, T, crit, oD, oG, *data, sync)
        losses.append(float(l))
    torch.cuda.synchronize()
    assert all(same(x) for x in oD.flat_state()) and all(same(x) for x in oG.flat_state())
    # the seven-graph runner (what more than one rank runs by default): 20 replays, replicas identical and finite after each.
    # Round 3's open bug (replays on the NULL stream with a second process on the GPU: garbage from the second replay on,
    # 21 of 21 runs on the round-4 box with this test's old NULL-stream checks) lived exactly here; the worker now runs as the loops do.
    runner = AdvMixGraphRunner(args, D, G, T, crit, oD, oG, *data, sync)
    assert runner.seq.n_graphs == 7
    for k in range(20):
        l, o = runner.step()
        losses.append(float(l))
        st = sync.replicas_state([oD, oG])
        assert st == {'identical': True, 'finite': True}, (k, st)
    torch.cuda.synchronize()
    assert all(same(x) for x in oD.flat_state()) and all(same(x) for x in oG.flat_state())
    sync.trace = []                                                         # and one traced step: every exchange == the mean
    runner.step()
    torch.cuda.synchronize()
    v_ok, v_worst = sync.verify_trace()
    assert v_ok and v_worst <= 2e-7 and len(sync.trace) == 6, (v_ok, v_worst, len(sync.trace))   # (fp32 sum vs the fp64 mean)
    sync.trace = None
    rm = D.state_dict()['bn1.running_mean']
    assert not same(rm)                                                     # statistics stay per replica
    la, lb = gathered(torch.tensor(losses, device='cuda'))
    assert not torch.equal(la, lb) and all(x == x for x in losses)          # each rank's own shard, finite
    assert sync.checkpoint_rank() == (rank == 0)
    # (3) the reference's own call site: train_advmix(...) WITHOUT a grad_sync argument, models wrapped like
    # tools/train.py:69,106,109 wraps them (dp.Replica bound as DataParallel): the loop finds the process group, creates
    # its GradSync, broadcasts rank 0's state and keeps the replicas identical (core.function._auto_sync)
    from advmix_amd.core import function as F_
    from advmix_amd.dp import Replica
    import logging
    logging.getLogger(F_.__name__).setLevel(logging.WARNING)
    cfg, D, G, T, crit, oD, oG, _ = _tiny_setup(salt=30 + 5 * rank, lr=1e-3)
    cfg['PRINT_FREQ'] = 10 ** 9
    assert not same(oD.flat_params)
    batches = []
    for it in range(3):
        vv, tt, ww = synth_batch('hrnet_tiny.loop%d.%d' % (rank, it), B, J, H, W)
        batches.append((vv, [tt, tt, tt], [ww, ww, ww], [{}, {}, {}]))
    wd = {'writer': types.SimpleNamespace(add_scalar=lambda *a, **k: None), 'train_global_steps': 0}
    F_.train_advmix(cfg, args, batches, [Replica(D), Replica(G), Replica(T)], crit, [oD, oG], 0, '/tmp', '/tmp', wd)
    torch.cuda.synchronize()
    assert all(same(x) for x in oD.flat_state()) and all(same(x) for x in oG.flat_state())
    assert not same(D.state_dict()['bn1.running_mean'])
    F_.release_graphs()


@pytest.mark.parametrize('lanes', ['4', '1'])
def test_data_parallel_path_two_ranks_on_one_gpu_over_gloo(lanes):
    """SURVEY 8 e / a12 with a world size of TWO ("RCCL has never seen N > 1 ranks" - it still has not: no
    multi-GPU box; this runs the same path with gloo as the transport, both ranks on cuda:0).  One lane - every captured
    segment a single chain - was round 3's 10-of-10 failure."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ('import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_models_gpu as t; '
            't._dp_two_rank_worker(); print("DP_TWO_RANKS_OK")' % (os.path.dirname(here), here))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='2964%s' % lanes,
                   ADVMIX_LANES=lanes)
        procs.append(subprocess.Popen([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and 'DP_TWO_RANKS_OK' in so, (so[-2000:], se[-4000:])


def test_data_parallel_path_on_one_rank_real_rccl():
    """Runs ``_dp_one_rank_worker`` in a CHILD process: RCCL's communicator setup / teardown and the seven-graph runner
    stay out of the pytest process (a HIP-graph replay after destroy_process_group() in the same process crashed the
    HIP runtime)."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ('import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_models_gpu as t; '
            't._dp_one_rank_worker(); print("DP_ONE_RANK_OK")' % (os.path.dirname(here), here))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and 'DP_ONE_RANK_OK' in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])


def test_frozen_teachers_filter_images_are_remade_only_when_its_filters_change(monkeypatch):
    """plan.PlanNet._wino_refresh: the Winograd / small-map filter images of a network are re-made at the start of EVERY forward
    pass - except for a network core.function has marked frozen (``wino_static``: the AdvMix teacher) in eval mode, whose
    filters only change through torch operations (version counters): once, then again only after such a change; any
    train-mode forward re-makes them regardless."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    from bench_common import HRNET_STAGES, hrnet_extra
    from advmix_amd import models, ops
    from advmix_amd.config import CfgNode
    cfg = CfgNode({'MODEL': {'NAME': 'pose_hrnet', 'NUM_JOINTS': 17, 'INIT_WEIGHTS': True, 'PRETRAINED': '',
                             'EXTRA': hrnet_extra(HRNET_STAGES['hrnet_w32'])}})
    net = models.pose_hrnet.get_pose_net(cfg, is_train=True).cuda()
    with torch.no_grad():
        for p_ in net.parameters():                         # (init_weights' N(0, 1e-3) filters: outputs of 1e-10, residual branches invisible)
            if p_.dim() == 4:
                torch.nn.init.kaiming_normal_(p_, mode='fan_in')
    calls = []
    real = ops.WinoBank.refresh
    monkeypatch.setattr(ops.WinoBank, 'refresh', lambda self, st=None: (calls.append(1), real(self, st))[1])
    x = torch.randn(16, 3, 128, 96, device='cuda')          # (16 images: the 32- and 64-channel branches reach the Winograd kernel's 96 workgroups)
    net.eval()
    with torch.no_grad():
        y0 = net(x)
        net(x)
        assert len(calls) == 2                              # not marked: every forward
        net.wino_static = True
        net(x)
        y1 = net(x)
        assert len(calls) == 3, len(calls)                  # marked: once
        assert float((y0 - y1).abs().max()) <= 1e-5 * float(y0.abs().max())     # (small maps split K over the grid: fp32 atomics, not bit-equal)
        assert ops.COUNTERS.get('wino', 0) > 0
        w = next(p for n, p in net.named_parameters() if n.endswith('conv2.weight') and p.shape[1] == 32 and p.dim() == 4)
        w.mul_(1.5)                                         # a torch operation on one filter bank
        y2 = net(x)
        assert len(calls) == 4
        net(x)
        assert len(calls) == 4
    monkeypatch.setattr(ops, 'WINO', False)                 # the same forward on the direct kernels: the images were current
    with torch.no_grad():
        y3 = net(x)
    assert float((y2 - y3).abs().max()) <= 1e-4 * float(y3.abs().max()) and float((y0 - y2).abs().max()) > 1e-3 * float(y0.abs().max())
    monkeypatch.setattr(ops, 'WINO', True)
    net.train()
    n0 = len(calls)
    net(x)
    net(x)
    assert len(calls) == n0 + 2


def test_bench_falls_back_to_the_eager_step_when_the_graphs_do_not_verify():
    """bench.py's N-rank run verifies its own execution before it times anything; when the replayed graphs fail that (forced
    here: ADVMIX_BENCH_FAIL_GRAPH_VERIFY=1, one rank with real RCCL) the SAME step runs without graphs from rank 0's state,
    is verified again, and the line says so - a number from an unverified execution is never printed as valid."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ADVMIX_FORCE_SYNC='1', ADVMIX_BENCH_FAIL_GRAPH_VERIFY='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT='29671', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--batch', '4', '--steps', '2', '--warmup', '1',
                          '--no-roofline', '--no-cpu-baseline', '--no-through-loop'], capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line['config']['exec'] == 'eager' and line['grad_exchange_verified'] is True
    assert line['replicas_identical'] is True and line['all_finite'] is True
    v = line['dp_verification']
    assert v['exec'] == 'eager' and all(v['checks'].values()) and 'graph_attempt' in v and v['graph_attempt']['verified'] is True
    # the line's self-diagnosis: this rank's own step time, what the two exchanges made the compute stream
    # wait (real RCCL all-reduces on the side stream, measured between two events), the bytes of D's + G's flat gradients
    r = line['ranks']
    assert r['transport'] == 'nccl' and len(r['ms_per_step']['per_rank']) == 1 and r['ms_per_step']['max'] > 0
    assert r['exchange_wait_ms']['max'] >= 0.0 and r['exchange_ranges_per_step'] == 6.0      # three pieces per network
    assert 200e6 < r['exchange_bytes_per_step'] < 260e6                                       # 108.9 + 111.6 MiB


@pytest.mark.parametrize('transport', ['device', 'host round trip'])
def test_seven_graph_runner_equals_the_eager_pieces_bit_for_bit_when_the_exchange_changes_the_data(transport):
    """VERDICT r3 item 1's separating experiment, kept as a test: ONE process, the gradient exchange replaced by a transport
    that CHANGES the data on the side stream (x 0.5 on the device; or gloo's path restated - an internal stream waits for an
    event of the side stream, copies to pinned memory, the HOST waits and halves, copies back, the side stream waits for that
    copy's event), seven-graph runner against the eager pieces from the same state in deterministic mode: parameters, Adam
    moments and gradients bit-identical after every one of four steps.  With one real rank an all-reduce is the identity and
    hides every ordering mistake between graphs, pieces and side stream; this does not."""
    from oracle.synth import synth_batch
    from advmix_amd import ops
    from advmix_amd.core.function import advmix_step
    from advmix_amd.dp import GradSync
    from advmix_amd.graph import AdvMixGraphRunner
    pool = [torch.cuda.Stream() for _ in range(3)]
    count = [0]

    class ChangingSync(GradSync):
        def __init__(self):
            super().__init__(bucket_mb=0.25, force=True)

        def _mean_(self, t):
            if transport == 'device':
                t.mul_(0.5)
                return
            cur = torch.cuda.current_stream()
            s = pool[count[0] % len(pool)]
            count[0] += 1
            ev = torch.cuda.Event()
            ev.record(cur)
            s.wait_event(ev)
            with torch.cuda.stream(s):
                tmp = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                tmp.copy_(t, non_blocking=True)
            s.synchronize()
            tmp.mul_(0.5)
            with torch.cuda.stream(s):
                t.copy_(tmp, non_blocking=True)
                ev2 = torch.cuda.Event()
                ev2.record(s)
            cur.wait_event(ev2)

    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    B, J, H, W = 2, 5, 64, 64
    v, t, w = synth_batch('hrnet_tiny.it0', B, J, H, W)
    data = ([x.cuda().contiguous() for x in v], t.cuda(), w.cuda())
    ops.set_deterministic(True)
    try:
        cfg, D, G, T, crit, oD, oG, _ = _tiny_setup(salt=10)
        cfg, D2, G2, T2, crit2, oD2, oG2, _ = _tiny_setup(salt=10)
        runner = AdvMixGraphRunner(args, D, G, T, crit, oD, oG, *data, ChangingSync())
        assert runner.seq.n_graphs == 7
        sync2 = ChangingSync()
        for k in range(4):
            runner.step()
            advmix_step(args, D2, G2, T2, crit2, oD2, oG2, *data, sync2)
            torch.cuda.synchronize()
            for a_, b_ in zip(oD.flat_state() + oG.flat_state() + [oD.flat_grads, oG.flat_grads],
                              oD2.flat_state() + oG2.flat_state() + [oD2.flat_grads, oG2.flat_grads]):
                assert torch.equal(a_, b_), k
        assert float(oD.flat_grads.abs().max()) > 0 and bool(torch.isfinite(oG.flat_params).all())
    finally:
        ops.set_deterministic(False)


def test_deterministic_mode_is_bit_reproducible():
    """ops.set_deterministic(True): ordered partial sums instead of fp32 / fp64 atomics everywhere (weight and bias
    gradients, the loss sum, BatchNorm statistics, no K split across the grid).  Two AdvMix steps run twice from the
    same state: every parameter, Adam moment and BatchNorm buffer BIT-identical; and the deterministic step agrees with
    the default (atomic) one to rounding."""
    from oracle.synth import synth_batch
    from advmix_amd import ops
    from advmix_amd.core.function import advmix_step
    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    B, J, H, W = 2, 5, 64, 64
    batches = []
    for it in range(2):
        v, t, w = synth_batch('hrnet_tiny.it%d' % it, B, J, H, W)
        batches.append(([x.cuda().contiguous() for x in v], t.cuda(), w.cuda()))

    def run():
        cfg, D, G, T, crit, oD, oG, _ = _tiny_setup(lr=1e-3)
        outs = []
        for b in batches:
            l, o = advmix_step(args, D, G, T, crit, oD, oG, *b)
            outs += [l.clone(), o.clone()]
        torch.cuda.synchronize()
        state = [t.clone() for opt in (oD, oG) for t in opt.flat_state()] + [b.clone() for b in D.buffers()]
        return outs + state
    try:
        ops.set_deterministic(True)
        a, b = run(), run()
    finally:
        ops.set_deterministic(False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)                               # bit for bit
    c = run()                                                  # default mode (atomics)
    assert abs(float(a[0]) - float(c[0])) <= 1e-6 * max(1.0, abs(float(c[0])))
    assert float((a[1] - c[1]).abs().max()) <= 0.05 * float(c[1].abs().max())


def test_graph_runner_matches_eager_step_from_the_same_state():
    """AdvMixGraphRunner (three HIP graphs, static batch buffers, load_batch) against the eager advmix_step.
    Two runs cannot be compared across several Adam updates - eager itself is not reproducible there: Adam's
    first steps move every element by ~lr * sign(g), and the sign of a rounding-noise gradient follows the order
    of the fp32 atomics (measured: two eager runs from one state differ by 3.6 % in loss after ONE update).  So
    the graph is replayed and the eager step run from the SAME parameter / optimizer / BN state: loss_D (computed
    before any update) must agree tightly, everything after the D update loosely."""
    import copy
    from oracle.synth import synth_batch
    from advmix_amd.core.function import advmix_step
    from advmix_amd.graph import AdvMixGraphRunner
    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    B, J, H, W = 2, 5, 64, 64
    data = []
    for it in range(2):
        v, t, w = synth_batch('hrnet_tiny.it%d' % it, B, J, H, W)
        data.append(([x.cuda().contiguous() for x in v], t.cuda(), w.cuda()))
    cfg, gD, gG, gT, crit, goD, goG, _ = _tiny_setup(lr=1e-4)
    runner = AdvMixGraphRunner(args, gD, gG, gT, crit, goD, goG, *data[0], warmup=2)
    # clone the post-warm-up state into eager models
    cfg, mD, mG, mT, crit2, optD, optG, _ = _tiny_setup(lr=1e-4)
    mD.load_state_dict(copy.deepcopy(gD.state_dict()))
    mG.load_state_dict(copy.deepcopy(gG.state_dict()))
    optD.load_state_dict(copy.deepcopy(goD.state_dict()))
    optG.load_state_dict(copy.deepcopy(goG.state_dict()))
    runner.load_batch(*data[1])
    loss_g, out_g = runner.step()
    loss_g, out_g = float(loss_g), out_g.detach().float().cpu().clone()
    loss_e, out_e = advmix_step(args, mD, mG, mT, crit2, optD, optG, *data[1])
    assert abs(float(loss_e) - loss_g) <= 1e-4 * max(1.0, abs(loss_g)), (float(loss_e), loss_g)
    scale = float(out_e.abs().max())
    assert float((out_e.detach().float().cpu() - out_g).abs().max()) <= 0.05 * scale      # after one Adam update
    tot = bad = 0
    for (k, a_), (_, b_) in zip(list(mD.named_parameters()) + list(mG.named_parameters()),
                                list(gD.named_parameters()) + list(gG.named_parameters())):
        d = (a_.detach() - b_.detach()).abs()
        tot += d.numel()
        bad += int((d > 2e-5).sum())
    assert 1.0 - bad / tot >= 0.9, 1.0 - bad / tot
    assert int(gD.state_dict()['bn1.num_batches_tracked']) == int(mD.state_dict()['bn1.num_batches_tracked'])
