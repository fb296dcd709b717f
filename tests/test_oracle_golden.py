"""Pin the CPU oracle against vectors produced by the REAL reference
(oracle/gen_golden.py, build container).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import detinit
from oracle import nms as onms
from oracle.loss import joints_loss
from oracle.posenet import posenet_spec, posenet_forward, calibrate, trainable
from oracle.unet import unet_spec, unet_forward
from oracle.step import Adam, advmix_step, plain_step
from oracle.synth import synth_batch, strided, checksum
from helpers import CASES, ALL_FORWARD, DOWNS, gold_files, gold_json, gold_npz, build_states, close, checksum_close, GOLD
from oracle import configs

torch.set_num_threads(8)


def test_state_dict_keys_match_reference():
    ref = gold_json('state_dict_keys.json')
    for tag, net, extra, J in (('hrnet_w32', 'pose_hrnet', configs.HRNET_W32, 17),
                               ('hrnet_w48', 'pose_hrnet', configs.HRNET_W48, 17),
                               ('resnet50', 'pose_resnet', configs.RES50, 17),
                               ('hrnet_tiny', 'pose_hrnet', configs.HRNET_TINY, 5),
                               ('resnet18_tiny', 'pose_resnet', configs.RES18_TINY, 5)):
        mine = {k: list(s) for k, s in posenet_spec(net, extra, J)}
        want = {k: s for k, s in ref[tag]}
        assert mine == want, tag
    for downs in (5, 6):
        mine = {k: list(s) for k, s in unet_spec(9, 3, downs)}
        assert mine == {k: s for k, s in ref['unet%d' % downs]}


def test_loss_known_answers():
    g = gold_json('loss_kat.json')
    o = (torch.arange(48, dtype=torch.float32) / 8 - 1.5).reshape(2, 3, 4, 2)
    t = torch.zeros(2, 3, 4, 2)
    t[:, :, 1, 1] = 1
    w = torch.tensor([[1, 0, 1], [1, 1, .5]]).reshape(2, 3, 1)
    k = g['kat']
    assert abs(k['smoothl1_w'] - 0.5127766728401184) < 1e-7
    assert abs(float(joints_loss(o, t, w, True)) - k['smoothl1_w']) < 1e-6
    assert abs(float(joints_loss(o, t, w, True, smooth_L1=True)) - k['mse_w']) < 1e-6
    assert abs(float(joints_loss(o, t, w, False)) - k['smoothl1_now']) < 1e-6
    assert abs(float(joints_loss(o[:1], t[:1], w[:1], True)) - k['b1']) < 1e-6
    for i, (name, c) in enumerate(sorted(g['random'].items())):
        B, J, H, W = c['shape']
        o = detinit.normal('loss.o%d' % i, (B, J, H, W), c['scale']).requires_grad_(True)
        t = detinit.uniform('loss.t%d' % i, (B, J, H, W))
        w = (detinit.uniform('loss.w%d' % i, (B, J, 1)) < 0.7).float()
        v = joints_loss(o, t, w, True)
        v.backward()
        assert abs(float(v) - c['loss']) < 1e-6 * max(1, abs(c['loss']))
        close(strided(o.grad, 64), c['grad_sample'], 1e-8, 1e-5)


def _dets(name, c):
    if c['dets'] is not None:
        return np.array(c['dets'], np.float32)
    return np.load('%s/nms_dets_%s.npy' % (GOLD, name))


def test_box_nms_matches_reference_numpy():
    g = gold_json('nms.json')['box']
    for name, c in g.items():
        d = _dets(name, c)
        assert onms.py_nms(d, c['thresh']) == c['keep'], name
        # where no IoU sits on the threshold the three reference semantics coincide
        assert onms.gpu_nms(d, c['thresh']) == c['keep'], name
        assert onms.cpu_nms(d, c['thresh']) == c['keep'], name


def test_box_nms_threshold_edge_semantics():
    # two 10x10 boxes (+1 convention: 11x11 px) shifted so IoU == 1/3 exactly:
    # inter = 11*5.5?  use integer geometry: a=[0,0,9,9] (100), b=[5,0,14,9] -> inter 50, union 150
    d = np.array([[0, 0, 9, 9, 0.9], [5, 0, 14, 9, 0.8]], np.float32)
    th = float(np.float32(50.0) / np.float32(150.0))          # the fp32 quotient, as a double
    assert onms.py_nms(d, th) == [0, 1]                        # keeps ovr <= thresh (nms.py:69)
    assert onms.gpu_nms(d, th) == [0, 1]                       # strict > in fp32 (nms_kernel.cu:70)
    assert onms.cpu_nms(d, th) == [0]                          # >= in double (cpu_nms.pyx:68)
    assert onms.py_nms(d[:0], 0.5) == [] and onms.gpu_nms(d[:0], 0.5) == []
    keep, mask = onms.gpu_nms(d, 0.3, return_mask=True)
    assert keep == [0] and mask[0, 0] == 2 and mask[1, 0] == 0


def test_oks_nms_matches_reference():
    g = gold_json('nms.json')['oks']
    for name, c in g.items():
        k = np.array(c['kpts'])
        db = [{'score': s, 'keypoints': kk, 'area': a} for s, kk, a in zip(c['score'], k, c['area'])]
        assert onms.oks_nms(db, c['thresh']) == c['keep'], name
        assert onms.soft_oks_nms(db, c['thresh']) == c['soft_keep'], name


def test_accuracy_matches_reference():
    """evaluate.accuracy (evaluate.py:41-99) on seeded heat-maps incl. ties, all-negative maps, border / missing targets:
    the oracle against what the REAL reference returned."""
    from oracle.gen_golden import accuracy_cases
    from oracle.loss import accuracy as oacc
    g = gold_json('accuracy_kat.json')
    for name, (o, t) in accuracy_cases().items():
        acc, avg, cnt, pred = oacc(o, t)
        assert acc.tolist() == g[name]['acc'] and avg == g[name]['avg'] and cnt == g[name]['cnt'], name
        assert pred.tolist() == g[name]['pred'], name
    assert -1.0 in g['edge']['acc'] and g['edge']['cnt'] < 5


def test_oks_with_in_vis_thre_matches_reference():
    """nms.py:90-92 (the visibility filter of oks_iou, threaded through oks_nms / soft_oks_nms) against what the REAL
    reference returned (oracle/gen_golden.py::gen_nmsvis), OKS values bit for bit."""
    g = gold_json('nms_vis.json')
    assert len(g) == 12
    for name, c in g.items():
        k = np.array(c['kpts'])
        db = [{'score': s, 'keypoints': kk, 'area': a} for s, kk, a in zip(c['score'], k, c['area'])]
        kf, ar = k.reshape(len(db), -1), np.array(c['area'])
        row = onms.oks_iou(kf[0], kf, ar[0], ar, None, c['in_vis_thre'])
        assert row.tolist() == c['iou_row0'], name
        assert onms.oks_nms(db, c['thresh'], None, c['in_vis_thre']) == c['keep'], name
        assert onms.soft_oks_nms(db, c['thresh'], None, c['in_vis_thre']) == c['soft_keep'], name
    assert any(0.0 in c['iou_row0'] for c in g.values())            # the "no joint passes" branch is in the fixture


@pytest.mark.parametrize('tag', list(ALL_FORWARD))
def test_forward_backward_matches_reference(tag):
    net, extra, J, B, H, W, _ = ALL_FORWARD[tag]
    downs = DOWNS.get(tag, 6)
    g = gold_npz(gold_files(tag)[0])
    D, _, G = build_states(net, extra, J, unet_downs=downs)
    views, tgt, tw = synth_batch(tag, B, J, H, W)
    calibrate(net, D, views[2], extra)
    with torch.no_grad():
        ye = posenet_forward(net, D, views[0], extra, False)
    names = trainable(D)
    for k in names:
        D[k].requires_grad_(True)
    x = views[1].clone().requires_grad_(True)
    yt = posenet_forward(net, D, x, extra, True)
    loss = joints_loss(yt, tgt, tw, True)
    grads = dict(zip(names + ['x'], torch.autograd.grad(loss, [D[k] for k in names] + [x])))
    close(strided(ye), g[tag + '.eval_out'])
    close(strided(yt), g[tag + '.train_out'])
    close([float(loss)], g[tag + '.loss'], 1e-5, 1e-4)
    close(strided(grads['x']), g[tag + '.dx'], 1e-6, 2e-3)
    for key in g.files:
        if key.startswith(tag + '.grad.'):
            k = key[len(tag) + 6:]
            s, a = g[key]
            gs, ga = float(grads[k].double().sum()), float(grads[k].double().abs().sum())
            assert abs(ga - a) <= 2e-3 * a and abs(gs - s) <= 2e-3 * a, (k, gs, ga, s, a)
        if key.startswith(tag + '.bn.'):
            close(D[key[len(tag) + 4:]].detach().numpy(), g[key], 1e-4, 1e-3)
    for k in G:
        G[k].requires_grad_(True)
    lg = unet_forward(G, torch.cat(views, 1), num_downs=downs)
    close(strided(lg), g[tag + '.unet_out'])
    gg = dict(zip(G, torch.autograd.grad((lg * detinit.normal(tag + '.gproj', lg.shape)).sum(), list(G.values()))))
    for key in g.files:
        if key.startswith(tag + '.ggrad.'):
            k = key[len(tag) + 7:]
            s, a = g[key]
            assert abs(float(gg[k].double().abs().sum()) - a) <= 2e-3 * a, k
            assert abs(float(gg[k].double().sum()) - s) <= 2e-3 * a, k


@pytest.mark.parametrize('tag', list(CASES))
def test_advmix_and_plain_steps_match_reference(tag):
    net, extra, J, B, H, W, iters = CASES[tag]
    downs = DOWNS.get(tag, 6)
    g = gold_npz(gold_files(tag)[1])
    meta = gold_json(gold_files(tag)[2])[tag]
    D, T, G = build_states(net, extra, J, unet_downs=downs, salt=10)
    calib = synth_batch(tag + '.calib', B, J, H, W)[0][0]
    calibrate(net, T, calib, extra)
    calibrate(net, D, calib, extra)
    optD = Adam(D, trainable(D))
    optG = Adam(G, list(G))
    for it in range(iters):
        v, t, w = synth_batch('%s.it%d' % (tag, it), B, J, H, W)
        r = advmix_step(net, extra, D, G, T, optD, optG, v, t, w, alpha=0.1, unet_kw={'num_downs': downs})
        want = g[tag + '.losses'][it]
        close([float(r['l_hm']), float(r['l_kd']), -float(r['loss_G'])], want, 1e-4, 1e-3)
        close(strided(r['out1'], 2048), g['%s.out1.it%d' % (tag, it)])
        close(strided(r['out2'], 2048), g['%s.out2.it%d' % (tag, it)], 2e-3, 2e-3)
    assert int(D['bn1.num_batches_tracked']) == meta['nbt']       # calib + 2 per iteration
    checksum_close(checksum(D, meta['D'].keys()), meta['D'])
    checksum_close(checksum(G, meta['G'].keys()), meta['G'])
    if 'plain_D' not in meta:                              # (C4 fixture: AdvMix loop only)
        return

    D, _, _ = build_states(net, extra, J, salt=20)
    calibrate(net, D, calib, extra)
    optD = Adam(D, trainable(D))
    for it in range(2):
        v, t, w = synth_batch('%s.plain%d' % (tag, it), B, J, H, W)
        r = plain_step(net, extra, D, optD, v[0], t, w)
        close([float(r['loss'])], [g[tag + '.plain_losses'][it]], 1e-4, 1e-3)
    checksum_close(checksum(D, meta['plain_D'].keys()), meta['plain_D'])


def test_forward_at_the_benchmarked_batch_matches_reference():
    """HRNet-W32 256x192 at B = 32 (the batch bench.py runs): eval forward, train forward and loss against the
    vectors the REAL reference produced at that batch (tests/golden/b32_forward.npz)."""
    tag, net, extra, J, B, H, W = 'hrnet_w32_b32', 'pose_hrnet', configs.HRNET_W32, 17, 32, 256, 192
    g = gold_npz('b32_forward.npz')
    D, _, _ = build_states(net, extra, J)
    views, tgt, tw = synth_batch(tag, B, J, H, W)
    calibrate(net, D, views[2], extra)
    with torch.no_grad():
        ye = posenet_forward(net, D, views[0], extra, False)
        yt = posenet_forward(net, D, views[1], extra, True)
        loss = joints_loss(yt, tgt, tw, True)
    close(strided(ye), g[tag + '.eval_out'])
    close(strided(yt), g[tag + '.train_out'])
    close([float(loss)], g[tag + '.loss'], 1e-5, 1e-4)


BENCH_TILE_CASES = {      # oracle/gen_golden.py::BENCH_TILE_CASES - the other two benchmarked networks at benchmark-sized batches
    'resnet50_b32': ('pose_resnet', configs.RES50, 17, 32, 256, 192),
    'hrnet_w48_b16': ('pose_hrnet', configs.HRNET_W48, 17, 16, 384, 288),
    'hrnet_w48_b32': ('pose_hrnet', configs.HRNET_W48, 17, 32, 384, 288),      # C4's benchmarked batch itself (c4_b32_forward.npz, forward only)
}


@pytest.mark.parametrize('tag', sorted(BENCH_TILE_CASES))
def test_forward_of_the_other_benchmarked_networks_matches_reference(tag):
    """ResNet-50 256x192 at B = 32 (C2) and HRNet-W48 384x288 at B = 16 (C4): eval forward, train forward, loss and
    running statistics against the REAL reference's vectors at those batches (tests/golden/benchtiles_forward.npz)."""
    net, extra, J, B, H, W = BENCH_TILE_CASES[tag]
    g = gold_npz('c4_b32_forward.npz' if tag == 'hrnet_w48_b32' else 'benchtiles_forward.npz')
    D, _, _ = build_states(net, extra, J, unet_downs=5 if 'w48' in tag else 6)
    views, tgt, tw = synth_batch(tag, B, J, H, W)
    calibrate(net, D, views[2], extra)
    with torch.no_grad():
        ye = posenet_forward(net, D, views[0], extra, False)
        yt = posenet_forward(net, D, views[1], extra, True)
        loss = joints_loss(yt, tgt, tw, True)
    close(strided(ye), g[tag + '.eval_out'])
    close(strided(yt), g[tag + '.train_out'])
    close([float(loss)], g[tag + '.loss'], 1e-5, 1e-4)
    for key in g.files:
        if key.startswith(tag + '.bn.'):
            close(D[key[len(tag) + 4:]].detach().numpy(), g[key], 1e-4, 1e-3)


def test_c1_plain_loop_literally_matches_reference():
    """BASELINE.json configs[0] as written: pose_resnet50 256x192, J = 16 (MPII), B = 4, the plain ``train`` loop
    (function.py:30-95), two iterations of the REAL reference: losses, heat-maps, post-step parameter checksums."""
    tag, net, extra, J, B, H, W = 'c1_resnet50_j16_b4', 'pose_resnet', configs.RES50, 16, 4, 256, 192
    g, meta = gold_npz('c1_plain_steps.npz'), gold_json('c1_plain_checksums.json')[tag]
    D, _, _ = build_states(net, extra, J, salt=20)
    calibrate(net, D, synth_batch(tag + '.calib', B, J, H, W)[0][0], extra)
    optD = Adam(D, trainable(D))
    for it in range(2):
        v, t, w = synth_batch('%s.plain%d' % (tag, it), B, J, H, W)
        r = plain_step(net, extra, D, optD, v[0], t, w)
        close([float(r['loss'])], [g[tag + '.plain_losses'][it]], 1e-4, 1e-3)
        close(strided(r['out'], 2048), g['%s.out.it%d' % (tag, it)], 2e-3 if it else 1e-3, 2e-3 if it else 1e-3)
    assert int(D['bn1.num_batches_tracked']) == meta['nbt']
    checksum_close(checksum(D, meta['plain_D'].keys()), meta['plain_D'])


# ---- validation path: oracle/validate.py against the real reference's outputs ----------------

from oracle import validate as oval                                   # noqa: E402
from oracle.synth import synth_heatmaps, synth_boxes                  # noqa: E402

VAL_CASES = {
    'hrnet_tiny': ('pose_hrnet', configs.HRNET_TINY, 5, 3, 64, 64),
    'resnet18_tiny': ('pose_resnet', configs.RES18_TINY, 5, 3, 64, 64),
    'hrnet_w32': ('pose_hrnet', configs.HRNET_W32, 17, 2, 256, 192),
}


def test_final_preds_and_flip_back_match_reference():
    g, meta = gold_npz('validate.npz'), gold_json('validate.json')
    for i in range(3):
        B, J, H, W = meta['fp%d' % i]
        hm = synth_heatmaps('val.hm%d' % i, B, J, H, W)
        c, s, _ = synth_boxes('val.box%d' % i, B)
        for pp in (0, 1):
            preds, maxvals, _ = oval.get_final_preds(hm.copy(), c, s, bool(pp))
            assert preds.dtype == np.float32
            assert np.array_equal(preds, g['fp%d.pp%d.preds' % (i, pp)]), (i, pp)
            assert np.array_equal(maxvals, g['fp%d.pp%d.maxvals' % (i, pp)])
        assert (g['fp%d.pp1.preds' % i] != g['fp%d.pp0.preds' % i]).any()          # the shift did something
    for i in range(2):
        m = meta['fb%d' % i]
        x = detinit.normal('val.fb%d' % i, tuple(m['shape']), 1.0).numpy()
        assert np.array_equal(oval.flip_back(x, m['pairs']), g['fb%d' % i])


def device_formula_preds(coords, center, scale, W, H):
    """numpy statement of the closed form advmix_final_preds evaluates on the device (postproc.hip):
    with rot = 0 the three-point affine is axis-aligned, so the 6x6 solve collapses to two slopes."""
    f32, f64 = np.float32, np.float64
    out = np.zeros(coords.shape, dtype=np.float32)
    for b in range(coords.shape[0]):
        cx, cy = f32(center[b, 0]), f32(center[b, 1])
        sw = f32(scale[b, 0]) * f32(200.0)
        cy1 = f32(f64(cy) + f64(sw * f32(-0.5)))
        d = f32(cy - cy1)
        s2x = f32(cx + (-d))
        hw, hh = f64(W) * 0.5, f64(H) * 0.5
        mx, my = (f64(cx) - f64(s2x)) / hw, (f64(cy) - f64(cy1)) / hw
        tx, ty = f64(cx) - mx * hw, f64(cy) - my * hh
        out[b, :, 0] = (mx * coords[b, :, 0].astype(f64) + tx).astype(f32)
        out[b, :, 1] = (my * coords[b, :, 1].astype(f64) + ty).astype(f32)
    return out


def test_device_affine_closed_form_equals_the_three_point_solve():
    g, meta = gold_npz('validate.npz'), gold_json('validate.json')
    for i in range(3):
        B, J, H, W = meta['fp%d' % i]
        hm = synth_heatmaps('val.hm%d' % i, B, J, H, W)
        c, s, _ = synth_boxes('val.box%d' % i, B)
        _, _, coords = oval.get_final_preds(hm.copy(), c, s, True)
        want = g['fp%d.pp1.preds' % i]
        got = device_formula_preds(coords, c, s, W, H)
        ulp = np.spacing(np.abs(want).astype(np.float32))
        assert (np.abs(got.astype(np.float64) - want) <= ulp).all()                # at most 1 float32 ulp
        assert (got == want).mean() > 0.98


@pytest.mark.parametrize('tag', sorted(VAL_CASES))
@pytest.mark.parametrize('mode', ['plain', 'flip'])
def test_validate_loop_matches_reference(tag, mode):
    net, extra, J, B, H, W = VAL_CASES[tag]
    g, meta = gold_npz('validate.npz'), gold_json('validate.json')
    key = '%s.%s' % (tag, mode)
    m = meta[key]
    flip = mode == 'flip'
    D, _, _ = build_states(net, extra, J, salt=30)
    calibrate(net, D, synth_batch(tag + '.valcalib', B, J, H, W)[0][0], extra)
    outs, cs, ss, scores, losses = [], [], [], [], []
    for it in range(2):
        v, t, w = synth_batch('%s.val%d' % (tag, it), B, J, H, W)
        c, s, score = synth_boxes('%s.valbox%d' % (tag, it), B)
        out, loss, _, _ = oval.validate_batch(net, extra, D, v[0], t, w, m['pairs'], flip, flip)
        close(strided(torch.from_numpy(out), 2048), g['%s.out%d' % (key, it)], 1e-4, 1e-3)
        outs.append(out); cs.append(c); ss.append(s); scores.append(score); losses.append(loss)
    close(losses, g[key + '.losses'], 1e-5, 1e-4)
    assert abs(np.mean(losses) - m['loss_avg']) <= 1e-4 * max(1.0, abs(m['loss_avg']))
    all_preds, all_boxes = oval.collect(np.concatenate(outs), np.concatenate(cs), np.concatenate(ss),
                                        np.concatenate(scores), flip)
    assert np.array_equal(all_boxes, g[key + '.all_boxes'])
    want = g[key + '.all_preds']
    close(all_preds[:, :, 2], want[:, :, 2], 1e-4, 1e-3)
    same = np.abs(all_preds[:, :, 0:2] - want[:, :, 0:2]).max(axis=2) <= 1e-3
    assert same.mean() >= 0.9, same.mean()        # an argmax may flip on a near-tie between two fp32 forwards


def test_rescoring_and_oks_nms_match_reference():
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
        ids = [7 + n // per_img for n in range(N)]
        got = oval.rescore_and_nms(kp, boxes, ids, m['in_vis'], m['oks_thre'], m['soft'])
        flat = [[img, row, sc] for img, kept in got for row, sc in kept]
        want = g['oks%d.kept' % i]
        assert len(flat) == len(want) and 0 < len(flat) <= N, (i, len(flat), len(want))
        assert m['soft'] or len(flat) < N          # hard NMS suppressed someone; soft NMS only re-orders (<= 20 kept)
        assert [(int(a), int(b)) for a, b, _ in flat] == [(int(a), int(b)) for a, b, _ in want]
        assert np.allclose([f[2] for f in flat], want[:, 2], rtol=1e-12, atol=0)


# ---- input pipeline: oracle/inputpipe.py against the real grid_aug / generate_target ----------

from oracle import inputpipe as oip                                    # noqa: E402

INPUT_CASES = [('small', 4, 5, 64, 48, 16, 12), ('coco', 3, 17, 256, 192, 64, 48), ('w48', 2, 17, 384, 288, 96, 72)]


def test_gridmask_and_targets_match_reference():
    g, meta = gold_npz('inputpipe.npz'), gold_json('inputpipe.json')
    for tag, B, J, H, W, Hh, Wh in INPUT_CASES:
        base, aug, jt, vis = oip.synth_samples('inp.' + tag, B, J, H, W)
        assert meta[tag]['masked'] >= 1
        for b in range(B):
            np.random.seed(1000 + 17 * b + H)
            draws = oip.grid_draws(H, W, 0.5, 0.7, 1, np.random)
            want = meta[tag]['draws'][b]
            assert (draws is None and want is None) or list(draws) == want
            img = oip.to_tensor_normalize(base[b])
            out, vis_out, mask = oip.grid_aug(img, jt[b], vis[b], draws, J)
            kept = np.unpackbits(g['%s.mask%d' % (tag, b)])[:H * W].reshape(H, W).astype(bool)
            if draws is None:
                assert kept.all() and torch.equal(out, img)
            else:
                assert np.array_equal(mask.astype(bool), kept)
                assert torch.equal(out, img * torch.from_numpy(kept.astype(np.float32)))
                assert 0.2 < kept.mean() < 0.95
            assert np.array_equal(vis_out, g['%s.vis%d' % (tag, b)])
            for name, vv in (('clean', vis[b]), ('grid', vis_out)):
                tgt, tw = oip.generate_target(jt[b], vv, (W, H), (Wh, Hh), 2)
                assert np.array_equal(tgt, g['%s.%s.target%d' % (tag, name, b)])
                assert np.array_equal(tw, g['%s.%s.tw%d' % (tag, name, b)])
    jw = np.array([1., 1., 1., 1., 1., 1., 1., 1.2, 1.2, 1.5, 1.5, 1., 1., 1.2, 1.2, 1.5, 1.5], np.float32).reshape(17, 1)
    base, aug, jt, vis = oip.synth_samples('inp.coco', 3, 17, 256, 192)
    _, tw = oip.generate_target(jt[0], vis[0], (192, 256), (48, 64), 2, jw)
    assert np.array_equal(tw, g['coco.jw.tw0'])


# ---- AutoAugment view (SURVEY.md 8 f2 remainder): oracle/autoaug.py against Pillow and the real ImageNetPolicy -----

def test_autoaug_operations_match_pillow_bit_for_bit():
    """Every reachable operation at every magnitude the policy table uses (and the sharpness signs) against the
    installed Pillow on random, low-entropy and constant images, odd sizes included."""
    from PIL import Image, ImageOps, ImageEnhance
    from oracle import autoaug as oa
    rng = np.random.RandomState(3)
    imgs = [rng.randint(0, 256, (h, w, 3)).astype(np.uint8) for h, w in ((3, 3), (17, 31), (64, 48), (40, 57), (2, 9), (9, 1))]
    imgs.append(np.full((12, 10, 3), 200, np.uint8))
    imgs.append((rng.randint(0, 256, (30, 30, 3)) // 64 * 64).astype(np.uint8))
    imgs.append(np.minimum(rng.randint(0, 256, (64, 64, 3)), 40).astype(np.uint8))
    for a in imgs:
        im = Image.fromarray(a)
        assert np.array_equal(oa.apply_op(a, oa.EQUALIZE, 0), np.array(ImageOps.equalize(im)))
        assert np.array_equal(oa.apply_op(a, oa.INVERT, 0), np.array(ImageOps.invert(im)))
        for idx in range(10):
            bits = oa.magnitude('posterize', idx)
            assert np.array_equal(oa.apply_op(a, oa.POSTERIZE, bits), np.array(ImageOps.posterize(im, bits)))
            th = oa.magnitude('solarize', idx)
            assert np.array_equal(oa.apply_op(a, oa.SOLARIZE, th), np.array(ImageOps.solarize(im, th)))
            for sign in (-1, 1):
                f = 1 + oa.magnitude('sharpness', idx) * sign
                assert np.array_equal(oa.apply_op(a, oa.SHARPNESS, f), np.array(ImageEnhance.Sharpness(im).enhance(f))), (a.shape, f)


def test_autoaug_matches_the_real_policy_and_replays_its_draws():
    """tests/golden/autoaug.*: outputs of the REAL ImageNetPolicy (advaug.py:10-108) for seeded ``random`` states; the
    restated draw order reproduces the recorded draws from the same seeds, the restated operations the recorded bytes."""
    import random
    import zlib
    from oracle import autoaug as oa
    from oracle import inputpipe as ip
    from oracle.gen_golden import AUTOAUG_CASES
    meta, g = gold_json('autoaug.json'), gold_npz('autoaug.npz')
    seen = set()
    for tag, B, H, W in AUTOAUG_CASES:
        base, _, _, _ = ip.synth_samples('aa.' + tag, B, 1, H, W)
        if tag == 'small':
            base[0] = 77
            base[1] = (base[1] // 128) * 200
            base[2, :, :, 1] = base[2, :, :, 0] // 64 * 60
        for b in range(B):
            random.seed(4242 + 31 * b + H)
            ops = oa.draw_policy(random)
            assert [[int(c), float(p)] for c, p in ops] == meta[tag]['draws'][b], (tag, b)
            out = oa.autoaug(base[b].astype(np.uint8), ops)
            assert zlib.crc32(np.ascontiguousarray(out).tobytes()) == meta[tag]['crc32'][b], (tag, b, ops)
            if tag == 'odd':
                assert np.array_equal(out, g['%s.out%d' % (tag, b)])
            seen.update(c for c, _ in ops)
    assert seen == {1, 2, 3, 4, 5}
