#!/usr/bin/env python3
"""Generate tests/golden/* by running the REAL reference (imported from
/root/reference, build container only) on seeded inputs.

    python -m oracle.gen_golden            # from the repo root

The reference never travels: only the small input/output vectors written here
are committed.  Shims: cv2 / torchvision stubbed in
sys.modules (only touched by debug-image code that is never reached),
``Tensor.cuda`` patched to identity, a dict-with-attributes cfg.
Weights come from oracle.detinit (name-keyed Philox), not torch RNG, so the
tests can rebuild them without storing 100 MB state dicts.
"""
import json
import os
import sys
import types
import logging

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')

from oracle import detinit, configs                       # noqa: E402
from oracle.posenet import posenet_spec                   # noqa: E402
from oracle.unet import unet_spec, unet_transposed_names  # noqa: E402
from oracle.synth import synth_batch, strided, checksum, synth_heatmaps, synth_boxes   # noqa: E402


class AD(dict):
    """dict with attribute access (ResNet uses cfg.MODEL.EXTRA.X, HRNet cfg['MODEL'])."""
    __getattr__ = dict.__getitem__

    @staticmethod
    def wrap(d):
        if isinstance(d, dict):
            return AD({k: AD.wrap(v) for k, v in d.items()})
        return d


def import_reference():
    sys.path.insert(0, os.path.join(REF, 'lib'))
    for m in ('cv2', 'torchvision', 'torchvision.utils'):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules['torchvision'].utils = sys.modules['torchvision.utils']
    for m in ('nms.cpu_nms', 'nms.gpu_nms'):              # dead native code
        mod = types.ModuleType(m)
        setattr(mod, m.split('.')[1], None)
        sys.modules[m] = mod
    torch.Tensor.cuda = lambda self, *a, **k: self
    import models.pose_hrnet, models.pose_resnet, models.Unet_generator   # noqa
    import core.loss, core.function, core.evaluate                        # noqa
    import nms.nms                                                        # noqa
    return sys.modules


def make_cfg(name, extra, J):
    return AD.wrap({'MODEL': {'NAME': name, 'EXTRA': extra, 'NUM_JOINTS': J,
                              'INIT_WEIGHTS': False, 'PRETRAINED': ''},
                    'PRINT_FREQ': 1000000, 'DEBUG': {'DEBUG': False}})


def build_ref_models(M, net, extra, J, unet_downs, salt=0):
    cfg = make_cfg(net, extra, J)
    mod = M['models.' + net]
    D = mod.get_pose_net(cfg, is_train=False)
    Tm = mod.get_pose_net(cfg, is_train=False)
    G = M['models.Unet_generator'].UnetGenerator(9, 3, unet_downs)
    dspec = posenet_spec(net, extra, J)
    gspec = unet_spec(9, 3, unet_downs)
    detinit.mark_transposed(unet_transposed_names(9, 3, unet_downs))
    sdD = detinit.fill_state_dict(dspec, salt=salt)
    sdT = detinit.fill_state_dict(dspec, salt=salt + 1)
    sdG = detinit.fill_state_dict(gspec, salt=salt + 2, gain=0.5)
    D.load_state_dict(sdD, strict=True)
    Tm.load_state_dict(sdT, strict=True)
    G.load_state_dict(sdG, strict=True)
    return cfg, D, G, Tm


def calibrate_ref(model, x):
    """One train-mode pass with BN momentum 1.0 (mirrors oracle.posenet.calibrate)."""
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    old = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(x)
    for m, o in zip(bns, old):
        m.momentum = o


def gen_keys(M):
    out = {}
    for tag, net, extra, J in (('hrnet_w32', 'pose_hrnet', configs.HRNET_W32, 17),
                               ('hrnet_w48', 'pose_hrnet', configs.HRNET_W48, 17),
                               ('resnet50', 'pose_resnet', configs.RES50, 17),
                               ('hrnet_tiny', 'pose_hrnet', configs.HRNET_TINY, 5),
                               ('resnet18_tiny', 'pose_resnet', configs.RES18_TINY, 5)):
        m = M['models.' + net].get_pose_net(make_cfg(net, extra, J), is_train=False)
        out[tag] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    for downs in (5, 6):
        g = M['models.Unet_generator'].UnetGenerator(9, 3, downs)
        out['unet%d' % downs] = [[k, list(v.shape)] for k, v in g.state_dict().items()]
    with open(os.path.join(OUT, 'state_dict_keys.json'), 'w') as f:
        json.dump(out, f)


def gen_loss(M):
    L = M['core.loss'].JointsMSELoss
    o = (torch.arange(48, dtype=torch.float32) / 8 - 1.5).reshape(2, 3, 4, 2)
    t = torch.zeros(2, 3, 4, 2)
    t[:, :, 1, 1] = 1
    w = torch.tensor([[1, 0, 1], [1, 1, .5]]).reshape(2, 3, 1)
    kat = {'smoothl1_w': float(L(True)(o, t, w)), 'mse_w': float(L(True, True)(o, t, w)),
           'smoothl1_now': float(L(False)(o, t, w)), 'b1': float(L(True)(o[:1], t[:1], w[:1]))}
    rnd = {}
    for i, (B, J, H, W, sc) in enumerate([(4, 17, 64, 48, 1.0), (3, 16, 8, 8, 3.0), (2, 5, 16, 16, 0.3)]):
        o = detinit.normal('loss.o%d' % i, (B, J, H, W), sc)
        t = detinit.uniform('loss.t%d' % i, (B, J, H, W))
        w = (detinit.uniform('loss.w%d' % i, (B, J, 1)) < 0.7).float()
        o.requires_grad_(True)
        v = L(True)(o, t, w)
        v.backward()
        rnd['case%d' % i] = {'shape': [B, J, H, W], 'scale': sc, 'loss': float(v),
                             'grad_abs_sum': float(o.grad.double().abs().sum()),
                             'grad_sample': strided(o.grad, 64).tolist()}
    with open(os.path.join(OUT, 'loss_kat.json'), 'w') as f:
        json.dump({'kat': kat, 'random': rnd}, f)


FORWARD_CASES = (('hrnet_tiny', 'pose_hrnet', configs.HRNET_TINY, 5, 2, 64, 64),
                 ('resnet18_tiny', 'pose_resnet', configs.RES18_TINY, 5, 2, 64, 64),
                 ('hrnet_w32', 'pose_hrnet', configs.HRNET_W32, 17, 2, 256, 192),
                 ('resnet50', 'pose_resnet', configs.RES50, 17, 2, 256, 192))
# BASELINE.json configs[3] (C4): HRNet-W48 384x288 with UnetGenerator(9, 3, 5) (tools/_init_parse.py:132-134)
C4_CASE = ('hrnet_w48', 'pose_hrnet', configs.HRNET_W48, 17, 2, 384, 288)
# the benchmarked batch: the tile configurations conv_direct picks at B = 32 differ from those at B = 2
B32_CASE = ('hrnet_w32_b32', 'pose_hrnet', configs.HRNET_W32, 17, 32, 256, 192)


def gen_forward(M, cases=FORWARD_CASES, fname='forward.npz'):
    """Per-model forward/backward vectors (train + eval mode)."""
    res = {}
    for tag, net, extra, J, B, H, W in cases:
        cfg, D, G, _ = build_ref_models(M, net, extra, J, 6 if H % 64 == 0 and W % 64 == 0 else 5)
        views, tgt, tw = synth_batch(tag, B, J, H, W)
        calibrate_ref(D, views[2])
        D.eval()
        with torch.no_grad():
            ye = D(views[0])
        D.train()
        x = views[1].clone().requires_grad_(True)
        yt = D(x)
        loss = M['core.loss'].JointsMSELoss(True)(yt, tgt, tw)
        loss.backward()
        sd = D.state_dict()
        gnames = [k for k, p in D.named_parameters()]
        pick = gnames[:3] + gnames[len(gnames) // 2:len(gnames) // 2 + 3] + gnames[-4:]
        res[tag + '.eval_out'] = strided(ye)
        res[tag + '.train_out'] = strided(yt)
        res[tag + '.loss'] = np.array([float(loss)])
        res[tag + '.dx'] = strided(x.grad)
        for k in pick:
            g = dict(D.named_parameters())[k].grad
            res['%s.grad.%s' % (tag, k)] = np.array([float(g.double().sum()), float(g.double().abs().sum())])
        bn = [k for k in sd if k.endswith('running_mean')]
        for k in (bn[0], bn[len(bn) // 2], bn[-1]):
            res['%s.bn.%s' % (tag, k)] = sd[k].numpy().copy()
            kv = k.replace('running_mean', 'running_var')
            res['%s.bn.%s' % (tag, kv)] = sd[kv].numpy().copy()
        # generator forward + backward
        gi = torch.cat(views, 1)
        lg = G(gi)
        res[tag + '.unet_out'] = strided(lg)
        (lg * detinit.normal(tag + '.gproj', lg.shape)).sum().backward()
        gp = dict(G.named_parameters())
        for k in list(gp)[:2] + list(gp)[-2:]:
            res['%s.ggrad.%s' % (tag, k)] = np.array([float(gp[k].grad.double().sum()),
                                                       float(gp[k].grad.double().abs().sum())])
        print('forward', tag, float(loss), flush=True)
    np.savez_compressed(os.path.join(OUT, fname), **res)


class Rec:
    """criterion wrapper that records every call's value (loss_D_hm, loss_D_kd, loss_G)."""

    def __init__(self, fn):
        self.fn, self.vals = fn, []

    def __call__(self, *a):
        v = self.fn(*a)
        self.vals.append(float(v))
        return v

    def cuda(self):
        return self


ADVMIX_CASES = (('hrnet_tiny', 'pose_hrnet', configs.HRNET_TINY, 5, 2, 64, 64, 3),
                ('resnet18_tiny', 'pose_resnet', configs.RES18_TINY, 5, 2, 64, 64, 3),
                ('hrnet_w32', 'pose_hrnet', configs.HRNET_W32, 17, 2, 256, 192, 2),
                ('resnet50', 'pose_resnet', configs.RES50, 17, 2, 256, 192, 2))


def gen_advmix(M, cases=ADVMIX_CASES, fname='advmix_steps.npz', jname='advmix_checksums.json', downs=6, plain=True):
    """2-3 iterations of the REAL train_advmix / train loops on tiny + full models."""
    fn = M['core.function']
    res, meta = {}, {}
    for tag, net, extra, J, B, H, W, iters in cases:
        cfg, D, G, Tm = build_ref_models(M, net, extra, J, downs, salt=10)
        args = AD(alpha=0.1, adv_loss_weight=1.0)
        calib = synth_batch(tag + '.calib', B, J, H, W)[0][0]
        calibrate_ref(Tm, calib)
        calibrate_ref(D, calib)
        optD = torch.optim.Adam(D.parameters(), lr=1e-3)      # utils.py:89-92
        optG = torch.optim.Adam(G.parameters(), lr=1e-3)
        crit = Rec(M['core.loss'].JointsMSELoss(True))
        batches = []
        for it in range(iters):
            v, t, w = synth_batch('%s.it%d' % (tag, it), B, J, H, W)
            batches.append((v, [t, t, t], [w, w, w], [{}, {}, {}]))
        outs = []
        hook = D.register_forward_hook(lambda m, i, o: outs.append(o.detach().clone()))
        wd = {'writer': types.SimpleNamespace(add_scalar=lambda *a, **k: None), 'train_global_steps': 0}
        cfg['PRINT_FREQ'] = 10 ** 9
        # PRINT_FREQ huge: i % PRINT_FREQ == 0 still fires for i == 0 -> logging only
        fn.save_debug_images = lambda *a, **k: None
        fn.train_advmix(cfg, args, batches, [D, G, Tm], crit, [optD, optG], 0, '/tmp', '/tmp', wd)
        hook.remove()
        res[tag + '.losses'] = np.array(crit.vals).reshape(iters, 3)     # hm, kd, G(pos) per iter
        for it in range(iters):
            res['%s.out1.it%d' % (tag, it)] = strided(outs[2 * it], 2048)
            res['%s.out2.it%d' % (tag, it)] = strided(outs[2 * it + 1], 2048)
        sdD, sdG = D.state_dict(), G.state_dict()
        meta[tag] = {'D': checksum(sdD, [k for k in sdD if sdD[k].is_floating_point()]),
                     'G': checksum(sdG, list(sdG)),
                     'nbt': int(sdD['bn1.num_batches_tracked'])}
        print('advmix', tag, res[tag + '.losses'].tolist(), flush=True)

        if not plain:
            continue
        # plain (non-AdvMix) loop, function.py:30-95
        cfg, D, _, _ = build_ref_models(M, net, extra, J, downs, salt=20)
        calibrate_ref(D, calib)
        optD = torch.optim.Adam(D.parameters(), lr=1e-3)
        crit = Rec(M['core.loss'].JointsMSELoss(True))
        pb = []
        for it in range(2):
            v, t, w = synth_batch('%s.plain%d' % (tag, it), B, J, H, W)
            pb.append((v[0], [t, t], w, {}))
        fn._tocuda = lambda x: x
        fn.train(cfg, None, pb, D, crit, optD, 0, '/tmp', '/tmp', wd)
        res[tag + '.plain_losses'] = np.array(crit.vals)
        sdD = D.state_dict()
        meta[tag]['plain_D'] = checksum(sdD, [k for k in sdD if sdD[k].is_floating_point()])
        print('plain', tag, crit.vals, flush=True)
    np.savez_compressed(os.path.join(OUT, fname), **res)
    with open(os.path.join(OUT, jname), 'w') as f:
        json.dump(meta, f)


def gen_c4(M):
    """C4 (HRNet-W48 384x288 + UnetGenerator(9,3,5)): forward/backward vectors and the real train_advmix loop."""
    gen_forward(M, (C4_CASE,), 'c4_forward.npz')
    gen_advmix(M, (C4_CASE + (2,),), 'c4_advmix_steps.npz', 'c4_advmix_checksums.json', downs=5, plain=False)


def gen_b32(M):
    """HRNet-W32 256x192 at the benchmarked batch (B = 32): forward/backward vectors and ONE real train_advmix
    iteration, so the tiles conv_direct dispatches at B = 32 are covered at network level."""
    gen_forward(M, (B32_CASE,), 'b32_forward.npz')
    gen_advmix(M, (B32_CASE + (1,),), 'b32_advmix_steps.npz', 'b32_advmix_checksums.json', plain=False)


# every benchmarked network at a batch that reaches the tile configurations conv_direct dispatches at B = 32
# (tests/test_models_gpu.py asserts the configuration sets): ResNet-50 256x192 at B = 32 itself, HRNet-W48 384x288 at
# B = 16 (configurations {2, 3, 6} like B = 32; the CPU oracle at B = 32 would need ~40 GB)
BENCH_TILE_CASES = (('resnet50_b32', 'pose_resnet', configs.RES50, 17, 32, 256, 192),
                    ('hrnet_w48_b16', 'pose_hrnet', configs.HRNET_W48, 17, 16, 384, 288))
# BASELINE.json configs[0] (C1), literally: pose_resnet50 256x192, MPII's 16 joints, batch 4, the plain ``train`` loop
C1_CASE = ('c1_resnet50_j16_b4', 'pose_resnet', configs.RES50, 16, 4, 256, 192)


# BASELINE.json configs[4] (C5) as far as the reference goes: it has no HigherHRNet code, but its pose_hrnet and
# UnetGenerator run at 512x512 - the shapes of that configuration (128x128x32 ... 16x16x256 branch maps)
C5_TRUNK_CASE = ('hrnet_w32_512', 'pose_hrnet', configs.HRNET_W32, 17, 2, 512, 512)


def gen_benchtiles(M):
    gen_forward(M, BENCH_TILE_CASES, 'benchtiles_forward.npz')


# C4 at ITS benchmarked batch (VERDICT r4 item 6 a): HRNet-W48 384x288, B = 32 - forward only (eval forward, train forward,
# loss, running statistics under no_grad: the backward of gen_forward would need ~40 GB here)
C4_B32_CASE = ('hrnet_w48_b32', 'pose_hrnet', configs.HRNET_W48, 17, 32, 384, 288)


def gen_c4b32(M):
    tag, net, extra, J, B, H, W = C4_B32_CASE
    cfg, D, G, _ = build_ref_models(M, net, extra, J, 5)
    views, tgt, tw = synth_batch(tag, B, J, H, W)
    calibrate_ref(D, views[2])
    res = {}
    with torch.no_grad():
        D.eval()
        res[tag + '.eval_out'] = strided(D(views[0]))
        D.train()
        yt = D(views[1])
        loss = M['core.loss'].JointsMSELoss(True)(yt, tgt, tw)
    res[tag + '.train_out'] = strided(yt)
    res[tag + '.loss'] = np.array([float(loss)])
    sd = D.state_dict()
    bn = [k for k in sd if k.endswith('running_mean')]
    for k in (bn[0], bn[len(bn) // 2], bn[-1]):
        res['%s.bn.%s' % (tag, k)] = sd[k].numpy().copy()
        kv = k.replace('running_mean', 'running_var')
        res['%s.bn.%s' % (tag, kv)] = sd[kv].numpy().copy()
    print('forward', tag, float(loss), flush=True)
    np.savez_compressed(os.path.join(OUT, 'c4_b32_forward.npz'), **res)


def gen_c4b32step(M):
    """C4 at ITS benchmarked batch, the STEP (VERDICT r5 weak 1 c): ONE iteration of the real train_advmix on HRNet-W48 384x288,
    B = 32, UnetGenerator(9, 3, 5) - losses, both D outputs (strided samples) and the parameter checksums after the update.
    ~45 GB of host memory (run it alone: ``ulimit -v`` below the container's 62 GB turns an overrun into a MemoryError)."""
    gen_advmix(M, (C4_B32_CASE + (1,),), 'c4_b32_advmix_steps.npz', 'c4_b32_advmix_checksums.json', downs=5, plain=False)


def gen_c5trunk(M):
    gen_forward(M, (C5_TRUNK_CASE,), 'c5_trunk_forward.npz')


def gen_c1(M):
    """C1: two iterations of the REAL ``train`` loop (function.py:30-95) at J = 16, B = 4."""
    fn = M['core.function']
    tag, net, extra, J, B, H, W = C1_CASE
    cfg, D, _, _ = build_ref_models(M, net, extra, J, 6, salt=20)
    calib = synth_batch(tag + '.calib', B, J, H, W)[0][0]
    calibrate_ref(D, calib)
    optD = torch.optim.Adam(D.parameters(), lr=1e-3)
    crit = Rec(M['core.loss'].JointsMSELoss(True))
    pb, outs = [], []
    for it in range(2):
        v, t, w = synth_batch('%s.plain%d' % (tag, it), B, J, H, W)
        pb.append((v[0], [t, t], w, {}))
    hook = D.register_forward_hook(lambda m, i, o: outs.append(o.detach().clone()))
    wd = {'writer': types.SimpleNamespace(add_scalar=lambda *a, **k: None), 'train_global_steps': 0}
    cfg['PRINT_FREQ'] = 10 ** 9
    fn.save_debug_images = lambda *a, **k: None
    fn.train(cfg, None, pb, D, crit, optD, 0, '/tmp', '/tmp', wd)
    hook.remove()
    res = {tag + '.plain_losses': np.array(crit.vals)}
    for it in range(2):
        res['%s.out.it%d' % (tag, it)] = strided(outs[it], 2048)
    sdD = D.state_dict()
    meta = {tag: {'plain_D': checksum(sdD, [k for k in sdD if sdD[k].is_floating_point()]),
                  'nbt': int(sdD['bn1.num_batches_tracked'])}}
    np.savez_compressed(os.path.join(OUT, 'c1_plain_steps.npz'), **res)
    with open(os.path.join(OUT, 'c1_plain_checksums.json'), 'w') as f:
        json.dump(meta, f)
    print('c1', crit.vals, flush=True)


def gen_nms(M):
    nm = M['nms.nms']
    rng = np.random.Generator(np.random.Philox(key=77))
    box, oks = {}, {}
    for N in (1, 2, 63, 64, 65, 200, 1000):
        for th in (0.3, 0.5, 0.7):
            c = rng.random((N, 2)) * 200
            wh = rng.random((N, 2)) * 80 + 4
            sc = rng.permutation(N).astype(np.float32) / N + 0.001          # distinct scores
            d = np.concatenate([c, c + wh, sc[:, None]], 1).astype(np.float32)
            keep = [int(i) for i in nm.nms(d, th)]
            box['N%d_t%g' % (N, th)] = {'dets': d.tolist() if N <= 65 else None, 'seed_N': N,
                                        'thresh': th, 'keep': keep}
            if N > 65:
                np.save(os.path.join(OUT, 'nms_dets_N%d_t%g.npy' % (N, th)), d)
    for N in (1, 5, 20, 40):
        for th in (0.5, 0.9):
            k = rng.random((N, 17, 3)) * 100
            base = rng.random((1, 17, 3)) * 100
            k[: N // 2] = base + rng.normal(0, 2.0, (N // 2, 17, 3))          # near-duplicates
            db = [{'score': float(s), 'keypoints': kk, 'area': float(a)}
                  for s, kk, a in zip(rng.permutation(N) / N + 0.01, k, rng.random(N) * 4000 + 500)]
            oks['N%d_t%g' % (N, th)] = {
                'score': [e['score'] for e in db], 'area': [e['area'] for e in db],
                'kpts': k.tolist(), 'thresh': th,
                'keep': [int(i) for i in nm.oks_nms(db, th)],
                'soft_keep': [int(i) for i in nm.soft_oks_nms(db, th)]}
    with open(os.path.join(OUT, 'nms.json'), 'w') as f:
        json.dump({'box': box, 'oks': oks}, f)


COCO_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]       # coco.py:71-72 (data)


def gen_nmsvis(M):
    """oks_iou / oks_nms / soft_oks_nms WITH ``in_vis_thre`` (nms.py:90-92: only the joints whose visibility exceeds the
    threshold count - through ``list(vg > t) and list(vd > t)``, i.e. the DETECTION's visibilities alone decide).  The
    reference's caller never passes it (coco.py:356-364); the argument is part of the interface."""
    nm = M['nms.nms']
    rng = np.random.Generator(np.random.Philox(key=78))
    out = {}
    for N in (1, 6, 24):
        for th, vis in ((0.5, 0.2), (0.9, 0.5), (0.9, 0.97), (0.5, 2.0)):      # 2.0: no joint passes -> every OKS is 0
            k = rng.random((N, 17, 3)) * 100
            base = rng.random((1, 17, 3)) * 100
            k[: N // 2] = base + rng.normal(0, 2.0, (N // 2, 17, 3))
            k[:, :, 2] = rng.random((N, 17))                                     # visibilities in [0, 1)
            if N >= 6:
                k[1, :, 2] = 0.0                                                 # a person with no visible joint
                k[2, :9, 2] = 1.0                                                # >= 8 visible joints: numpy's 8-way sum
            db = [{'score': float(s), 'keypoints': kk, 'area': float(a)}
                  for s, kk, a in zip(rng.permutation(N) / N + 0.01, k, rng.random(N) * 4000 + 500)]
            kf = np.array([e['keypoints'].flatten() for e in db])
            ar = np.array([e['area'] for e in db])
            out['N%d_t%g_v%g' % (N, th, vis)] = {
                'score': [e['score'] for e in db], 'area': [e['area'] for e in db], 'kpts': k.tolist(), 'thresh': th,
                'in_vis_thre': vis,
                'iou_row0': [float(v) for v in nm.oks_iou(kf[0], kf, ar[0], ar, None, vis)],
                'keep': [int(i) for i in nm.oks_nms(db, th, None, vis)],
                'soft_keep': [int(i) for i in nm.soft_oks_nms(db, th, None, vis)]}
    with open(os.path.join(OUT, 'nms_vis.json'), 'w') as f:
        json.dump(out, f)


def accuracy_cases():
    """Seeded (outputs, target) pairs for the accuracy fixture - shared by the generator and the tests."""
    cases = {}
    o = detinit.normal('acc.o0', (4, 17, 64, 48))
    t = torch.zeros(4, 17, 64, 48)
    cx = (detinit.uniform('acc.cx0', (4, 17)) * 48).long().clamp(0, 47)
    cy = (detinit.uniform('acc.cy0', (4, 17)) * 64).long().clamp(0, 63)
    for b in range(4):
        for j in range(17):
            t[b, j, cy[b, j], cx[b, j]] = 1.0
            if (b + j) % 3 == 0:                           # prediction on / near the target
                o[b, j, cy[b, j], min(47, int(cx[b, j]) + (j % 2))] = 9.0
    cases['coco'] = (o, t)
    o = detinit.normal('acc.o1', (2, 5, 8, 8))
    t = detinit.uniform('acc.t1', (2, 5, 8, 8))
    o[0, 0] = 0.25                                          # constant map: ties -> first index
    o[0, 1] = -o[0, 1].abs() - 0.1                          # nothing positive: the prediction is zeroed (inference.py:44-47)
    t[0, 2] = 0.0
    t[0, 2, 0, 5] = 1.0                                     # target in row 0: invalid (evaluate.py:21)
    t[1, 2] = 0.0
    t[1, 2, 1, 1] = 1.0                                     # (1, 1): not > 1 either
    t[:, 3] = 0.0                                           # a joint nobody annotates: all -1 -> acc -1, not counted
    t[0, 4] = 0.0
    t[0, 4, 2, 2] = 1.0
    o[0, 4] = 0.0
    o[0, 4, 2, 2] = 1.0                                     # exact hit
    cases['edge'] = (o, t)
    return cases


def gen_accuracy(M):
    """evaluate.accuracy (evaluate.py:41-99) from the REAL reference on the seeded cases above."""
    ev = M['core.evaluate']
    out = {}
    for name, (o, t) in accuracy_cases().items():
        acc, avg, cnt, pred = ev.accuracy(o.clone(), t.clone())
        out[name] = {'acc': [float(v) for v in acc], 'avg': float(avg), 'cnt': int(cnt), 'pred': pred.tolist()}
        print('accuracy', name, avg, cnt, flush=True)
    with open(os.path.join(OUT, 'accuracy_kat.json'), 'w') as f:
        json.dump(out, f)


def gen_validate(M):
    """Validation path from the REAL reference: get_final_preds / flip_back on synthetic maps,
    the ``validate`` loop on tiny + full nets, COCODataset.evaluate's rescoring + OKS-NMS.  The one missing
    third-party piece, cv2.getAffineTransform, is the restatement in oracle/validate.py (see its header)."""
    from oracle import validate as ov
    sys.modules['cv2'].getAffineTransform = ov.cv_get_affine_transform
    import utils.transforms as T
    T.cv2 = sys.modules['cv2']
    inf, fn = M['core.inference'], M['core.function']
    torch.cuda.synchronize = lambda *a, **k: None
    res, meta = {}, {}

    # (1) get_final_preds on synthetic maps
    for i, (B, J, H, W) in enumerate([(4, 17, 64, 48), (3, 5, 16, 16), (2, 17, 96, 72)]):
        hm = synth_heatmaps('val.hm%d' % i, B, J, H, W)
        c, s, _ = synth_boxes('val.box%d' % i, B)
        for pp in (0, 1):
            cfg = AD.wrap({'TEST': {'POST_PROCESS': bool(pp)}, 'MODEL': {'IMAGE_SIZE': [W * 4, H * 4]}})
            preds, maxvals = inf.get_final_preds(cfg, None, hm.copy(), c, s)
            res['fp%d.pp%d.preds' % (i, pp)] = preds
            res['fp%d.pp%d.maxvals' % (i, pp)] = maxvals
        meta['fp%d' % i] = [B, J, H, W]

    # (2) flip_back
    for i, (B, J, H, W, pairs) in enumerate([(2, 17, 16, 12, COCO_PAIRS), (3, 5, 8, 8, [[0, 3], [1, 2]])]):
        x = detinit.normal('val.fb%d' % i, (B, J, H, W), 1.0).numpy()
        res['fb%d' % i] = T.flip_back(x.copy(), pairs)
        meta['fb%d' % i] = {'shape': [B, J, H, W], 'pairs': pairs}

    # (3) the validate loop
    for tag, net, extra, J, B, H, W, pairs in (
            ('hrnet_tiny', 'pose_hrnet', configs.HRNET_TINY, 5, 3, 64, 64, [[0, 3], [1, 2]]),
            ('resnet18_tiny', 'pose_resnet', configs.RES18_TINY, 5, 3, 64, 64, [[0, 3], [1, 2]]),
            ('hrnet_w32', 'pose_hrnet', configs.HRNET_W32, 17, 2, 256, 192, COCO_PAIRS)):
        for mode, (flip, shift, pp) in (('plain', (False, False, False)), ('flip', (True, True, True))):
            cfg, D, _, _ = build_ref_models(M, net, extra, J, 6, salt=30)
            calibrate_ref(D, synth_batch(tag + '.valcalib', B, J, H, W)[0][0])
            cfg['TEST'] = AD(FLIP_TEST=flip, SHIFT_HEATMAP=shift, POST_PROCESS=pp)
            cfg['MODEL']['IMAGE_SIZE'] = [W, H]
            cfg['PRINT_FREQ'] = 10 ** 9
            batches, paths = [], []
            for it in range(2):
                v, t, w = synth_batch('%s.val%d' % (tag, it), B, J, H, W)
                c, s, score = synth_boxes('%s.valbox%d' % (tag, it), B)
                names = ['img/%012d.jpg' % (100 + (it * B + k) // 2) for k in range(B)]
                batches.append((v[0], [t, t], w, {'center': torch.from_numpy(c), 'scale': torch.from_numpy(s),
                                                  'score': torch.from_numpy(score), 'image': names}))
            seen = {}

            class DS:
                flip_pairs = pairs

                def __len__(self):
                    return 2 * B

                def evaluate(self, cfg_, preds, out_dir, all_boxes, img_path, *a, **k):
                    seen['preds'], seen['boxes'], seen['paths'] = preds.copy(), all_boxes.copy(), list(img_path)
                    return {'AP': 0.0}, 0.0

            crit = Rec(M['core.loss'].JointsMSELoss(True))
            outs = []
            inner = crit.fn

            def rec_out(o, t, w, inner=inner, outs=outs):
                outs.append(o.detach().clone())
                return inner(o, t, w)
            crit.fn = rec_out
            scal = {}
            wd = {'writer': types.SimpleNamespace(add_scalar=lambda k, v, g: scal.__setitem__(k, float(v)),
                                                  add_scalars=lambda *a, **k: None),
                  'valid_global_steps': 0}
            fn.save_debug_images = lambda *a, **k: None
            fn._tocuda = lambda x: x
            fn.validate(cfg, None, batches, DS(), D, crit, '/tmp', '/tmp', wd)
            key = '%s.%s' % (tag, mode)
            res[key + '.all_preds'] = seen['preds']
            res[key + '.all_boxes'] = seen['boxes']
            res[key + '.losses'] = np.array(crit.vals)
            for it in range(2):
                res['%s.out%d' % (key, it)] = strided(outs[it], 2048)
            meta[key] = {'loss_avg': scal['valid_loss'], 'acc_avg': scal['valid_acc'], 'paths': seen['paths'],
                         'B': B, 'pairs': pairs}
            print('validate', key, crit.vals, scal, flush=True)

    # (4) COCODataset.evaluate: rescoring + OKS-NMS (coco.py:318-371), constructor bypassed
    for m, attrs in (('pycocotools', {}), ('pycocotools.coco', {'COCO': None}),
                     ('pycocotools.cocoeval', {'COCOeval': None}), ('json_tricks', {}),
                     ('imagecorruptions', {'corrupt': None, 'get_corruption_names': None})):
        mod = types.ModuleType(m)
        for k, v in attrs.items():
            setattr(mod, k, v)
        sys.modules.setdefault(m, mod)
    pkg = types.ModuleType('dataset')
    pkg.__path__ = [os.path.join(REF, 'lib', 'dataset')]
    sys.modules['dataset'] = pkg
    import dataset.coco as rc
    for i, (N, per_img, soft, in_vis, oks_thre) in enumerate([(24, 6, False, 0.2, 0.6), (24, 6, True, 0.2, 0.6),
                                                             (9, 3, False, 0.0, 0.5)]):
        J = 17
        base = detinit.uniform('val.oks%d.base' % i, (N // per_img, J, 2)).numpy() * 200 + 50
        jit = detinit.normal('val.oks%d.jit' % i, (N, J, 2), 3.0).numpy()
        kp = np.zeros((N, J, 3), dtype=np.float32)
        kp[:, :, 0:2] = base[np.arange(N) // per_img] + jit
        kp[:, :, 2] = detinit.uniform('val.oks%d.conf' % i, (N, J)).numpy()
        boxes = np.zeros((N, 6))
        boxes[:, 4] = detinit.uniform('val.oks%d.area' % i, (N,)).numpy().astype(np.float64) * 20000 + 5000
        boxes[:, 5] = detinit.uniform('val.oks%d.score' % i, (N,)).numpy().astype(np.float64)
        paths = ['img/%012d.jpg' % (7 + n // per_img) for n in range(N)]
        ds = object.__new__(rc.COCODataset)
        ds.test_robust, ds.corruption_type, ds.image_set = False, 'clean', 'test-dev2017'
        ds.num_joints, ds.in_vis_thre, ds.oks_thre, ds.soft_nms = J, in_vis, oks_thre, soft
        got = {}
        ds._write_coco_keypoint_results = lambda kpts, f, got=got: got.__setitem__('k', kpts)
        ds.evaluate(AD(RANK=0), kp.copy(), '/tmp/advmix_golden_eval', boxes.copy(), paths)
        kept = []
        for img_kpts in got['k']:
            for person in img_kpts:
                row = int(np.where((kp == person['keypoints']).all(axis=(1, 2)))[0][0])
                kept.append([int(person['image']), row, float(person['score'])])
        res['oks%d.kept' % i] = np.array(kept, dtype=np.float64)
        meta['oks%d' % i] = {'N': N, 'per_img': per_img, 'soft': soft, 'in_vis': in_vis, 'oks_thre': oks_thre}
    np.savez_compressed(os.path.join(OUT, 'validate.npz'), **res)
    with open(os.path.join(OUT, 'validate.json'), 'w') as f:
        json.dump(meta, f)


INPUT_CASES = [('small', 4, 5, 64, 48, 16, 12), ('coco', 3, 17, 256, 192, 64, 48), ('w48', 2, 17, 384, 288, 96, 72)]


def gen_inputpipe(M):
    """Input pipeline from the REAL reference: ``grid_aug`` as MixCombine calls it and
    ``JointsDataset.generate_target`` (constructor bypassed).  ToTensor + Normalize is torchvision (absent):
    the images fed to grid_aug come from oracle.inputpipe.to_tensor_normalize (see its header)."""
    from oracle import inputpipe as ip
    for m, attrs in (('pycocotools', {}), ('pycocotools.coco', {'COCO': None}),
                     ('pycocotools.cocoeval', {'COCOeval': None}), ('json_tricks', {}),
                     ('imagecorruptions', {'corrupt': None, 'get_corruption_names': None})):
        mod = types.ModuleType(m)
        for k, v in attrs.items():
            setattr(mod, k, v)
        sys.modules.setdefault(m, mod)
    pkg = types.ModuleType('dataset')
    pkg.__path__ = [os.path.join(REF, 'lib', 'dataset')]
    sys.modules.setdefault('dataset', pkg)
    import dataset.advaug as adv
    import dataset.JointsDataset as jd
    res, meta = {}, {}
    for tag, B, J, H, W, Hh, Wh in INPUT_CASES:
        base, aug, jt, vis = ip.synth_samples('inp.' + tag, B, J, H, W)
        draws = []
        for b in range(B):
            img = ip.to_tensor_normalize(base[b])
            np.random.seed(1000 + 17 * b + H)
            state = np.random.get_state()
            out, _, vis_out, rec = adv.grid_aug(AD(joints_num=J), img.clone(), jt[b].copy(), vis[b].copy(), True, True,
                                                1, False, 0.5, 1, 0.7, {})
            np.random.set_state(state)                      # replay the draws through the restatement
            draws.append(ip.grid_draws(H, W, 0.5, 0.7, 1, np.random))
            kept = (out != 0)[0].numpy() if draws[-1] is not None else np.ones((H, W), bool)
            res['%s.mask%d' % (tag, b)] = np.packbits(kept)
            res['%s.vis%d' % (tag, b)] = np.asarray(vis_out)
            fake = types.SimpleNamespace(num_joints=J, heatmap_size=np.array([Wh, Hh]), image_size=np.array([W, H]),
                                         sigma=2, target_type='gaussian', use_different_joints_weight=False,
                                         joints_weight=1)
            for name, vv in (('clean', vis[b]), ('grid', np.asarray(vis_out))):
                tgt, tw = jd.JointsDataset.generate_target(fake, jt[b].copy(), vv.copy())
                res['%s.%s.target%d' % (tag, name, b)] = tgt[0]
                res['%s.%s.tw%d' % (tag, name, b)] = tw
        meta[tag] = {'draws': draws, 'masked': int(sum(d is not None for d in draws))}
        print('inputpipe', tag, meta[tag], flush=True)
    # different joints weights (JointsDataset.py:488-489)
    tag, B, J, H, W, Hh, Wh = INPUT_CASES[1]
    base, aug, jt, vis = ip.synth_samples('inp.' + tag, B, J, H, W)
    jw = np.array([1., 1., 1., 1., 1., 1., 1., 1.2, 1.2, 1.5, 1.5, 1., 1., 1.2, 1.2, 1.5, 1.5],
                  dtype=np.float32).reshape((J, 1))          # coco.py:74-80 (data)
    fake = types.SimpleNamespace(num_joints=J, heatmap_size=np.array([Wh, Hh]), image_size=np.array([W, H]),
                                 sigma=2, target_type='gaussian', use_different_joints_weight=True, joints_weight=jw)
    _, tw = jd.JointsDataset.generate_target(fake, jt[0].copy(), vis[0].copy())
    res['coco.jw.tw0'] = tw
    np.savez_compressed(os.path.join(OUT, 'inputpipe.npz'), **res)
    with open(os.path.join(OUT, 'inputpipe.json'), 'w') as f:
        json.dump(meta, f)


AUTOAUG_CASES = (('small', 48, 64, 48), ('coco', 6, 256, 192), ('odd', 6, 33, 21))


def gen_autoaug(M):
    """The AutoAugment view (SURVEY 8 f2 remainder) from the REAL ``ImageNetPolicy`` (advaug.py:10-108, constructed as
    MixCombine does, :175) on synthetic crops: per sample Python's ``random`` is seeded, the real policy is applied to
    the PIL image (as JointsDataset / MixCombine do, advaug.py:184-186), and the draws are replayed through
    oracle.autoaug.draw_policy from the same state.  Stores the outputs (uint8) and the replayed draws.
    ``np.int`` (removed in numpy 1.24) is shimmed for advaug.py:56."""
    import random
    import zlib
    from PIL import Image
    from oracle import autoaug as oa
    from oracle import inputpipe as ip
    if not hasattr(np, 'int'):
        np.int = int
    for m, attrs in (('pycocotools', {}), ('pycocotools.coco', {'COCO': None}),
                     ('pycocotools.cocoeval', {'COCOeval': None}), ('json_tricks', {}),
                     ('imagecorruptions', {'corrupt': None, 'get_corruption_names': None})):
        mod = types.ModuleType(m)
        for k, v in attrs.items():
            setattr(mod, k, v)
        sys.modules.setdefault(m, mod)
    pkg = types.ModuleType('dataset')
    pkg.__path__ = [os.path.join(REF, 'lib', 'dataset')]
    sys.modules.setdefault('dataset', pkg)
    import dataset.advaug as adv
    policy = adv.MixCombine().autoaug
    res, meta = {}, {}
    seen = set()
    for tag, B, H, W in AUTOAUG_CASES:
        base, _, _, _ = ip.synth_samples('aa.' + tag, B, 1, H, W)
        if tag == 'small':
            base[0] = 77                                    # one colour: equalize leaves it alone (len(histo) <= 1)
            base[1] = (base[1] // 128) * 200                # two levels
            base[2, :, :, 1] = base[2, :, :, 0] // 64 * 60  # few levels in one band: step == 0 for small images
        draws, crcs = [], []
        for b in range(B):
            seed = 4242 + 31 * b + H
            random.seed(seed)
            st = random.getstate()
            out = np.array(policy(Image.fromarray(base[b].astype(np.uint8))))
            random.setstate(st)
            ops = oa.draw_policy(random)
            draws.append([[int(c), float(p)] for c, p in ops])
            seen.update(c for c, _ in ops)
            crcs.append(zlib.crc32(np.ascontiguousarray(out).tobytes()))
            if tag == 'odd':                                # full outputs for the small case, CRC-32 of the bytes for all
                res['%s.out%d' % (tag, b)] = out
        meta[tag] = {'draws': draws, 'crc32': crcs}
        print('autoaug', tag, draws, flush=True)
    assert seen == {1, 2, 3, 4, 5}, seen                    # every reachable operation is covered by some sample
    np.savez_compressed(os.path.join(OUT, 'autoaug.npz'), **res)
    with open(os.path.join(OUT, 'autoaug.json'), 'w') as f:
        json.dump(meta, f)


def main():
    logging.basicConfig(level=logging.WARNING)
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    M = import_reference()
    which = sys.argv[1:] or ['keys', 'loss', 'nms', 'forward', 'advmix', 'validate', 'inputpipe', 'c4', 'b32', 'benchtiles', 'c1', 'autoaug', 'c5trunk', 'nmsvis', 'accuracy', 'c4b32']      # ('c4b32step': on request - it needs the whole container's memory)
    for w in which:
        globals()['gen_' + w](M)
        print('done', w, flush=True)


if __name__ == '__main__':
    main()
