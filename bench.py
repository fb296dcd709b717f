#!/usr/bin/env python3
"""Headline benchmark: images/sec of one full AdvMix train step (HRNet-W32 256x192, B=32 per
GPU, synthetic inputs resident in HBM, random-init weights), fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hrnet_w32|resnet50|hrnet_w48]

N > 1: either the driver's `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, or plain
`python bench.py --gpus N`, which starts its own N ranks (advmix_amd/launch.py: one fresh process per GPU before
anything touches the GPU; exits non-zero if fewer than N GPUs are visible - it never silently runs one rank).
One rank per GPU, RCCL; per-GPU work is fixed (weak scaling); `rccl_ranks` in the line is dist.get_world_size().  A "step" is the
body of train_advmix's batch loop (lib/core/function.py:137-171): G fwd, softmax-mix, D step
(heat-map + KD loss, backward, Adam), G step through the updated frozen D (backward, Adam),
loss.item() and the PCK accuracy read-out - nothing is skipped inside the timed region.

The JSON line also carries
  roofline     - the dominant kernel (fp32-MFMA implicit-GEMM conv) timed live with HIP events:
                 algorithmic FLOPs per launch / mean launch time vs the 157.3 TFLOP/s fp32 matrix peak
  cpu_baseline - the CPU oracle (a restatement of the reference step, pinned to it by golden
                 vectors) timed on this box's host cores on a bounded sample (B=4, a few steps).
"""
import argparse
import json
import os
import sys
import time
import types

if int(os.environ.get('WORLD_SIZE', '1')) > 1:
    # (read by the HIP runtime when it comes up, i.e. before ``import torch``: see advmix_amd/launch.py - the driver's
    #  torch.distributed.run form reaches this file without passing through the launcher)
    os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"

HRNET_STAGES = {
    'hrnet_w32': (32, 64, 128, 256),
    'hrnet_w48': (48, 96, 192, 384),
}


def hrnet_extra(widths):
    ex = {'FINAL_CONV_KERNEL': 1, 'PRETRAINED_LAYERS': ['*']}
    for i, (st, nmod) in enumerate(((2, 1), (3, 4), (4, 3))):
        ex['STAGE%d' % st] = {'NUM_MODULES': nmod, 'NUM_BRANCHES': st, 'BLOCK': 'BASIC',
                              'NUM_BLOCKS': [4] * st, 'NUM_CHANNELS': list(widths[:st]),
                              'FUSE_METHOD': 'SUM'}
    return ex


WORKLOADS = {
    # name: (MODEL.NAME, EXTRA, joints, H, W, unet downs, step GFLOP/img from SURVEY.md 8(d4))
    'hrnet_w32': ('pose_hrnet', hrnet_extra(HRNET_STAGES['hrnet_w32']), 17, 256, 192, 6, 118.58),
    'hrnet_w48': ('pose_hrnet', hrnet_extra(HRNET_STAGES['hrnet_w48']), 17, 384, 288, 5, 480.6),
    'resnet50': ('pose_resnet', {'FINAL_CONV_KERNEL': 1, 'DECONV_WITH_BIAS': False, 'NUM_DECONV_LAYERS': 3,
                                 'NUM_DECONV_FILTERS': [256, 256, 256], 'NUM_DECONV_KERNELS': [4, 4, 4],
                                 'NUM_LAYERS': 50}, 17, 256, 192, 6, 91.77),
    # BASELINE.json configs[4] (C5) as far as it can be built: the reference has NO HigherHRNet model, loss or grouping
    # code (README.md:72-73 lists its accuracy; tools/test_corruption.py:147 is a dead branch), so there is no oracle.
    # This is the HRNet-W32 trunk + UnetGenerator(9,3,6) AdvMix step at 512x512 - the 128x128x32 ... 16x16x256 shapes of
    # that resolution - as a THROUGHPUT-ONLY line, never the headline.  GFLOP / image: the 256x192 counts x (512*512)/
    # (256*192): 6 * 81.55 + 3 * 48.19 - 1.43.
    'hrnet_w32_512': ('pose_hrnet', hrnet_extra(HRNET_STAGES['hrnet_w32']), 17, 512, 512, 6, 632.4),
}
NO_ORACLE = {'hrnet_w32_512': 'no oracle for HigherHRNet - the reference has no such code (README.md:72-73): trunk + generator '
                              'step only, no associative-embedding head / grouping; the trunk and the generator THEMSELVES are '
                              'parity-tested at 512x512 (tests: hrnet_w32_512, vectors from the real pose_hrnet / UnetGenerator)'}


def synth(B, J, H, W, device, seed):
    """3 N(0,1) views, Gaussian sigma=2 targets, weights in {0,1} (P=0.8)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    views = [torch.randn(B, 3, H, W, generator=g).to(device) for _ in range(3)]
    hh, ww = H // 4, W // 4
    cx = torch.randint(0, ww, (B, J, 1, 1), generator=g).float()
    cy = torch.randint(0, hh, (B, J, 1, 1), generator=g).float()
    ys = torch.arange(hh).float().view(1, 1, hh, 1)
    xs = torch.arange(ww).float().view(1, 1, 1, ww)
    tgt = torch.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / 8.0)
    tgt[tgt < 0.0111] = 0
    tw = (torch.rand(B, J, 1, generator=g) < 0.8).float()
    return views, tgt.to(device).contiguous(), tw.to(device)


def build_models(workload, device):
    from advmix_amd import models
    from advmix_amd.config import CfgNode
    from advmix_amd.core.loss import JointsMSELoss
    from advmix_amd.utils.utils import get_optimizer
    net, extra, J, H, W, downs, _ = WORKLOADS[workload]
    cfg = CfgNode({'MODEL': {'NAME': net, 'EXTRA': extra, 'NUM_JOINTS': J, 'INIT_WEIGHTS': True, 'PRETRAINED': ''},
                   'TRAIN': {'OPTIMIZER': 'adam', 'LR': 1e-3}, 'LOSS': {'USE_TARGET_WEIGHT': True}})
    torch.manual_seed(1234)
    mod = getattr(models, net)
    D = mod.get_pose_net(cfg, is_train=True)                       # tools/train.py:60
    T = mod.get_pose_net(cfg, is_train=False)
    T.load_state_dict(D.state_dict())                              # copy.deepcopy(model), train.py:65
    G = models.Unet_generator.UnetGenerator(9, 3, downs)           # train.py:67
    D, T, G = D.to(device), T.to(device), G.to(device)
    crit = JointsMSELoss(use_target_weight=True)
    optD, optG = get_optimizer(cfg, D), get_optimizer(cfg, G)
    D.train(); G.train(); T.eval()
    return cfg, D, G, T, crit, optD, optG


def _event_time(run, iters, reps=5):
    """Median over ``reps`` HIP-event measurements of ``iters`` back-to-back launches (ms per launch);
    events are recorded on the stream the kernels are launched on (torch's current stream)."""
    for _ in range(20):
        run()
    vals = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        vals.append(e0.elapsed_time(e1) / iters)
    vals.sort()
    return vals[len(vals) // 2], vals


CONV_FAMILY = ((32, 64, 48), (64, 32, 24), (128, 16, 12), (256, 8, 6))     # HRNet-W32's 3x3 s1 C->C convs: 1.81 GF each at B = 32
# launches per AdvMix step of each kind of a given conv: two train-mode student forwards + the eval-mode teacher, two
# input gradients (D step, G step) - of a BasicBlock's two convs one takes the residual path's gradient as addend and the
# sign of y from the bit mask, the other recomputes it from c - one weight gradient (D step)
KIND_WEIGHT = {'fwd+BN-sums': 2, 'fwd+BN-eval+ReLU': 1, 'dgrad+addend+BN-bwd-sums (act mask)': 1,
               'dgrad+BN-bwd-sums (sign from c)': 1, 'wgrad': 1}


def _pmc_file(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.relpath(files[-1], ROOT)


def _latest_pmc():
    """HBM bytes per launch of the dominant kernel from this round's rocprofv3 PMC passes (tools/pmc_conv.sh +
    tools/summarize_pmc.py: FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate --pmc runs); newest profiles/r*_pmc file."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_conv32_epi.json')))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return round(json.load(f)['hbm_bytes_per_launch']), os.path.relpath(files[-1], ROOT)


def time_conv_family(B, device, iters=100, family=None):
    """The roofline object.  The step's time is the MFMA convs' (SURVEY 8 d3), and no single launch dominates: the four
    branch resolutions of HRNet-W32 each run the same 1.81 GFLOP 3x3 conv, as forward (+ BatchNorm column sums, or +
    eval BatchNorm + ReLU for the teacher), input gradient (+ the BatchNorm-backward sums of its producer) and weight
    gradient.  Every member is timed live, back to back through the C ABI with HIP events on the launching stream;
    ``frac`` is the launch-count-weighted FLOP/s of the whole family against the fp32 matrix peak, ``members`` lets
    each number be recomputed, ``dominant`` is the most frequent single kernel (3x3 32->32 @64x48 + sums, 128 launches
    per step) with its measured HBM traffic.  ``hbm_kernels``: the two BatchNorm kernels left on the path against the
    8 TB/s HBM peak."""
    import ctypes
    from advmix_amd._lib import call
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())       # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    members, hbm = [], []
    tot_f = tot_t = 0.0
    dominant = None
    family = family or CONV_FAMILY
    for C, H, W in family:
        rows = B * H * W
        x = torch.randn(B, H, W, C, device=device)
        w = torch.randn(C, 3, 3, C, device=device) * (9 * C) ** -0.5
        y, c2, dx = torch.empty_like(x), torch.randn_like(x), torch.empty_like(x)
        dy = torch.randn_like(x)
        dw = torch.zeros_like(w)
        g, b, rm = (torch.randn(C, device=device) for _ in range(3))
        rv = torch.rand(C, device=device) + 0.5
        mean, invstd = torch.zeros(C, device=device), torch.ones(C, device=device)
        slots = torch.zeros(2 * C * 64, device=device, dtype=torch.float64)
        amask = torch.randint(0, 16, (rows * C // 4,), device=device, dtype=torch.uint8)     # a bit per element of y
        nbg = ctypes.c_int(0)
        geom = (B, H, W, C, H, W, C, 3, 3, 1, 1)
        flops = 2.0 * rows * C * C * 9

        def reset():
            nbg.value = 0
        runs = {
            'fwd+BN-sums': lambda: (reset(), call('advmix_conv_fwd_ex', P(x), P(w), None, P(y), *geom, None, None, None,
                                                  None, 0.0, None, 0, P(slots), ctypes.byref(nbg), st)),
            'fwd+BN-eval+ReLU': lambda: call('advmix_conv_fwd_ex', P(x), P(w), None, P(y), *geom, P(g), P(b), P(rm), P(rv),
                                             1e-5, None, 1, None, None, st),
            'dgrad+addend+BN-bwd-sums (act mask)': lambda: (reset(), call(
                'advmix_conv_tr_w_bnb', P(dy), P(w), P(c2), P(dx), *geom, P(amask), P(c2), P(mean), P(invstd), None, None, 1,
                P(slots), ctypes.byref(nbg), st)),
            'dgrad+BN-bwd-sums (sign from c)': lambda: (reset(), call(
                'advmix_conv_tr_w_bnb', P(dy), P(w), None, P(dx), *geom, None, P(c2), P(mean), P(invstd), P(g), P(b), 1,
                P(slots), ctypes.byref(nbg), st)),
        }
        # the weight gradients of a branch's eight 3x3 convs go out as ONE launch (ops.Chain.bwd, advmix_conv_wgrad_group): timed
        # as that launch, reported per problem
        NG = 8
        gdy = [dy] + [torch.randn_like(x) for _ in range(NG - 1)]
        gx = [x] + [torch.randn_like(x) for _ in range(NG - 1)]
        gdw = [dw] + [torch.zeros_like(w) for _ in range(NG - 1)]
        arr = ctypes.c_void_p * NG
        ga, gb, gd = arr(*[t.data_ptr() for t in gdy]), arr(*[t.data_ptr() for t in gx]), arr(*[t.data_ptr() for t in gdw])
        from advmix_amd._lib import lib as _lib
        grouped = _lib.advmix_conv_wgrad_group(NG, ga, gb, gd, B, H, W, C, H, W, C, 3, 3, 1, 1, st) == 0   # (not every width is served)
        if grouped:
            runs['wgrad'] = lambda: call('advmix_conv_wgrad_group', NG, ga, gb, gd, B, H, W, C, H, W, C, 3, 3, 1, 1, st)
        else:
            runs['wgrad'] = lambda: call('advmix_conv_wgrad', P(dy), P(x), P(dw), B, H, W, C, H, W, C, 3, 3, 1, 1, st)
        for kind, run in runs.items():
            ms, rr = _event_time(run, iters if not (kind == 'wgrad' and grouped) else max(iters // 4, 10))
            if kind == 'wgrad' and grouped:
                ms, rr = ms / NG, [v / NG for v in rr]      # per problem of the eight-problem launch
            wgt = KIND_WEIGHT[kind]
            tot_f += wgt * flops
            tot_t += wgt * ms * 1e-3
            m = {'kernel': '3x3 s1 %d->%d @%dx%d %s' % (C, C, H, W, kind if not (kind == 'wgrad' and grouped) else 'wgrad (1 of 8 problems of one launch)'),
                 'us_per_launch': round(ms * 1e3, 2),
                 'tflops': round(flops / (ms * 1e-3) / 1e12, 2), 'frac': round(flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                 'launches_per_step_weight': wgt}
            members.append(m)
            if C == family[0][0] and kind == 'fwd+BN-sums':
                dominant = dict(m, us_per_launch_runs=[round(v * 1e3, 2) for v in rr],
                                algorithmic_gflop_per_launch=round(flops / 1e9, 3))
        if C in (family[0][0], family[2][0]):               # the two BatchNorm kernels left on the train path
            res = torch.randn_like(x)
            nbt = torch.zeros((), dtype=torch.int64, device=device)
            call('advmix_conv_fwd_ex', P(x), P(w), None, P(c2), *geom, None, None, None, None, 0.0, None, 0, P(slots),
                 ctypes.byref(nbg), st)
            ns = nbg.value
            for name, run, passes in (
                    ('norm_apply_slots (BN + residual + ReLU, statistics from %d slots)' % ns,
                     lambda: call('advmix_norm_apply_slots', P(c2), P(slots), ns, rows, C, 1e-5, P(g), P(b), P(res), P(y), 1,
                                  P(mean), P(invstd), P(rm), P(rv), P(nbt), 0.1, P(amask), st), 3),
                    ('norm_bwd_apply_slots (BN backward from the slot sums)',
                     lambda: call('advmix_norm_bwd_apply_slots', P(dy), P(c2), P(mean), P(invstd), P(g), P(slots), ns, rows,
                                  C, P(dx), None, None, st), 3)):
                ms, _ = _event_time(run, iters)
                nbytes = passes * rows * C * 4
                hbm.append({'kernel': '%s rows %d x C %d' % (name, rows, C), 'us_per_launch': round(ms * 1e3, 2),
                            'algorithmic_bytes_per_launch': nbytes, 'achieved_GBps': round(nbytes / (ms * 1e-3) / 1e9, 1),
                            'frac_of_8TBps': round(nbytes / (ms * 1e-3) / 1e9 / 8000.0, 4)})
    traffic, src = _latest_pmc() if (B == 32 and family == CONV_FAMILY) else (None, None)
    bnb_traffic = None
    if B == 32 and family == CONV_FAMILY:                   # the input gradient + BatchNorm-backward member (5 tensors of 12.6 MB)
        d, f = _pmc_file('r*_pmc_conv32_dgrad_bnb.json')
        if d:
            bnb_traffic = {'hbm_bytes_per_launch': round(d['hbm_bytes_per_launch']), 'algorithmic_bytes': d['algorithmic_bytes_per_launch'],
                           'ratio': round(d['hbm_bytes_per_launch'] / d['algorithmic_bytes_per_launch'], 3), 'source': f}
    step_util = None
    if B == 32 and family == CONV_FAMILY:                   # SQ_VALU_MFMA_BUSY_CYCLES summed over one step (tools/pmc_step.sh)
        d, f = _pmc_file('r*_pmc_step_mfma.json')
        if d:
            step_util = {'mfma_busy_simd_cycles_per_step': round(d['mfma_busy_cycles_per_step']),
                         'algorithmic_simd_cycles_per_step': round(118.58e9 * 32 / 64),
                         'utilisation_at_the_profiled_step_time': round(d['mfma_utilisation_of_step'], 4), 'source': f}
    agg = tot_f / tot_t / 1e12
    C0, H0, W0 = family[0]
    algo_bytes = 2 * B * H0 * W0 * C0 * 4 + 9 * C0 * C0 * 4
    return {'bound': 'mfma',
            'kernel': 'conv_direct / conv_wgrad family: 3x3 s1 C->C at the four HRNet branch resolutions of this workload x '
                      '{fwd+BN sums, fwd+BN eval, dgrad+BN-bwd sums, wgrad}, launch-count weighted',
            'achieved': round(agg, 3), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(agg / FP32_MFMA_PEAK_TFLOPS, 4),
            'traffic': traffic, 'traffic_unit': 'HBM bytes per launch of the dominant member (rocprofv3 PMC, corrected)',
            'traffic_source': src, 'traffic_algorithmic_bytes': algo_bytes,
            'traffic_ratio': round(traffic / algo_bytes, 3) if traffic else None,
            'traffic_dgrad_bnb': bnb_traffic,
            'step_mfma_utilisation_pmc': step_util,
            'dominant': dominant, 'members': members, 'hbm_kernels': hbm}


def time_eval_conv(B, device, iters=100):
    """Dominant kernel of the validation path: the same 3x3 s1 32->32 conv with the eval-mode BatchNorm +
    ReLU folded into its epilogue (one launch per conv+bn+relu)."""
    import ctypes
    from advmix_amd._lib import call
    x = torch.randn(B, 64, 48, 32, device=device)
    w = torch.randn(32, 3, 3, 32, device=device) * 0.05
    y = torch.empty(B, 64, 48, 32, device=device)
    g, b, rm = (torch.randn(32, device=device) for _ in range(3))
    rv = torch.rand(32, device=device) + 0.5
    P = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: call('advmix_conv_fwd_ex', P(x), P(w), None, P(y), B, 64, 48, 32, 64, 48, 32, 3, 3, 1, 1,  # noqa: E731
                       P(g), P(b), P(rm), P(rv), 1e-5, None, 1, None, None, st)
    ms, runs = _event_time(run, iters)
    flops = 2.0 * B * 64 * 48 * 32 * 32 * 9
    return {'bound': 'mfma', 'kernel': 'conv_direct<1,1,4,1,32,fwd,epilogue=BN-eval+ReLU> 3x3 s1 32->32 @64x48',
            'achieved': round(flops / (ms * 1e-3) / 1e12, 3), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': None,
            'us_per_launch': round(ms * 1e3, 2), 'us_per_launch_runs': [round(v * 1e3, 2) for v in runs],
            'algorithmic_gflop_per_launch': round(flops / 1e9, 3)}


def bench_validate(a, device, rank, world):
    """--path validate: images/sec of the validate() batch body (function.py:223-300) with the COCO test
    settings of the experiment YAMLs (FLIP_TEST, SHIFT_HEATMAP, POST_PROCESS): two eval forwards, the
    fused flip-back/shift/average kernel, loss.item(), PCK accuracy, device get_final_preds + its D2H."""
    import numpy as np
    from advmix_amd.core.function import validate_batch
    from advmix_amd.core.evaluate import accuracy
    from advmix_amd.core.inference import get_final_preds
    from advmix_amd.dataset.coco import COCO_FLIP_PAIRS
    from advmix_amd.config import CfgNode
    net, extra, J, H, W, downs, _ = WORKLOADS[a.workload]
    cfg, D, G, T, crit, optD, optG = build_models(a.workload, device)
    cfg['TEST'] = CfgNode({'FLIP_TEST': True, 'SHIFT_HEATMAP': True, 'POST_PROCESS': True})
    D.eval()
    views, tgt, tw = synth(a.batch, J, H, W, device, 1234 + rank)
    rng = np.random.default_rng(7 + rank)
    center = (rng.random((a.batch, 2)) * [600, 440] + 20).astype(np.float32)
    sw = (rng.random(a.batch) * 2.5 + 0.4).astype(np.float32)
    scale = np.stack([sw, sw / np.float32(0.75)], 1)
    graph = None
    if a.exec_mode == 'graph':
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                validate_batch(cfg, D, crit, views[0], tgt, tw, COCO_FLIP_PAIRS)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        from advmix_amd import ops as _ops
        graph = _ops.GraphSeq(device)
        gseg, (g_out, g_loss) = graph.capture(lambda: validate_batch(cfg, D, crit, views[0], tgt, tw, COCO_FLIP_PAIRS))

    def one_batch():
        if graph is not None:
            graph.replay(gseg)
            out, loss = g_out, g_loss
        else:
            out, loss = validate_batch(cfg, D, crit, views[0], tgt, tw, COCO_FLIP_PAIRS)
        lv = loss.item()
        accuracy(out, tgt)
        preds, maxvals = get_final_preds(cfg, None, out, center, scale)
        return lv, preds

    import torch.distributed as dist
    for _ in range(a.warmup):
        one_batch()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        lv, preds = one_batch()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank != 0:
        return None
    fwd_gflop = {'hrnet_w32': 15.290, 'hrnet_w48': 70.613, 'resnet50': 10.853, 'hrnet_w32_512': 81.55}[a.workload]
    value = a.batch * world * a.steps / dt
    line = {'metric': 'images/sec validate batch, flip test (%s)' % a.workload, 'value': round(value, 2),
            'unit': 'images/sec', 'n_gpus': world, 'rccl_ranks': _ranks(), 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic (N(0,1) images, Gaussian targets, random boxes), random-init weights',
            'config': {'workload': '%s_%dx%d_validate_flip' % (a.workload, H, W), 'batch_per_gpu': a.batch,
                       'global_batch': a.batch * world, 'parallelism': 'dp%d' % world,
                       'exec': 'hipgraph' if graph is not None else 'eager',
                       'batch_gflop_per_image': 2 * fwd_gflop},
            'step_tflops_per_gpu': round(value / world * 2 * fwd_gflop / 1e3, 2),
            'step_frac_of_fp32_mfma_peak': round(value / world * 2 * fwd_gflop / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4),
            'last_loss': round(lv, 6)}
    if not a.no_roofline:
        line['roofline'] = time_eval_conv(a.batch, device)
    if world == 1 and not a.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(a.workload, path='validate')
    return line


def bench_inputs(a, device, rank, world):
    """--path inputs: images/sec of the device input pipeline: from ONE uint8 crop per sample and the
    workers' draws to the AutoAugment view (device, round 3), the three normalised float views (GridMask on the third)
    and the gaussian targets / target weights.  HBM-bound: 6 B read + 36 B written per pixel by the view kernel."""
    import numpy as np
    import random as pyrandom
    from advmix_amd.dataset.advaug import make_views, pack_grid, grid_params, auto_augment, pack_autoaug, autoaug_params
    from advmix_amd.dataset.JointsDataset import TargetRenderer
    net, extra, J, H, W, downs, _ = WORKLOADS[a.workload]
    rng = np.random.RandomState(99 + rank)
    base = torch.from_numpy(rng.randint(0, 256, (a.batch, H, W, 3), dtype=np.uint8)).to(device)
    grid = pack_grid([grid_params(H, W, rng=rng) for _ in range(a.batch)], device)
    prng = pyrandom.Random(7 + rank)
    aa = pack_autoaug([autoaug_params(prng) for _ in range(a.batch)], device)      # the workers' draws (advaug.py:38-40,102-105)
    joints = np.zeros((a.batch, J, 3)); joints[:, :, 0] = rng.rand(a.batch, J) * W; joints[:, :, 1] = rng.rand(a.batch, J) * H
    vis = np.zeros((a.batch, J, 3)); vis[:, :, :2] = (rng.rand(a.batch, J, 1) < 0.8)
    jd, vd = torch.from_numpy(joints).to(device), torch.from_numpy(vis).to(device)
    rend = TargetRenderer((W, H), (W // 4, H // 4), 2, device=device)

    def one_batch():
        aug = auto_augment(base, aa)                        # the AutoAugment view on the device (round 3)
        views = make_views(base, aug, grid)
        tgt, tw = rend.render(jd, vd)
        return views, tgt, tw
    aug = auto_augment(base, aa)

    for _ in range(a.warmup):
        one_batch()
    torch.cuda.synchronize()
    steps = max(a.steps, 200)
    t0 = time.perf_counter()
    for _ in range(steps):
        one_batch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank != 0:
        return None
    import ctypes
    from advmix_amd._lib import call
    v = [torch.empty((a.batch, 3, H, W), device=device) for _ in range(3)]
    P = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731
    m = (ctypes.c_float * 3)(0.485, 0.456, 0.406); sd = (ctypes.c_float * 3)(0.229, 0.224, 0.225)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms, runs = _event_time(lambda: call('advmix_make_views', P(base), P(aug), P(grid), ctypes.cast(m, ctypes.c_void_p),
                                        ctypes.cast(sd, ctypes.c_void_p), P(v[0]), P(v[1]), P(v[2]), a.batch, H, W, st), 100)
    nbytes = a.batch * H * W * (6 + 36)
    value = a.batch * world * steps / dt
    ms_aa, _ = _event_time(lambda: auto_augment(base, aa), 100)
    line = {'metric': 'images/sec device input pipeline: AutoAugment + 3 views + targets (%dx%d)' % (H, W), 'value': round(value, 1),
            'unit': 'images/sec', 'n_gpus': world, 'steps': steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'u8->f32', 'data': 'synthetic uint8 crops resident in HBM, random joints',
            'config': {'workload': 'inputs_%dx%d_3views_targets' % (H, W), 'batch_per_gpu': a.batch, 'joints': J},
            'roofline': {'bound': 'hbm', 'kernel': 'make_views_kernel', 'achieved': round(nbytes / (ms * 1e-3) / 1e9, 1),
                         'peak': 8000.0, 'unit': 'GB/s', 'frac': round(nbytes / (ms * 1e-3) / 1e9 / 8000.0, 4),
                         'traffic': None, 'us_per_launch': round(ms * 1e3, 2),
                         'algorithmic_bytes_per_launch': nbytes},
            'autoaug_us_per_batch': round(ms_aa * 1e3, 2)}
    if world == 1 and not a.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(a.workload, path='inputs')
    return line


def bench_nms(a, device, rank, world):
    """--path nms: the lib/nms row.  A step = the post-process of one image: box NMS over
    N = 1000 scored boxes through the reproduced ``_nms`` ABI (H2D, 64-wide bitmask kernel, D2H, host greedy
    pass - per-call malloc/free like the reference) plus OKS-NMS over 30 person candidates (fp64 OKS matrix on
    the device, greedy pass on the host).  Latency-bound by design: the reference's interface is host to host."""
    import numpy as np
    from advmix_amd.nms.nms import gpu_nms, oks_nms
    rng = np.random.RandomState(11 + rank)
    N = 1000
    xy = rng.rand(N, 2) * 400
    wh = rng.rand(N, 2) * 120 + 10
    dets = np.concatenate([xy, xy + wh, rng.rand(N, 1)], 1).astype(np.float32)
    people = []
    base = rng.rand(6, 17, 2) * 300 + 50
    for n in range(30):
        k = np.zeros((17, 3)); k[:, :2] = base[n % 6] + rng.randn(17, 2) * 4; k[:, 2] = rng.rand(17)
        people.append({'keypoints': k.reshape(-1), 'area': float(rng.rand() * 20000 + 5000), 'score': float(rng.rand())})

    def one():
        return len(gpu_nms(dets, 0.5)), len(oks_nms(people, 0.9))
    for _ in range(a.warmup):
        one()
    torch.cuda.synchronize()
    steps = max(a.steps, 100)
    t0 = time.perf_counter()
    for _ in range(steps):
        kept = one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(steps):
        gpu_nms(dets, 0.5)
    torch.cuda.synchronize()
    dt_box = time.perf_counter() - t1
    if rank != 0:
        return None
    line = {'metric': 'images/sec NMS post-process (box NMS N=1000 + OKS-NMS 30 persons)', 'value': round(world * steps / dt, 1),
            'unit': 'images/sec', 'n_gpus': world, 'steps': steps, 'warmup': a.warmup, 'ms_per_step': round(dt / steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 IoU / f64 OKS -> int indices',
            'data': 'synthetic boxes / keypoints (host arrays, as the reference interface takes them)',
            'config': {'workload': 'nms_box1000_oks30', 'kept': list(kept)},
            'box_nms_us_per_call': round(dt_box / steps * 1e6, 1),
            'roofline': {'bound': 'latency', 'kernel': 'nms_mask (16 x 16 tiles of 64 x 64 IoUs, one ballot per row)',
                         'achieved': None, 'peak': None, 'unit': None, 'frac': None, 'traffic': None,
                         'note': '1 M IoUs = a few microseconds of device work; the call is bound by hipMalloc/free + '
                                 'two PCIe copies + the host greedy pass, all of which the reference ABI prescribes'}}
    if world == 1 and not a.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(a.workload, path='nms')
    return line


def _ranks():
    """Ranks in the RCCL process group as torch.distributed sees them (1 when no group was needed)."""
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def cpu_baseline(workload, budget_s=25.0, hard_timeout_s=240.0, path="train"):
    """The CPU oracle's AdvMix step on this host (bounded sample: B=4, 1 warm-up + a few timed
    steps), in a CPU-only child process with a hard timeout so the bench always finishes."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, 'oracle', 'cpu_bench.py'), workload, str(budget_s), path]
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=hard_timeout_s, env=env, cwd=ROOT)
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith('{'):
                return json.loads(ln)
        return {'value': None, 'unit': 'images/sec', 'cores': 0, 'kind': 'port',
                'sample': 'cpu oracle failed: ' + (out.stderr.strip().splitlines() or ['?'])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {'value': None, 'unit': 'images/sec', 'cores': 0, 'kind': 'port',
                'sample': 'cpu oracle exceeded the %.0fs hard timeout' % hard_timeout_s}


def rendezvous(a, backend, rank, world, local):
    """--path rendezvous: the launcher's self-test.  Every rank joins the process group, one all-reduce checks that
    all ``world`` ranks are really there, rank 0 prints a JSON line.  With the default backend (nccl = RCCL) each rank
    binds its own GPU; ADVMIX_BENCH_BACKEND=gloo runs the same path on CPU (tests/test_host_cpu.py)."""
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29555')
    if backend == 'nccl':
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    else:
        device = torch.device('cpu')
        dist.init_process_group(backend, rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)], device=device)
    dist.all_reduce(t)
    ok = float(t.item()) == world * (world + 1) / 2
    ranks = dist.get_world_size()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'metric': 'rendezvous', 'n_gpus': a.gpus, 'rccl_ranks': ranks, 'backend': backend,
                          'allreduce_ok': ok}), flush=True)
    if not ok:
        raise SystemExit(4)


def dp_verdict(line, sync, optimizers, verification, lv):
    """The three fields every N-rank line carries, from the SAME calls the train path makes, and the job's exit verdict."""
    replicas = sync.replicas_state(optimizers)
    line['replicas_identical'] = replicas['identical']
    line['all_finite'] = replicas['finite'] and (lv == lv)
    line['grad_exchange_verified'] = verification[0] if verification is not None else None
    return not line['replicas_identical'] or not line['all_finite'] or (verification is not None and not verification[0])


def replicas_selftest(a, backend, rank, world, local):
    """--path replicas: the N-rank verdict's self-test (ADVMIX_BENCH_BACKEND=gloo: on CPU, tests/test_host_cpu.py).  Every rank
    trains a small network for a few synced steps through dp.GradSync with its exchanges traced; ADVMIX_BENCH_CORRUPT=
    weight | nan | exchange makes rank 1 move one weight by a few ulps / put a NaN into its Adam moments / hand back a wrong
    exchange result - the line must say so and the job must exit 5."""
    import torch.distributed as dist
    import torch.nn as nn
    from advmix_amd.dp import GradSync
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29555')
    if backend != 'gloo':
        raise SystemExit('--path replicas is a CPU self-test: ADVMIX_BENCH_BACKEND=gloo')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 2, 1))
    opt = torch.optim.Adam(net.parameters(), 1e-2)
    sync = GradSync(bucket_mb=0.001)
    sync.broadcast_state([net], [opt])
    sync.trace = []
    corrupt = os.environ.get('ADVMIX_BENCH_CORRUPT', '')
    lv = 0.0
    for step in range(3):
        opt.zero_grad()
        loss = net(torch.randn(4, 3, 8, 8)).square().mean()
        loss.backward()
        flat = torch.cat([p.grad.view(-1) for p in net.parameters()])
        sync.reduce_async(flat, 0, flat.numel())
        if corrupt == 'exchange' and rank == 1 and step == 1:
            sync.trace[-1][4][3] += 1.0
        off = 0
        for p in net.parameters():
            p.grad.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        opt.step()
        lv = float(loss)
    with torch.no_grad():
        if corrupt == 'weight' and rank == 1:
            next(net.parameters()).view(-1)[5] += 1e-7
        if corrupt == 'nan' and rank == 1:
            opt.state[next(net.parameters())]['exp_avg'].view(-1)[0] = float('nan')
    verification = sync.verify_trace()
    line = {'metric': 'replicas self-test', 'n_gpus': a.gpus, 'rccl_ranks': dist.get_world_size(), 'backend': backend,
            'corrupt': corrupt or None}
    failed = dp_verdict(line, sync, [opt], (verification[0], {}), lv)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if failed:
        raise SystemExit(5)



def verify_data_parallel(step, args, nets, crit, opts, data, sync, steps=3):
    """An N-rank run proves itself (the driver is the only one who can run RCCL with N > 1).  nn.DataParallel
    re-broadcasts GPU 0's weights before every forward (tools/train.py:69,106,109), so the reference cannot drift or train
    on a bad exchange; this design could, silently.  ``steps`` steps through the execution under test itself (``step``: the
    seven-graph runner or the eager pieces the timed region uses - the SAME object, no second capture), each checked:
      exchange  - what every all-reduce left in the flat gradient buffer == the mean of what the ranks handed to it
                  (dp.GradSync.verify_trace: all-gather of the operands; exact for sum-and-scale, <= 1e-4 of the range's largest element for RCCL's AVG);
      coverage  - the exchanged ranges tile each flat gradient buffer exactly once;
      operands  - what this rank handed to the exchange is finite and is the gradient: recomputed from the same state WITHOUT
                  pieces, side stream or graphs (plain backward).  The two evaluations differ by the order of their fp32 / fp64
                  atomics and by the ReLU masks those flip (the timed region is not the deterministic mode: observed 0.018 at
                  init_weights()), so D's operands are held to a relative L2 distance of 0.2 - a missing (1.0), partial or
                  garbage operand is caught, one that is merely a step old on this constant batch is not: that is what the
                  exact exchange check and the replica fold after the timed steps are for; G's gradient at init_weights() is
                  rounding noise through the frozen D, so it is held to finiteness and to the NORM of the
                  recomputed one within a factor of ten.
    With one rank (ADVMIX_FORCE_SYNC=1) the exchange is the identity and the operand checks still hold the ordering of
    graphs, pieces and side stream to the plain step.  Leaves the models where the verified steps left them."""
    from advmix_amd.graph import _snapshot, _restore
    from advmix_amd.core.function import advmix_phase_a, advmix_phase_b
    D, G, T = nets
    optD, optG = opts
    views, tgt, tw = data
    worst = {'exchange': 0.0, 'operands_D_rel_l2': 0.0, 'operands_G_norm_ratio': 1.0}
    ok = {'exchange': True, 'coverage': True, 'operands_finite': True}

    def l2(a, b):
        nb = float(b.double().norm())
        d = float((a.double() - b.double()).norm())
        return d / nb if nb > 0 else (0.0 if d == 0 else float('inf'))

    try:
        for _ in range(steps):
            before = _snapshot([D, G, T], [optD, optG])
            sync.trace = []
            step()
            torch.cuda.synchronize()
            trace = sync.trace
            after = _snapshot([D, G, T], [optD, optG])
            e_ok, e_worst = sync.verify_trace()
            sync.trace = None
            ok['exchange'] &= e_ok
            worst['exchange'] = max(worst['exchange'], e_worst)
            for opt in (optD, optG):
                rs = sorted((lo, hi) for f, lo, hi, _a, _b in trace if f is opt.flat_grads)
                ok['coverage'] &= bool(rs) and rs[0][0] == 0 and rs[-1][1] == opt.flat_grads.numel() and \
                    all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            ok['operands_finite'] &= all(bool(torch.isfinite(pre).all()) for _f, _lo, _hi, pre, _post in trace)
            _restore(before)                                # the same state, the plain way
            _l, tmp = advmix_phase_a(args, D, G, T, crit, optD, views, tgt, tw)
            mine = torch.cat([pre for f, lo, hi, pre, post in sorted(trace, key=lambda e: e[1]) if f is optD.flat_grads])
            worst['operands_D_rel_l2'] = max(worst['operands_D_rel_l2'], l2(mine, optD.flat_grads))
            for f, lo, hi, pre, post in trace:
                if f is optD.flat_grads:
                    optD.flat_grads[lo:hi].copy_(post)      # adopt the exchanged gradient
            advmix_phase_b(args, D, crit, optD, optG, tmp, tgt, tw)
            mine = torch.cat([pre for f, lo, hi, pre, post in sorted(trace, key=lambda e: e[1]) if f is optG.flat_grads])
            n_mine, n_ref = float(mine.double().norm()), float(optG.flat_grads.double().norm())
            ratio = n_mine / n_ref if n_ref > 0 else (1.0 if n_mine == 0 else float('inf'))
            if not (ratio == ratio):
                ratio = float('inf')
            worst['operands_G_norm_ratio'] = max(worst['operands_G_norm_ratio'], ratio, 1.0 / ratio if ratio > 0 else float('inf'))
            torch.cuda.synchronize()
            del tmp, mine
            _restore(after)                                 # go on from where the execution under test is
        ok['operands_D'] = worst['operands_D_rel_l2'] <= 0.2     # (observed 0.018: atomics order + flipped ReLU masks at init_weights())
        ok['operands_G'] = worst['operands_G_norm_ratio'] <= 10.0
    finally:
        sync.trace = None
    verdict = torch.tensor([0.0 if all(ok.values()) else 1.0], device=views[0].device)
    if sync.world > 1:
        import torch.distributed as dist
        with sync.off_null(verdict):
            dist.all_reduce(verdict, op=dist.ReduceOp.MAX)  # one answer for the job
    return float(verdict.item()) == 0.0, {'steps': steps, 'checks': ok,
                                          'worst': {k: float('%.3g' % v) for k, v in worst.items()}}


SYNC_METRICS = os.environ.get('ADVMIX_SYNC_METRICS') == '1'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)          # discard >= 10 warm-up steps, average >= 50
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='hrnet_w32', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=32, help='images per GPU (TRAIN.BATCH_SIZE_PER_GPU)')
    ap.add_argument('--exec', dest='exec_mode', default='graph', choices=['graph', 'eager'])
    ap.add_argument('--path', default='train', choices=['train', 'validate', 'inputs', 'nms', 'rendezvous', 'replicas'],
                    help='train = the headline AdvMix step; validate = the validate() batch body (SURVEY 8 f1); '
                         'inputs = the device input pipeline (SURVEY 8 f2); rendezvous = start the ranks, one all-reduce, '
                         'report (launcher self-test; ADVMIX_BENCH_BACKEND=gloo runs it without GPUs)')
    ap.add_argument('--through-loop', action='store_true',
                    help='time core.function.train_advmix itself (the drop-in entry point): pinned host batches, H2D copies, '
                         'graph replay, loss.item(), accuracy - the reference loop body lib/core/function.py:107-197')
    ap.add_argument('--dump-shapes', default=None, metavar='CSV',
                    help='log kernel template, grid, shape and FLOPs of every MFMA launch of ONE eager step (for '
                         'tools/kernel_shapes.py) and exit')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-verify', action='store_true', help='skip the data-parallel self-verification steps (N > 1 ranks)')
    ap.add_argument('--no-through-loop', action='store_true', help='skip the extra train_advmix-with-H2D measurement')
    ap.add_argument('--no-roofline', action='store_true')
    a = ap.parse_args()
    if a.dump_shapes:
        os.environ['ADVMIX_TRACE_SHAPES'] = os.path.abspath(a.dump_shapes)

    import torch.distributed as dist
    backend = os.environ.get('ADVMIX_BENCH_BACKEND', 'nccl')      # 'gloo' only for --path rendezvous (CPU launcher test)
    # ADVMIX_BENCH_SHARE_GPU=1: a FUNCTIONAL run of the N-rank path on a box with fewer GPUs - every rank uses cuda:0 and the
    # gradient exchange goes over gloo (RCCL refuses two ranks on one device).  The line says so; it is not a scaling number.
    share_gpu = os.environ.get('ADVMIX_BENCH_SHARE_GPU') == '1'
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        # One command, N ranks (the reference's multi-GPU entry is one command too: GPUS in the YAML ->
        # nn.DataParallel, tools/train.py:69,106,109).  This parent has NOT touched the GPU; it starts one fresh
        # process per GPU, relays rank 0's JSON line and fails loudly rather than run fewer ranks than asked for.
        from advmix_amd.launch import spawn_ranks
        need = not (a.path in ('rendezvous', 'replicas') and backend == 'gloo') and not share_gpu
        raise SystemExit(spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], a.gpus, need_gpus=need))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit('WORLD_SIZE %d != --gpus %d' % (world, a.gpus))
    if a.path == 'rendezvous':
        return rendezvous(a, backend, rank, world, local)
    if a.path == 'replicas':
        return replicas_selftest(a, backend, rank, world, local)
    if backend != 'nccl':
        raise SystemExit('ADVMIX_BENCH_BACKEND=%s is only for --path rendezvous' % backend)
    if share_gpu:
        local, backend = 0, 'gloo'
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    force_sync = os.environ.get('ADVMIX_FORCE_SYNC') == '1'     # exercise the RCCL path with a single rank
    if world > 1 or force_sync:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29555')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if a.path in ('validate', 'inputs', 'nms'):
        line = {'validate': bench_validate, 'inputs': bench_inputs, 'nms': bench_nms}[a.path](a, device, rank, world)
        if world > 1 or force_sync:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            sys.stdout.flush()
            print(json.dumps(line), flush=True)
        return

    from advmix_amd.core.function import advmix_step
    from advmix_amd.core.evaluate import accuracy
    from advmix_amd.dp import GradSync
    from advmix_amd.graph import AdvMixGraphRunner

    # Everything from here on runs on a CREATED stream (as core.function's loops do): graph replays, exchanges, barriers and
    # the verification collectives inherit it - none is ever issued with the NULL stream current.
    main_stream = torch.cuda.Stream(device=device)
    main_stream.wait_stream(torch.cuda.current_stream(device))
    stream_ctx = torch.cuda.stream(main_stream)
    stream_ctx.__enter__()
    net, extra, J, H, W, downs, gflop_img = WORKLOADS[a.workload]
    cfg, D, G, T, crit, optD, optG = build_models(a.workload, device)
    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    views, tgt, tw = synth(a.batch, J, H, W, device, 1234 + rank)
    sync = GradSync(force=force_sync) if (world > 1 or force_sync) else None
    if sync is not None:
        sync.broadcast_state([D, G, T], [optD, optG])          # every replica starts from rank 0's weights / Adam state

    if a.dump_shapes:
        from advmix_amd import ops as _ops
        advmix_step(args, D, G, T, crit, optD, optG, views, tgt, tw, sync)          # lazy buffers first
        torch.cuda.synchronize()
        _ops.set_option('trace_shapes', 1)
        advmix_step(args, D, G, T, crit, optD, optG, views, tgt, tw, sync)
        torch.cuda.synchronize()
        _ops.set_option('trace_shapes', 0)
        print('wrote', a.dump_shapes)
        return
    loop_note = None
    from advmix_amd.core import function as F_
    from advmix_amd.core.evaluate import PendingAccuracy
    if world > 1 and a.exec_mode == 'graph' and not F_.DP_GRAPH:
        a.exec_mode = 'eager'                               # core.function.DP_GRAPH (ADVMIX_DP_GRAPH=0): the line says which

    def make_step(holder=None):
        """The execution the timed region uses: the HIP-graph runner (seven graphs with data parallelism) or the eager step."""
        if a.exec_mode == 'graph':
            runner = AdvMixGraphRunner(args, D, G, T, crit, optD, optG, views, tgt, tw, sync)
            if holder is not None:
                holder['runner'] = runner
            return lambda: runner.step() + (runner.target,)
        return lambda: advmix_step(args, D, G, T, crit, optD, optG, views, tgt, tw, sync) + (tgt,)

    def through_loop(n_warm, n_timed):
        """The drop-in entry point itself: train_advmix over a loader of pinned HOST batches (DataLoader(pin_memory=True) in
        tools/train.py:295-301), i.e. H2D copies, capture on the first batch, replay, loss.item(), accuracy, meters -
        SURVEY 8 d1's full step."""
        import logging
        logging.getLogger(F_.__name__).setLevel(logging.WARNING)
        cfg['PRINT_FREQ'] = 10 ** 9
        host = []
        for k in range(4):                                  # four distinct pinned batches, cycled
            v, t, w = synth(a.batch, J, H, W, torch.device('cpu'), 1234 + rank + 100 * k)
            host.append(([x.pin_memory() for x in v], [t.pin_memory()] * 3, [w.pin_memory()] * 3, [{}, {}, {}]))

        class Loader:
            def __init__(self, n):
                self.n = n

            def __len__(self):
                return self.n

            def __iter__(self):
                return (host[i % len(host)] for i in range(self.n))

        seen = {}

        def run_loop(n):
            wd = {'writer': types.SimpleNamespace(add_scalar=lambda k, v, s: seen.__setitem__(k, float(v))),
                  'train_global_steps': 0}
            F_.train_advmix(cfg, args, Loader(n), [D, G, T], crit, [optD, optG], 0, '', '', wd, sync)
        old_exec = F_.GRAPH_EXEC
        F_.GRAPH_EXEC = a.exec_mode == 'graph'
        try:
            run_loop(max(n_warm, 3))                        # capture + warm-up
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_loop(n_timed)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            return time.perf_counter() - t0, seen.get('train_loss', float('nan')), seen.get('train_acc', 0.0)
        finally:
            F_.GRAPH_EXEC = old_exec
            F_.release_graphs()

    def max_over_ranks(x):
        if world > 1:
            tmax = torch.tensor([x], device=device, dtype=torch.float64)
            with sync.off_null(tmax):                       # (no collective with the NULL stream current, dp.GradSync.off_null)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            return float(tmax.item())
        return x

    verification = None
    dt_loop = None
    if a.through_loop:
        dt, lv, acc = through_loop(a.warmup, a.steps)
        loop_note = 'core.function.train_advmix over pinned host batches (H2D inside the timed region)'
    else:
        hold = {}
        step = make_step(hold)
        if sync is not None and not a.no_verify:
            # before anything is timed: the N-rank execution proves itself - three steps through THIS runner
            verification = verify_data_parallel(step, args, (D, G, T), crit, (optD, optG), (views, tgt, tw), sync)

        def launch():
            loss_D, out, target = step()
            return PendingAccuracy(out, target, loss_D)     # function.py:167-168, device half enqueued

        def run_steps(n):
            """n steps; loss.item() and accuracy() of every step are read - one step late, while the next one runs (the
            loop mirror core.function.train_advmix does the same), the last one before returning."""
            pend, lv, acc = None, float('nan'), 0.0
            for _ in range(n):
                cur = launch()
                if SYNC_METRICS:                            # A/B switch: read every step's numbers before the next launch
                    cur.get()
                if pend is not None:
                    _, acc, _, _, lv = pend.get()
                pend = cur
            if pend is not None:
                _, acc, _, _, lv = pend.get()
            return lv, acc

        run_steps(a.warmup)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lv, acc = run_steps(a.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if not a.no_through_loop and sync is None:
            # (one rank: a second capture in one process is fine there; with the data-parallel runner it is not attempted)
            # SURVEY 8 d1's step includes the H2D of step 1: the same workload through train_advmix itself, beside the
            # resident-input figure (never instead of it)
            hold.clear()
            del step
            torch.cuda.empty_cache()
            dt_loop, _lv2, _acc2 = through_loop(10, 50 if a.steps >= 20 else a.steps)
            dt_loop = (max_over_ranks(dt_loop), 50 if a.steps >= 20 else a.steps)
    dt = max_over_ranks(dt)
    replicas = sync.replicas_state([optD, optG]) if sync is not None else None
    from advmix_amd._lib import lib as _hiplib
    variant = _hiplib.advmix_build_flags()                  # non-zero: a tools/build_variant.sh library (ADVMIX_SO=...)
    if not (lv == lv) and not variant:
        raise SystemExit('loss is NaN')

    line = None
    if rank == 0:
        ms = dt / a.steps * 1e3
        value = a.batch * world * a.steps / dt
        line = {
            'metric': 'images/sec AdvMix train step (HRNet-W32 256x192)' if a.workload == 'hrnet_w32'
            else 'images/sec AdvMix train step (%s)' % a.workload,
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'rccl_ranks': _ranks(), 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic (N(0,1) views, Gaussian targets), random-init weights',
            'config': {'workload': '%s_%dx%d_advmix' % (a.workload, H, W), 'batch_per_gpu': a.batch,
                       'global_batch': a.batch * world, 'generator': 'UnetGenerator(9,3,%d)' % downs,
                       'parallelism': 'dp%d' % world, 'exec': 'hipgraph' if a.exec_mode == 'graph' else 'eager',
                       'entry': loop_note or 'graph.AdvMixGraphRunner.step (inputs resident in HBM)',
                       'step_gflop_per_image': gflop_img},
        }
        if a.workload in NO_ORACLE:
            line['config']['parity'] = NO_ORACLE[a.workload]
        if share_gpu:
            line['config']['shared_gpu'] = ('all %d ranks on cuda:0, gradient exchange over gloo: a functional run of the '
                                            'data-parallel path, NOT a scaling measurement' % world)
        line.update({
            'step_tflops_per_gpu': round(value / world * gflop_img / 1e3, 2),
            'step_frac_of_fp32_mfma_peak': round(value / world * gflop_img / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4),
            'last_loss_D': round(lv, 6) if lv == lv else None,
        })
        if dt_loop is not None:
            line['value_through_loop'] = round(a.batch * world * dt_loop[1] / dt_loop[0], 2)
            line['through_loop'] = {'entry': 'core.function.train_advmix over pinned host batches: H2D copies, graph replay, '
                                             'loss.item(), accuracy inside the timed region (SURVEY 8 d1)',
                                    'steps': dt_loop[1], 'warmup': 10, 'ms_per_step': round(dt_loop[0] / dt_loop[1] * 1e3, 3)}
        if sync is not None:
            line['replicas_identical'], line['all_finite'] = replicas['identical'], replicas['finite'] and (lv == lv)
            line['grad_exchange_verified'] = verification[0] if verification is not None else None
            line['dp_verification'] = dict(verification[1], exec=line['config']['exec'],
                                           transport=dist.get_backend()) if verification is not None else 'skipped (--no-verify)'
        if variant:
            line['INVALID_variant_build_flags'] = variant   # measurement build: never a benchmark result
        if not a.no_roofline:
            widths = HRNET_STAGES.get('hrnet_w48' if a.workload == 'hrnet_w48' else 'hrnet_w32')
            line['roofline'] = time_conv_family(a.batch, device, family=tuple(
                (c, (H // 4) >> i, (W // 4) >> i) for i, c in enumerate(widths)))
        if world == 1 and not a.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(a.workload)


    failed = sync is not None and (not replicas['identical'] or not replicas['finite']
                                   or (verification is not None and not verification[0]))
    if world > 1 or force_sync:
        dist.barrier()
    torch.cuda.synchronize()
    stream_ctx.__exit__(None, None, None)
    if world > 1 or force_sync:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so the JSON line is the LAST line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)
    if failed:
        raise SystemExit(5)                                 # the line says which of the three checks failed


if __name__ == '__main__':
    main()
