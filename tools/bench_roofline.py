"""bench.py's ``roofline`` object: the step's MFMA conv family timed live through the C ABI (HIP events on the launching
stream), each member on its ALGORITHMIC (direct-convolution) FLOPs against the fp32 matrix peak; the in-step time shares
come from the committed per-shape profile of the same step (tools/profile_step.sh -> profiles/rNN_per_shape_1lane.csv)."""
import json
import os

import torch

from bench_common import ROOT, FP32_MFMA_PEAK_TFLOPS, _event_time

CONV_FAMILY = ((32, 64, 48), (64, 32, 24), (128, 16, 12), (256, 8, 6))     # HRNet-W32's 3x3 s1 C->C convs: 1.81 GF each at B = 32
# launches per AdvMix step of each kind of a given conv: two train-mode student forwards + the eval-mode teacher, two
# input gradients (D step, G step) - of a BasicBlock's two convs one takes the residual path's gradient as addend and the
# sign of y from the bit mask, the other recomputes it from c - one weight gradient (D step)
KIND_WEIGHT = {'fwd+BN-sums': 2, 'fwd+BN-eval+ReLU': 1, 'dgrad+addend+BN-bwd-sums (act mask)': 1,
               'dgrad+BN-bwd-sums (sign from c)': 1, 'wgrad': 1}


WINOGRAD_MULTIPLY_SAVING = 2.25    # F(2x2,3x3) and F(3x3,2x2): 16 multiplies where the direct form has 36


def branch_convs_per_pass(workload):
    """{channels: number of 3x3 / stride 1 C -> C convs of the network's BRANCHES in one pass} from the plan the product
    executes (plan.hrnet_plan: HRNet-W32 64 / 64 / 56 / 24) - what a family member's launches-per-step weight is multiplied by
    (without it ``dominant`` was the member with the largest weight x time of ONE conv per shape)."""
    from bench_common import WORKLOADS
    from advmix_amd.plan import hrnet_plan
    net, extra, J = WORKLOADS[workload][:3]
    if net != 'pose_hrnet':
        return {}
    P = hrnet_plan(extra, J)
    shape = {n: sh for n, sh, _k in P.params}
    out = {}
    for st in P.steps:
        if st[0] == 'conv' and st[1].startswith('stage') and '.branches.' in st[1]:
            co, ci, r, s_ = shape[st[1] + '.weight']
            if (r, s_) == (3, 3) and st[4] == 1 and co == ci:
                out[co] = out.get(co, 0) + 1
    return out


def _profile_file(kind, pattern):
    """(parsed JSON or None, repo-relative path) of the profile summary of ``kind``: the file profiles/LATEST.json names (written
    by tools/collect_r06.sh: the collection's tag, git head and the files it produced - VERDICT r5 weak 9: with hundreds of
    files under profiles/ the 'latest set' is machine-selected), else the last match of ``pattern`` in name order."""
    import glob
    path = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'LATEST.json')) as f:
            path = json.load(f).get('files', {}).get(kind)
    except (OSError, ValueError):
        pass
    if path and os.path.exists(os.path.join(ROOT, path)):
        full = os.path.join(ROOT, path)
    else:
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))
        if not files:
            return None, None
        full = files[-1]
    if full.endswith('.json'):
        with open(full) as f:
            return json.load(f), os.path.relpath(full, ROOT)
    return full, os.path.relpath(full, ROOT)


def _pmc_file(pattern, kind=None):
    return _profile_file(kind or pattern, pattern)


def _time_shares():
    """{class: (share of the step's kernel time, time-weighted fraction of the fp32 matrix peak)} from the newest committed
    per-shape profile of the headline step (serialized one-lane trace joined with the library's launch log:
    tools/profile_step.sh, tools/kernel_shapes.py)."""
    import csv
    full, rel = _profile_file('per_shape_1lane', 'r*_per_shape_1lane.csv')
    if not full:
        return None
    cls = {}
    for r in csv.DictReader(open(full)):
        k, shp = r['kernel'], r['shape']
        share = float(r['share_of_kernel_time'] or 0)
        if k.startswith('ALL MFMA'):
            continue
        tf = float(r['tflops'] or 0)
        if k.startswith('wino4'):                            # (csrc/conv_wino4.hip's input / output / filter transforms: no FLOPs counted)
            c = "U-Net 4x4 s2 convs in the Winograd domain: transforms"
        elif ' 1x1 ' in shp and shp.endswith('(B 16)'):       # (its 16 GEMMs, one launch of conv_direct / conv_wgrad_group over 16
            c = "U-Net 4x4 s2 convs in the Winograd domain: the 16 GEMMs (their own FLOPs, 2.25x fewer than the direct form's)"   # 'images')
        elif 'wgrad' in shp and ' x8 ' in shp:
            c = 'grouped weight gradients'
        elif 'wgrad' in shp or k.startswith(('conv_wgrad', 'wgrad')):
            c = 'single weight gradients'
        elif k.startswith('conv_wino'):
            c = 'branch 3x3 convs, Winograd kernel (32 / 64 / 128 channels)'
        elif k.startswith('conv_smap'):                     # (conv_smap / conv_smapw)
            c = 'branch 3x3 convs, image-per-workgroup kernel (256 channels @8x6)'
        elif k.startswith(('conv_direct', 'conv_group', 'conv_igemm')) and ' 3x3 s1 ' in shp and shp.split(' s1 ')[1].split(' ')[0].split('->')[0] == shp.split(' s1 ')[1].split(' ')[0].split('->')[1]:
            c = 'branch 3x3 convs, direct kernel (256 channels)'
        elif k.startswith(('conv_direct', 'conv_group', 'conv_igemm', 'conv_tr', 'deconv')):
            c = 'other convs'
        elif k.startswith(('norm_apply_slots', 'norm_bwd_apply_slots')):
            c = 'BatchNorm slot kernels'
        else:
            c = 'other'
        e = cls.setdefault(c, [0.0, 0.0, 0.0])
        e[0] += share
        if tf > 0:
            e[1] += share
            e[2] += share * tf                              # sum of FLOPs / sum of time = the time-weighted mean of FLOP/s
    out = {c: {'share_of_kernel_time': round(v[0], 4),
               'frac_of_fp32_mfma_peak_time_weighted': round(v[2] / v[1] / FP32_MFMA_PEAK_TFLOPS, 4) if v[1] > 0 else None}
           for c, v in sorted(cls.items(), key=lambda kv: -kv[1][0])}
    out['source'] = rel
    return out


def time_conv_family(B, device, iters=100, family=None, workload='hrnet_w32'):
    """The roofline object.  The step's time is the MFMA convs'  and no single launch dominates: the four
    branch resolutions of HRNet-W32 each run the same 1.81 GFLOP 3x3 conv as forward (+ BatchNorm column sums, or + eval
    BatchNorm + ReLU for the teacher), input gradient (+ the BatchNorm-backward sums of its producer) and weight gradient.
    Every member is timed live, back to back through the C ABI with HIP events on the launching stream, through the
    kernel the STEP runs for it (round 5: the Winograd kernel for the 32- and 64-channel branches, on the algorithmic
    direct-convolution FLOPs); ``frac`` is the launch-count-weighted FLOP/s of the family against the fp32 matrix peak,
    ``members`` lets each number be recomputed, ``dominant`` is the member with the largest weight x time product.
    ``step_kernel_time_share``: where the step's kernel time goes, by class, from the committed per-shape profile.
    ``hbm_kernels``: the two BatchNorm kernels left on the path against the 8 TB/s HBM peak."""
    import ctypes
    from advmix_amd import ops
    from advmix_amd._lib import call, lib
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())       # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    members, hbm = [], []
    tot_f = tot_t = tot_own = 0.0
    family = family or CONV_FAMILY
    per_pass = branch_convs_per_pass(workload)              # convs of each width in one pass of the network (W32: 64 / 64 / 56 / 24)
    for C, H, W in family:
        rows = B * H * W
        x = torch.randn(B, H, W, C, device=device)
        w = (torch.randn(C, 3, 3, C, device=device) * (9 * C) ** -0.5).permute(0, 3, 1, 2)    # logical OIHW, [O][R][S][I] in memory
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
        wino = bool(ops.WINO and lib.advmix_conv_wino_config(B, H, W, C, C) >= ops.WINO_MIN_WGS)
        smap = bool(ops.WINO and ops.SMAP and C == ops.SMAP_C and lib.advmix_conv_smap_config(B, H, W, C, C) >= ops.WINO_MIN_WGS)
        skind = 'smapw' if (ops.SMAP_WINO and lib.advmix_conv_smapw_config(B, H, W, C, C) >= ops.WINO_MIN_WGS) else 'smap'   # (the images WinoBank makes follow ops.SMAP_WINO)
        smap = smap and (skind == 'smapw' or not ops.SMAP_WINO)
        kfwd, kdg = ('advmix_conv3x3_%s_fwd' % skind, 'advmix_conv3x3_%s_dgrad' % skind) if smap else ('advmix_conv3x3_wino_fwd', 'advmix_conv3x3_wino_dgrad')

        def reset():
            nbg.value = 0
        if wino or smap:                                    # what ops.ConvBN.fwd / ops._conv_dgrad launch for this shape
            bank = ops.WinoBank([w])
            bank.refresh()
            uf, ud = bank.images(w)
            runs = {
                'fwd+BN-sums': lambda: (reset(), call(kfwd, P(x), uf, P(y), B, H, W, C, C, None, None, None, None,
                                                      0.0, None, 0, P(slots), ctypes.byref(nbg), st)),
                'fwd+BN-eval+ReLU': lambda: call(kfwd, P(x), uf, P(y), B, H, W, C, C, P(g), P(b), P(rm), P(rv),
                                                 1e-5, None, 1, None, None, st),
                'dgrad+addend+BN-bwd-sums (act mask)': lambda: (reset(), call(
                    kdg, P(dy), ud, P(c2), P(dx), B, H, W, C, C, P(amask), P(c2), P(mean), P(invstd), None,
                    None, 1, P(slots), ctypes.byref(nbg), st)),
                'dgrad+BN-bwd-sums (sign from c)': lambda: (reset(), call(
                    kdg, P(dy), ud, None, P(dx), B, H, W, C, C, None, P(c2), P(mean), P(invstd), P(g), P(b),
                    1, P(slots), ctypes.byref(nbg), st)),
            }
        else:
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
        wgw = bool(ops.WGRAD_WINO and ops.WINO and lib.advmix_wgrad_wino_config(B, H, W, C, C) * NG >= ops.WGRAD_WINO_MIN_UNITS)
        grouped = wgw or lib.advmix_conv_wgrad_group(NG, ga, gb, gd, B, H, W, C, H, W, C, 3, 3, 1, 1, st) == 0   # (not every width is served)
        if wgw:                                             # what ops._flush_wgrads launches for this geometry (round 5)
            runs['wgrad'] = lambda: call('advmix_conv3x3_wgrad_wino_group', NG, ga, gb, gd, B, H, W, C, C, st)
        elif grouped:
            runs['wgrad'] = lambda: call('advmix_conv_wgrad_group', NG, ga, gb, gd, B, H, W, C, H, W, C, 3, 3, 1, 1, st)
        else:
            runs['wgrad'] = lambda: call('advmix_conv_wgrad', P(dy), P(x), P(dw), B, H, W, C, H, W, C, 3, 3, 1, 1, st)
        for kind, run in runs.items():
            ms, rr = _event_time(run, iters if not (kind == 'wgrad' and grouped) else max(iters // 4, 10))
            if kind == 'wgrad' and grouped:
                ms, rr = ms / NG, [v / NG for v in rr]      # per problem of the eight-problem launch
            wgt = KIND_WEIGHT[kind] * per_pass.get(C, 1)    # launches of this member per AdvMix step
            # Winograd paths multiply 2.25x fewer numbers than the direct form whose FLOPs ``frac`` is quoted on: ``frac`` can
            # exceed 1.0 for them without any work being skipped - ``frac_own`` prices the kernel's OWN
            # multiplies against the same peak, ``ceiling_frac`` is what ``frac`` could reach at all
            wino_path = (kind == 'wgrad' and wgw) or (kind != 'wgrad' and (wino or (smap and skind == 'smapw')))
            own = flops / WINOGRAD_MULTIPLY_SAVING if wino_path else flops
            tot_f += wgt * flops
            tot_own += wgt * own
            tot_t += wgt * ms * 1e-3
            members.append({
                'kernel': '3x3 s1 %d->%d @%dx%d %s' % (C, C, H, W, kind if not (kind == 'wgrad' and grouped) else 'wgrad (1 of 8 problems of one launch)'),
                'path': ('wgrad_wino (Winograd F(3x3,2x2))' if wgw else 'conv_wgrad_group') if kind == 'wgrad' else ('conv_wino (Winograd F(2x2,3x3))' if wino else (('conv_smapw (image per workgroup, Winograd F(2x2,3x3))' if skind == 'smapw' else 'conv_smap (image per workgroup)') if smap else 'conv_direct')),
                'us_per_launch': round(ms * 1e3, 2), 'us_per_launch_runs': [round(v * 1e3, 2) for v in rr],
                'algorithmic_gflop_per_launch': round(flops / 1e9, 3),
                'tflops': round(flops / (ms * 1e-3) / 1e12, 2), 'frac': round(flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                'own_gflop_per_launch': round(own / 1e9, 3),
                'frac_own': round(own / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                'ceiling_frac': WINOGRAD_MULTIPLY_SAVING if wino_path else 1.0,
                'launches_per_step': wgt, 'launches_per_step_x_us': round(wgt * ms * 1e3, 1)})
        if wino or smap:
            bank.release()
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
    headline = B == 32 and family == CONV_FAMILY
    dominant = max(members, key=lambda m: m['launches_per_step_x_us'])      # the member the step spends most time in (launches per step x time)
    traffic = src = None
    if headline:                                            # HBM bytes per launch of the dominant member, when a PMC pass of it is committed
        d, src = _pmc_file('r*_pmc_dominant.json', 'pmc_dominant')
        d = (d or {}).get('members', {}).get(dominant['kernel']) if d and 'members' in d else d      # (one file, keyed by member)
        if d and d.get('kernel') == dominant['kernel']:
            traffic = round(d['hbm_bytes_per_launch'])
        else:
            src = None
    step_util = busy = None
    if headline:                                            # SQ_VALU_MFMA_BUSY_CYCLES summed over one step (tools/pmc_step.sh)
        d, f = _pmc_file('r*_pmc_step_mfma.json', 'pmc_step_mfma')
        if d:
            busy = round(d['mfma_utilisation_of_step'], 4)
            step_util = {'mfma_busy_simd_cycles_per_step': round(d['mfma_busy_cycles_per_step']),
                         'algorithmic_simd_cycles_per_step_direct_form': round(118.58e9 * 32 / 64),
                         'utilisation_at_the_profiled_step_time': busy, 'source': f}
    agg = tot_f / tot_t / 1e12
    Cd = int(dominant['kernel'].split('->')[1].split(' ')[0])
    Hd, Wd = (int(v) for v in dominant['kernel'].split('@')[1].split(' ')[0].split('x'))
    nt = 5 if 'addend' in dominant['kernel'] else (4 if 'dgrad' in dominant['kernel'] else 2)   # tensors of rows x C floats it must move
    algo_bytes = nt * B * Hd * Wd * Cd * 4 + 9 * Cd * Cd * 4
    return {'bound': 'mfma',
            'kernel': 'conv_wino / conv_smapw / wgrad_wino / conv_direct family: 3x3 s1 C->C at the four HRNet branch resolutions of this '
                      'workload x {fwd+BN sums, fwd+BN eval, dgrad+BN-bwd sums, wgrad}, weighted by launches per step (role weight x convs '
                      'of that width per pass: %s), algorithmic (direct-convolution) FLOPs' % per_pass,
            'achieved': round(agg, 3), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(agg / FP32_MFMA_PEAK_TFLOPS, 4),
            # north_star names MFMA utilisation: the two readings side by side.  ``frac`` = algorithmic (direct-form) FLOP/s of the
            # family over the fp32 matrix peak - a throughput; the Winograd members multiply 2.25x fewer numbers, so their ceiling on
            # this scale is 2.25 (``ceiling_frac`` per member).  ``frac_own`` = the family's OWN multiplies over the same peak;
            # ``mfma_busy_frac`` = the MFMA pipe's busy cycles of the WHOLE step by counter (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
            # 2.4 GHz x step time), profiles/: tools/pmc_step.sh).
            'frac_own': round(tot_own / tot_t / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            'mfma_busy_frac': busy,
            'traffic': traffic, 'traffic_unit': 'HBM bytes per launch of the dominant member (rocprofv3 PMC, corrected)',
            'traffic_source': src, 'traffic_algorithmic_bytes': algo_bytes,
            'traffic_ratio': round(traffic / algo_bytes, 3) if traffic else None,
            'step_mfma_utilisation_pmc': step_util,
            'step_kernel_time_share': _time_shares() if headline else None,
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


