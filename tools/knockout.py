"""Wall-time bounds by knock-out (NOT valid steps: what ANY improvement of an ingredient can buy at most).  Same process,
interleaved A B A B on the headline workload (HRNet-W32 256x192, B = 32, HIP-graph replay):
  wgrad_all      every weight-gradient launch dropped
  wgrad_c64plus  the weight gradients of the 64- / 128- / 256-channel 3x3 convs only (conv_wgrad<2,2,true,1,1>)
  wgrad_small    the weight gradients below 0.5 GFLOP only (the fuse layers' strided 3x3 and 1x1 convs: ~57 launches of 8-23 us)
  norm_lowres    norm_apply_slots / norm_bwd_apply_slots launches on the three low-resolution branches (rows <= 32 x 32 x 24:
                 VERDICT r3 item 5) dropped - an upper bound for grouping them (a grouped launch still does their work)
  small_convs    every direct-kernel conv launch (forward / input gradient) below 0.45 GFLOP dropped: the ~90 strided / 1x1 / fuse
                 convs of 13-23 us (round 5)
  norm_all       EVERY norm_apply_slots / norm_bwd_apply_slots launch dropped (round 5: what fusing them away could buy at most)
usage: python tools/knockout.py [steps] [modes, comma separated]"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from advmix_amd import ops
from advmix_amd.graph import AdvMixGraphRunner

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
real_wgrad, real_call, real_lib = ops._wgrad, ops.call, ops.lib
MODE = {'v': 'base'}


def wgrad(st, lane, a, b, w, geom):
    if MODE['v'] == 'wgrad_all':
        return
    if MODE['v'] == 'wgrad_c64plus' and w.shape[2] == 3 and w.shape[0] >= 64 and w.shape[0] == w.shape[1]:
        return
    if MODE['v'] == 'wgrad_small':                          # B, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad
        if 2.0 * geom[0] * geom[1] * geom[2] * geom[3] * geom[6] * geom[7] * geom[8] < 0.5e9:
            return
    return real_wgrad(st, lane, a, b, w, geom)


def call(name, *a):
    if MODE['v'] == 'norm_lowres' and name == 'advmix_norm_bwd_apply_slots' and a[7] <= 32 * 32 * 24:
        return 0
    if MODE['v'] == 'norm_all' and name == 'advmix_norm_bwd_apply_slots':
        return 0
    return real_call(name, *a)


class Lib:
    def __getattr__(self, k):
        f = getattr(real_lib, k)
        if k in ('advmix_conv_fwd_ex', 'advmix_conv_tr_w_bnb', 'advmix_conv_tr_w_add'):
            def h(*a):
                # (x, w, bias|addend, y, N, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride, pad, ...): FLOPs of the launch
                fl = 2.0 * a[4] * max(a[5] * a[6], a[8] * a[9]) / (a[13] * a[13]) * a[7] * a[10] * a[11] * a[12]
                if MODE['v'] == 'small_convs' and fl < 0.45e9:
                    if hasattr(a[-2], '_obj'):              # (the slot count the skipped launch would have reported)
                        a[-2]._obj.value = 16
                    return 0
                return f(*a)
            return h
        if k == 'advmix_norm_apply_slots':
            def g(*a):
                if MODE['v'] == 'norm_lowres' and a[3] <= 32 * 32 * 24:
                    return 0
                if MODE['v'] == 'norm_all':
                    return 0
                return f(*a)
            return g
        return f


ops._wgrad, ops.call, ops.lib = wgrad, call, Lib()


def measure(mode):
    MODE['v'] = mode
    cfg, D, G, T, crit, optD, optG = bench.build_models('hrnet_w32', dev)
    net, extra, J, H, W, downs, _ = bench.WORKLOADS['hrnet_w32']
    views, tgt, tw = bench.synth(32, J, H, W, dev, 1234)
    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        runner = AdvMixGraphRunner(args, D, G, T, crit, optD, optG, views, tgt, tw)
        for _ in range(5):
            runner.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    del runner
    torch.cuda.empty_cache()
    return ms


for rep in range(2):
    for mode in (sys.argv[2].split(',') if len(sys.argv) > 2 else ('base', 'wgrad_all', 'wgrad_c64plus', 'wgrad_small', 'norm_lowres', 'norm_all')):
        print('%-14s %.2f ms/step' % (mode, measure(mode)), flush=True)
