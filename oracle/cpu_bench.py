"""Time the CPU oracle's AdvMix step on the host cores (bench.py's cpu_baseline leg runs this
in a child process with a hard timeout).  Test infrastructure; prints one JSON line."""
import json
import os
import sys
import time


def effective_cpus(cap=32):
    """CPUs this process may really use: affinity mask, cgroup quota and a cap (an OpenMP team
    wider than the quota spins instead of working)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                    n = min(n, max(1, q // p))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, cap))


def validate_main(workload, budget, cores):
    """One validate batch (function.py:223-300, FLIP_TEST/SHIFT_HEATMAP/POST_PROCESS on) on the host."""
    import numpy as np
    from oracle import detinit, configs
    from oracle.posenet import posenet_spec, calibrate
    from oracle.synth import synth_batch, synth_boxes
    from oracle import validate as ov
    W = {'hrnet_w32': ('pose_hrnet', configs.HRNET_W32, 17, 256, 192),
         'hrnet_w48': ('pose_hrnet', configs.HRNET_W48, 17, 384, 288),
         'resnet50': ('pose_resnet', configs.RES50, 17, 256, 192)}
    net, extra, J, H, Wd = W[workload]
    D = detinit.fill_state_dict(posenet_spec(net, extra, J))
    B = 8
    v, t, w = synth_batch('bench.cpu.val', B, J, H, Wd)
    c, s, score = synth_boxes('bench.cpu.valbox', B)
    calibrate(net, D, v[0], extra)
    pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]

    def batch():
        out, loss, _, _ = ov.validate_batch(net, extra, D, v[0], t, w, pairs, True, True)
        ov.collect(out, c, s, score, True)
    t0 = time.time()
    batch()
    warm = time.time() - t0
    n, t0 = 0, time.time()
    while n < 16 and (n == 0 or time.time() - t0 + warm < budget):
        batch()
        n += 1
    dt = time.time() - t0
    print(json.dumps({'value': round(B * n / dt, 3), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
                      'sample': '%s validate batch (flip test + final preds), B=%d, %d timed batches after 1 warm-up '
                                '(%.1fs), torch CPU fp32 + numpy oracle, %d threads (os.cpu_count=%d)' % (
                                    workload, B, n, warm, cores, os.cpu_count() or 0)}), flush=True)


def inputs_main(workload, budget, cores):
    """The reference-style per-sample host pipeline: 3 x ToTensor+Normalize, GridMask, one target render."""
    import numpy as np
    from oracle import inputpipe as ip
    H, W = (384, 288) if workload == 'hrnet_w48' else (256, 192)
    B, J = 32, 17
    base, aug, jt, vis = ip.synth_samples('bench.cpu.inp', B, J, H, W)
    rng = np.random.RandomState(5)
    import random
    from oracle import autoaug as oa
    prng = random.Random(5)

    def batch():
        for b in range(B):
            v0 = ip.to_tensor_normalize(base[b])
            ip.to_tensor_normalize(oa.autoaug_pil(base[b], oa.draw_policy(prng)))  # the AutoAugment view through PIL (advaug.py:180-187)
            ip.grid_aug(ip.to_tensor_normalize(base[b]), jt[b], vis[b], ip.grid_draws(H, W, 0.5, 0.7, 1, rng), J)
            ip.generate_target(jt[b], vis[b], (W, H), (W // 4, H // 4), 2)
        return v0
    batch()
    n, t0 = 0, time.time()
    while n < 50 and (n == 0 or time.time() - t0 < budget):
        batch()
        n += 1
    dt = time.time() - t0
    print(json.dumps({'value': round(B * n / dt, 1), 'unit': 'images/sec', 'cores': 1, 'kind': 'port',
                      'sample': 'per-sample host pipeline (AutoAugment, 3x ToTensor+Normalize, GridMask, generate_target), %dx%d, '
                                '%d batches of %d in one process (a DataLoader worker)' % (H, W, n, B)}), flush=True)


def nms_main(budget):
    """The oracle's restatements of the native box NMS (C, the .cu semantics) and of oks_nms (numpy) on one image."""
    import numpy as np
    from oracle import nms as onms
    rng = np.random.RandomState(11)
    N = 1000
    xy = rng.rand(N, 2) * 400
    wh = rng.rand(N, 2) * 120 + 10
    dets = np.concatenate([xy, xy + wh, rng.rand(N, 1)], 1).astype(np.float32)
    people = []
    base = rng.rand(6, 17, 2) * 300 + 50
    for n in range(30):
        k = np.zeros((17, 3)); k[:, :2] = base[n % 6] + rng.randn(17, 2) * 4; k[:, 2] = rng.rand(17)
        people.append({'keypoints': k.reshape(-1), 'area': float(rng.rand() * 20000 + 5000), 'score': float(rng.rand())})
    onms.build()
    onms.gpu_nms(dets, 0.5); onms.oks_nms(people, 0.9)
    n, t0 = 0, time.time()
    while n < 2000 and time.time() - t0 < budget:
        onms.gpu_nms(dets, 0.5); onms.oks_nms(people, 0.9)
        n += 1
    dt = time.time() - t0
    print(json.dumps({'value': round(n / dt, 1), 'unit': 'images/sec', 'cores': 1, 'kind': 'port',
                      'sample': 'C restatement of the box NMS (N=1000) + numpy oks_nms (30 persons), %d images, 1 thread' % n}),
          flush=True)


def main():
    workload, budget = sys.argv[1], float(sys.argv[2])
    cores = effective_cpus()
    os.environ.setdefault('OMP_NUM_THREADS', str(cores))
    import torch
    torch.set_num_threads(cores)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if len(sys.argv) > 3 and sys.argv[3] == 'validate':
        return validate_main(workload, budget, cores)
    if len(sys.argv) > 3 and sys.argv[3] == 'nms':
        return nms_main(10.0)
    if len(sys.argv) > 3 and sys.argv[3] == 'inputs':
        torch.set_num_threads(1)
        return inputs_main(workload, 10.0, 1)
    from oracle import detinit, configs
    from oracle.posenet import posenet_spec, trainable
    from oracle.unet import unet_spec, unet_transposed_names
    from oracle.step import Adam, advmix_step
    from oracle.synth import synth_batch
    W = {'hrnet_w32': ('pose_hrnet', configs.HRNET_W32, 17, 256, 192, 6),
         'hrnet_w48': ('pose_hrnet', configs.HRNET_W48, 17, 384, 288, 5),
         'resnet50': ('pose_resnet', configs.RES50, 17, 256, 192, 6),
         'hrnet_w32_512': ('pose_hrnet', configs.HRNET_W32, 17, 512, 512, 6)}    # (bench.py: throughput only, no reference code)
    net, extra, J, H, Wd, downs = W[workload]
    detinit.mark_transposed(unet_transposed_names(9, 3, downs))
    D = detinit.fill_state_dict(posenet_spec(net, extra, J))
    T = {k: v.clone() for k, v in D.items()}
    G = detinit.fill_state_dict(unet_spec(9, 3, downs), gain=0.5)
    oD, oG = Adam(D, trainable(D)), Adam(G, list(G))
    B = 4
    v, t, w = synth_batch('bench.cpu', B, J, H, Wd)
    kw = dict(unet_kw={'num_downs': downs})
    t0 = time.time()
    for _ in range(3):                                                 # >= 3 warm-up + >= 10 timed steps
        advmix_step(net, extra, D, G, T, oD, oG, v, t, w, **kw)
    warm = time.time() - t0
    n, t0 = 0, time.time()
    while n < 10 or (n < 16 and time.time() - t0 + warm < budget):
        advmix_step(net, extra, D, G, T, oD, oG, v, t, w, **kw)
        n += 1
    dt = time.time() - t0
    print(json.dumps({'value': round(B * n / dt, 3), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
                      'sample': '%s AdvMix step, B=%d, %d timed steps after 3 warm-up steps (%.1fs), torch CPU fp32 '
                                'oracle, %d threads (os.cpu_count=%d)' % (workload, B, n, warm, cores,
                                                                           os.cpu_count() or 0)}), flush=True)


if __name__ == '__main__':
    main()
