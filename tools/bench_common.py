"""What bench.py and the measurement tools share: the workloads (BASELINE.json's configs), synthetic inputs,
model construction as tools/train.py does it, HIP-event timing on the launching stream, and the CPU-baseline child process."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


