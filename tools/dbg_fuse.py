import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
import advmix_amd.ops as ops
from advmix_amd.plan import hrnet_plan, PlanNet
from test_ops_gpu import rnd, cl, dev
extra = {'FINAL_CONV_KERNEL': 1,
         'STAGE2': {'NUM_MODULES': 1, 'NUM_BRANCHES': 2, 'BLOCK': 'BASIC', 'NUM_BLOCKS': [1, 1], 'NUM_CHANNELS': [32, 64], 'FUSE_METHOD': 'SUM'},
         'STAGE3': {'NUM_MODULES': 2, 'NUM_BRANCHES': 3, 'BLOCK': 'BASIC', 'NUM_BLOCKS': [1, 1, 1], 'NUM_CHANNELS': [32, 64, 128], 'FUSE_METHOD': 'SUM'},
         'STAGE4': {'NUM_MODULES': 1, 'NUM_BRANCHES': 4, 'BLOCK': 'BASIC', 'NUM_BLOCKS': [1, 1, 1, 1], 'NUM_CHANNELS': [32, 64, 128, 256], 'FUSE_METHOD': 'SUM'}}
P = hrnet_plan(extra, 5)
torch.manual_seed(11)
net = PlanNet(P)
with torch.no_grad():
    for k, p in net.named_parameters():
        if p.dim() == 1 and k.endswith('.weight'): p.uniform_(0.6, 1.4)
        elif p.dim() == 1: p.normal_(0, 0.2)
net = net.to(dev()).train()
B, H, W = 4, 64, 64
xin = rnd(B, 3, H, W, seed=91); dyo = rnd(B, 5, H // 4, W // 4, seed=92)
got = {}
for mode in ('all', 'nofuse', 'none'):
    ops.BNB_FUSED = mode != 'none'; ops.FUSE_BNB = mode == 'all'
    for p in net.parameters(): p.grad = None
    xg = cl(xin).requires_grad_(True)
    y = net(xg); y.backward(cl(dyo)); torch.cuda.synchronize()
    got[mode] = {k: p.grad.detach().cpu().double() for k, p in net.named_parameters()}
    got[mode]['x'] = xg.grad.detach().cpu().double()
    print(mode, ops.COUNTERS)
order = sorted(got['all'], key=lambda k: getattr(dict(net.named_parameters()).get(k), '_flat_rank', -1))
for k in order:
    a, b, c = got['all'][k], got['nofuse'][k], got['none'][k]
    sc = float(c.abs().max()) + 1e-30
    e1, e2 = float((a - c).abs().max()) / sc, float((b - c).abs().max()) / sc
    if e1 > 1e-4 or e2 > 1e-4:
        print('%-45s all-vs-none %.2e  nofuse-vs-none %.2e' % (k, e1, e2))
