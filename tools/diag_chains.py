"""Where does test_launch_chains_equal_the_level_schedule's input-gradient difference come from?  The same HRNET_TINY forward /
backward under: the chain schedule twice (run-to-run noise), the chain schedule with the BatchNorm-backward epilogue off, with
the activation mask off (residual layers unfused), and the level schedule.  Prints max |dx_a - dx_b| / max |dx_b|."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle import configs
from oracle.synth import synth_batch
from helpers import build_states
from smoke_step import product_models
from advmix_amd import plan as plan_mod, ops
from advmix_amd.core.loss import JointsMSELoss

net, extra, J, B, H, W = 'pose_hrnet', configs.HRNET_TINY, 5, 2, 64, 64
D_sd, T_sd, G_sd = build_states(net, extra, J, salt=40)
v, t, w = synth_batch('chains.check', B, J, H, W)


def run(chains=True, bnb=True, mask=True):
    plan_mod.CHAINS, ops.BNB_FUSED, ops.ACT_MASK = chains, bnb, mask
    cfg, D, G, _ = product_models(net, extra, J, D_sd, T_sd, G_sd)
    D.train()
    x = v[0].cuda().requires_grad_(True)
    out = D(x)
    JointsMSELoss(True).cuda()(out, t.cuda(), w.cuda()).backward()
    torch.cuda.synchronize()
    plan_mod.CHAINS, ops.BNB_FUSED, ops.ACT_MASK = True, True, True
    return out.detach().cpu().double(), x.grad.detach().cpu().double(), ops.COUNTERS.get('bnb', 0)


def rel(a, b):
    return float((a - b).abs().max()) / float(b.abs().max())


base = run()
rows = [('chains again', run()), ('chains, no bnb epilogue', run(bnb=False)), ('chains, no act mask', run(mask=False)),
        ('levels', run(chains=False)), ('levels again', run(chains=False))]
for name, r in rows:
    print('%-26s out %.2e  dx %.2e' % (name, rel(r[0], base[0]), rel(r[1], base[1])))
print('levels vs levels again     dx %.2e' % rel(rows[3][1][1], rows[4][1][1]))
