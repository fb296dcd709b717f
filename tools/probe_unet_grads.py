#!/usr/bin/env python3
"""Where does the generator's gradient error come from?  The 6-down U-Net of tests' ``hrnet_w32_512`` case (512x512, B = 2)
built THREE times from the same functional description - torch CPU fp64 (the truth), torch CPU fp32 (the oracle's
arithmetic), and the HIP library through advmix_amd.ops' functional spellings (conv2d / instance_norm / conv_transpose2d /
cat_act: the launches plan.unet_plan makes) - with every intermediate tensor kept.  Prints, per intermediate and per
parameter, max|value - fp64| and max|gradient - fp64| relative to the fp64 tensor's max, for the fp32 oracle and for HIP:
the first line where HIP leaves the oracle's error level is where to look.
usage: probe_unet_grads.py [H=512] [W=512] [B=2] [downs=6]     (ADVMIX_* switches apply)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
import torch.nn.functional as F
from oracle import detinit
from oracle.unet import unet_levels
from helpers import build_states
from oracle import configs

H, W, B, downs = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 512), (2, 512), (3, 2), (4, 6)))
_, _, G = build_states('pose_hrnet', configs.HRNET_W32, 17, unet_downs=downs)
views = [detinit.normal('hrnet_w32_512.view%d' % k, (B, 3, H, W)) for k in range(3)]
x0 = torch.cat(views, 1)
lv = unet_levels(9, 3, downs)


def build(P, x, be):
    """The network of oracle/unet.py / plan.unet_plan on backend ``be``; returns (output, [(name, tensor)])."""
    rec = []

    def R(name, t):
        if t.requires_grad:
            t.retain_grad()
        rec.append((name, t))
        return t

    def names(i):
        L = lv[i]
        return '%s.model.%d' % (L['pre'], L['di']), '%s.model.%d' % (L['pre'], L['ui'])

    def level(i, a):
        dn, un = names(i)
        d = R('L%d conv out' % i, be['conv'](a, P[dn + '.weight'], P[dn + '.bias']))
        if i == downs - 1:
            r = R('L%d relu' % i, be['relu'](d))
        else:
            r = level(i + 1, R('L%d inorm+leaky (skip of L%d)' % (i, i + 1), be['inorm_leaky'](d)))
        dc = R('L%d deconv out' % i, be['deconv'](r, P[un + '.weight'], P[un + '.bias']))
        u = R('L%d up inorm' % i, be['inorm'](dc))
        return R('L%d relu(cat)' % i, be['cat_relu'](a, u))

    dn, un = names(0)
    d = R('L0 conv out', be['conv'](x, P[dn + '.weight'], P[dn + '.bias']))
    r = level(1, R('L0 leaky (skip of L1)', be['leaky'](d)))
    return R('L0 deconv out (logits)', be['deconv'](r, P[un + '.weight'], P[un + '.bias'])), rec


def torch_backend():
    return dict(conv=lambda x, w, b: F.conv2d(x, w, b, 2, 1), deconv=lambda x, w, b: F.conv_transpose2d(x, w, b, 2, 1),
                relu=F.relu, leaky=lambda x: F.leaky_relu(x, 0.2), inorm=lambda x: F.instance_norm(x, eps=1e-5),
                inorm_leaky=lambda x: F.leaky_relu(F.instance_norm(x, eps=1e-5), 0.2),
                cat_relu=lambda a, u: F.relu(torch.cat([a, u], 1)))


def hip_backend():
    from advmix_amd import ops
    return dict(conv=lambda x, w, b: ops.conv2d(x, w, b, 2, 1), deconv=lambda x, w, b: ops.conv_transpose2d(x, w, b, 2, 1),
                relu=lambda x: ops.activation(x, ops.ACT_RELU), leaky=lambda x: ops.activation(x, ops.ACT_LEAKY),
                inorm=lambda x: ops.instance_norm(x, ops.ACT_NONE), inorm_leaky=lambda x: ops.instance_norm(x, ops.ACT_LEAKY),
                cat_relu=lambda a, u: ops.cat_act(a, u, ops.ACT_RELU))


def run(kind):
    if kind == 'hip':
        from advmix_amd import ops
        P = {k: v.detach().clone().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) if v.dim() == 4
             else v.detach().clone().cuda().requires_grad_(True) for k, v in G.items()}
        bank = ops.WinoBank([p for p in P.values() if p.dim() == 4 and p.shape[0] % 16 == 0 and p.shape[1] % 16 == 0])       # (the filter images plan.PlanNet keeps)
        bank.refresh()
        x = x0.cuda().contiguous(memory_format=torch.channels_last)
        out, rec = build(P, x, hip_backend())
        proj = detinit.normal('hrnet_w32_512.gproj', out.shape).cuda()
    else:
        dt = torch.float64 if kind == 'f64' else torch.float32
        P = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in G.items()}
        out, rec = build(P, x0.to(dt), torch_backend())
        proj = detinit.normal('hrnet_w32_512.gproj', out.shape).to(dt)
    (out * proj).sum().backward()
    vals = {n: t.detach().double().cpu() for n, t in rec}
    grads = {n: t.grad.detach().double().cpu() for n, t in rec if t.grad is not None}
    pg = {k: p.grad.detach().double().cpu() for k, p in P.items()}
    if kind == 'hip':
        torch.cuda.synchronize()
        bank.release()
    return vals, grads, pg, [n for n, _ in rec]


v64, g64, p64, order = run('f64')
v32, g32, p32, _ = run('f32')
vh, gh, ph, _ = run('hip') if torch.cuda.is_available() else (v32, g32, p32, None)     # (no GPU: the table's shape only)
rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-300))
print('%-36s %23s   %23s' % ('intermediate', 'value err (fp32 | hip)', 'gradient err (fp32 | hip)'))
for n in order:
    print('%-36s %10.2e | %10.2e   %10.2e | %10.2e' % (n, rel(v32[n], v64[n]), rel(vh[n], v64[n]),
                                                       rel(g32[n], g64[n]) if n in g64 else float('nan'),
                                                       rel(gh[n], g64[n]) if n in gh and n in g64 else float('nan')))
print('%-60s %s' % ('parameter gradient', 'fp32 | hip'))
for k in G:
    print('%-60s %10.2e | %10.2e   (|g64| max %.2e)' % (k, rel(p32[k], p64[k]), rel(ph[k], p64[k]), float(p64[k].abs().max())))
