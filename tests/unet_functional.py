"""The augmentation generator (oracle/unet.py = lib/models/Unet_generator.py:13-112) from ONE functional description on three
backends - torch CPU fp64 / fp32 and the HIP library through advmix_amd.ops' functional spellings (the op classes
plan.unet_plan launches) - with every intermediate tensor kept (test infrastructure; tools/probe_unet_grads.py prints it).

``pin``: the fp64 / fp32 evaluation takes its ReLU / LeakyReLU masks from the signs of another run's activations.  An
InstanceNorm output within rounding of zero takes either side of the ReLU in ANY fp32 evaluation; when that element carries
a large gradient, the weight gradient in front of it moves by 1e-2 of its scale (profiles/EXPERIMENTS.md K2) - which says
nothing about the kernels.  With the masks pinned to the device's, what is compared is the arithmetic alone."""
import torch
import torch.nn.functional as F
from oracle.unet import unet_levels


def build(P, x, be, downs):
    """Returns (logits, [(name, tensor)]) in forward order."""
    lv = unet_levels(9, 3, downs)
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
            r = R('L%d relu' % i, be['relu'](d, 'L%d relu' % i))
        else:
            n = 'L%d inorm+leaky (skip of L%d)' % (i, i + 1)
            r = level(i + 1, R(n, be['inorm_leaky'](d, n)))
        dc = R('L%d deconv out' % i, be['deconv'](r, P[un + '.weight'], P[un + '.bias']))
        u = R('L%d up inorm' % i, be['inorm'](dc))
        return R('L%d relu(cat)' % i, be['cat_relu'](a, u, 'L%d relu(cat)' % i))

    dn, un = names(0)
    d = R('L0 conv out', be['conv'](x, P[dn + '.weight'], P[dn + '.bias']))
    r = level(1, R('L0 leaky (skip of L1)', be['leaky'](d, 'L0 leaky (skip of L1)')))
    return R('L0 deconv out (logits)', be['deconv'](r, P[un + '.weight'], P[un + '.bias'])), rec


def torch_backend(pin=None):
    def relu(x, name):
        return F.relu(x) if pin is None else x * (pin[name] > 0).to(x.dtype)

    def leaky(x, name):
        return F.leaky_relu(x, 0.2) if pin is None else x * torch.where(pin[name] > 0, 1.0, 0.2).to(x.dtype)
    return dict(conv=lambda x, w, b: F.conv2d(x, w, b, 2, 1), deconv=lambda x, w, b: F.conv_transpose2d(x, w, b, 2, 1),
                relu=relu, leaky=leaky, inorm=lambda x: F.instance_norm(x, eps=1e-5),
                inorm_leaky=lambda x, name: leaky(F.instance_norm(x, eps=1e-5), name),
                cat_relu=lambda a, u, name: relu(torch.cat([a, u], 1), name))


def hip_backend():
    from advmix_amd import ops
    return dict(conv=lambda x, w, b: ops.conv2d(x, w, b, 2, 1), deconv=lambda x, w, b: ops.conv_transpose2d(x, w, b, 2, 1),
                relu=lambda x, name: ops.activation(x, ops.ACT_RELU), leaky=lambda x, name: ops.activation(x, ops.ACT_LEAKY),
                inorm=lambda x: ops.instance_norm(x, ops.ACT_NONE), inorm_leaky=lambda x, name: ops.instance_norm(x, ops.ACT_LEAKY),
                cat_relu=lambda a, u, name: ops.cat_act(a, u, ops.ACT_RELU))


def run(kind, G, x0, proj, downs, pin=None):
    """One forward + backward of sum(logits * proj) on backend ``kind`` ('f64', 'f32', 'hip'): ({name: value},
    {name: gradient}, {parameter: gradient}, [names in forward order]), everything as float64 CPU tensors."""
    if kind == 'hip':
        from advmix_amd import ops
        P = {k: (v.detach().clone().cuda().contiguous(memory_format=torch.channels_last) if v.dim() == 4
                 else v.detach().clone().cuda()).requires_grad_(True) for k, v in G.items()}
        bank = ops.WinoBank([p for p in P.values() if p.dim() == 4 and p.shape[0] % 16 == 0 and p.shape[1] % 16 == 0])
        bank.refresh()                                      # (the filter images plan.PlanNet keeps beside the weights)
        out, rec = build(P, x0.cuda().contiguous(memory_format=torch.channels_last), hip_backend(), downs)
        proj = proj.cuda()
    else:
        dt = torch.float64 if kind == 'f64' else torch.float32
        P = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in G.items()}
        out, rec = build(P, x0.to(dt), torch_backend(pin), downs)
        proj = proj.to(dt)
    (out * proj).sum().backward()
    vals = {n: t.detach().double().cpu() for n, t in rec}
    grads = {n: t.grad.detach().double().cpu() for n, t in rec if t.grad is not None}
    pg = {k: p.grad.detach().double().cpu() for k, p in P.items()}
    if kind == 'hip':
        torch.cuda.synchronize()
        bank.release()
    return vals, grads, pg, [n for n, _ in rec]
