"""One AdvMix / plain training step on CPU (oracle; test infra only).

``advmix_step`` follows lib/core/function.py:129-171 (the body of
``train_advmix``'s batch loop); ``plain_step`` follows :48-64.  ``Adam`` restates
``torch.optim.Adam(params, lr)`` as built by lib/utils/utils.py:89-92
(betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).
"""
import math
import torch
import torch.nn.functional as F

from .posenet import posenet_forward, trainable
from .unet import unet_forward
from .loss import joints_loss, accuracy


class Adam:
    def __init__(self, P, names, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.P, self.names = P, list(names)
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.t = 0
        self.m = {k: torch.zeros_like(P[k]) for k in self.names}
        self.v = {k: torch.zeros_like(P[k]) for k in self.names}

    @torch.no_grad()
    def step(self, grads):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for k, g in zip(self.names, grads):
            if g is None:
                continue
            m, v = self.m[k], self.v[k]
            m.lerp_(g, 1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            self.P[k].addcdiv_(m, denom, value=-self.lr / bc1)


def _set_grad(P, names, flag):
    for k in names:
        P[k].requires_grad_(flag)


def mix_views(inputs, logits):
    """function.py:138-144: softmax over the 3 generator maps, per-pixel convex mix."""
    w = F.softmax(logits, dim=1)
    tmp = inputs[0] * w[:, 0:1]
    for k in range(1, len(inputs)):
        tmp = tmp + inputs[k] * w[:, k:k + 1]
    return tmp, w


def advmix_step(net, extra, D, G, T, optD, optG, inputs, target, tw,
                alpha=0.1, adv_loss_weight=1.0, use_target_weight=True,
                unet_kw=None, after_D_step=None):
    """Returns dict(loss_D, loss_G, out1, out2, teacher, tmp, avg_acc, ...).
    ``after_D_step`` (test hook, not reference behaviour): called right after the D update so a
    parity test can overwrite D with the device path's updated weights ("teacher forcing"):
    Adam's first steps are ~lr*sign(g), so elements whose gradient is rounding noise move in
    implementation-dependent directions and an unforced comparison measures that, not kernels."""
    unet_kw = unet_kw or {}
    dn, gn = optD.names, optG.names
    _set_grad(G, gn, True)
    logits = unet_forward(G, torch.cat(inputs, 1), **unet_kw)           # :137-138
    tmp, mixw = mix_views(inputs, logits)                                # :142-144

    _set_grad(D, dn, True)                                               # :140
    out1 = posenet_forward(net, D, tmp.detach(), extra, True)            # :146
    with torch.no_grad():
        teacher = posenet_forward(net, T, inputs[0], extra, False)       # :148-149
    l_hm = joints_loss(out1, target, tw, use_target_weight)
    l_kd = joints_loss(out1, teacher, tw, use_target_weight)
    loss_D = l_hm * (1 - alpha) + l_kd * alpha                           # :151-153
    gD = torch.autograd.grad(loss_D, [D[k] for k in dn], allow_unused=True)
    optD.step(gD)                                                        # :154-155
    if after_D_step is not None:
        after_D_step()

    _set_grad(D, dn, False)                                              # :158
    out2 = posenet_forward(net, D, tmp, extra, True)                     # :160 (updated D)
    loss_G = -joints_loss(out2, target, tw, use_target_weight) * adv_loss_weight
    gG = torch.autograd.grad(loss_G, [G[k] for k in gn], allow_unused=True)
    optG.step(gG)                                                        # :163-164
    _set_grad(G, gn, False)
    _, avg_acc, cnt, pred = accuracy(out2, target)                       # :168
    return dict(loss_D=loss_D.detach(), loss_G=loss_G.detach(), l_hm=l_hm.detach(),
                l_kd=l_kd.detach(), out1=out1.detach(), out2=out2.detach(),
                teacher=teacher, tmp=tmp.detach(), mixw=mixw.detach(),
                logits=logits.detach(), gD=dict(zip(dn, gD)), gG=dict(zip(gn, gG)),
                avg_acc=avg_acc, cnt=cnt, pred=pred)


def plain_step(net, extra, D, optD, x, target, tw, use_target_weight=True):
    """function.py:48-64 (non-AdvMix ``train``)."""
    dn = optD.names
    _set_grad(D, dn, True)
    out = posenet_forward(net, D, x, extra, True)
    loss = joints_loss(out, target, tw, use_target_weight)
    g = torch.autograd.grad(loss, [D[k] for k in dn], allow_unused=True)
    optD.step(g)
    _set_grad(D, dn, False)
    _, avg_acc, cnt, pred = accuracy(out, target)
    return dict(loss=loss.detach(), out=out.detach(), g=dict(zip(dn, g)),
                avg_acc=avg_acc, cnt=cnt, pred=pred)
