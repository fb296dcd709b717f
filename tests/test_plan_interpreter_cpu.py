"""CPU: what the pinned-mask GPU tests take for the truth IS the oracle's function.  tests/test_models_gpu.py evaluates a
plan.Plan's steps in fp64 (conv / deconv / bn / inorm / act / catact / fuse) with the device's ReLU masks; here the same
reading of the steps - with each activation's OWN mask - is held against the oracle's forward passes (oracle/posenet.py,
oracle/unet.py, themselves pinned to the real reference by tests/test_oracle_golden.py), for HRNet-W32, ResNet-50 / the tiny
ResNet-18 and both generator depths, in fp64; and the tapping transformation (_tap_every_relu) is checked to leave the tapped values alone."""
import importlib.util
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]
from helpers import build_states                                             # noqa: E402
from oracle import configs, detinit                                          # noqa: E402

from plan_functional import interpret, activated_slots, pooled_slots           # noqa: E402


def _gpu_test_module():
    spec = importlib.util.spec_from_file_location('_tm_gpu', os.path.join(HERE, 'test_models_gpu.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_plan_steps_read_functionally_are_the_oracles_hrnet():
    from advmix_amd.plan import hrnet_plan
    from oracle.posenet import posenet_forward
    D, _, _ = build_states('pose_hrnet', configs.HRNET_W32, 17)
    P = hrnet_plan(configs.HRNET_W32, 17)
    assert {n for n, _, _ in P.params} | {n for n, _, _ in P.buffers} == set(D)          # the reference's state-dict keys
    x = detinit.normal('interp.hrnet.x', (2, 3, 128, 96)).double()
    W = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in D.items()}
    got = interpret(P, W, x)[P.out]
    want = posenet_forward('pose_hrnet', {k: v.clone() for k, v in W.items()}, x, configs.HRNET_W32, True)
    assert got.shape == want.shape == (2, 17, 32, 24)
    assert float((got - want).abs().max()) <= 1e-12 * float(want.abs().max())


def test_plan_steps_read_functionally_are_the_oracles_resnet():
    from advmix_amd.plan import resnet_plan
    from oracle.posenet import posenet_forward
    for extra, tag in ((configs.RES50, "r50"), (configs.RES18_TINY, "r18")):
        D, _, _ = build_states('pose_resnet', extra, 17)
        P = resnet_plan(extra, 17)
        assert {n for n, _, _ in P.params} | {n for n, _, _ in P.buffers} == set(D)
        x = detinit.normal('interp.%s.x' % tag, (2, 3, 64, 64)).double()
        W = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in D.items()}
        got = interpret(P, W, x)[P.out]
        want = posenet_forward('pose_resnet', {k: v.clone() for k, v in W.items()}, x, extra, True)
        assert got.shape == want.shape and float((got - want).abs().max()) <= 1e-12 * float(want.abs().max())


def test_plan_steps_read_functionally_are_the_oracles_generator():
    from advmix_amd.plan import unet_plan
    from oracle.unet import unet_forward
    for downs, hw in ((6, (128, 64)), (5, (64, 96))):
        _, _, G = build_states('pose_hrnet', configs.HRNET_W32, 17, unet_downs=downs)
        P = unet_plan(9, 3, downs)
        assert {n for n, _, _ in P.params} == set(G)
        x = detinit.normal('interp.unet%d.x' % downs, (2, 9) + hw).double()
        W = {k: v.double() for k, v in G.items()}
        got = interpret(P, W, x)[P.out]
        want = unet_forward(W, x, num_downs=downs)
        assert float((got - want).abs().max()) <= 1e-12 * float(want.abs().max())


def test_tapping_every_activation_leaves_the_network_alone():
    """_tap_every_relu (tests/test_models_gpu.py): every activated slot is tapped, the original steps are untouched, every tap
    reaches the new output (its gradient is non-zero), and the new output has the finest tapped shape."""
    from advmix_amd.plan import hrnet_plan, unet_plan
    tm = _gpu_test_module()
    for make, cin, hw in ((lambda: hrnet_plan(configs.HRNET_W32, 17), 3, (64, 64)), (lambda: unet_plan(9, 3, 6), 9, (128, 64))):
        P0, P = make(), make()
        taps = tm._tap_every_relu(P)
        assert P.steps[:len(P0.steps)] == P0.steps and sorted(taps) == sorted(tm._activated_slots(P0)) and len(taps) >= 11
        assert all(st[0] in ('conv', 'fuse') for st in P.steps[len(P0.steps):])
        g = torch.Generator().manual_seed(5)
        W = {n: (torch.randn(shape, generator=g, dtype=torch.float64) * (0.05 if len(shape) == 4 else 0.1) + (1.0 if kind == 'bn_w' else 0.0))
             for n, shape, kind in P.params}
        x = torch.randn((1, cin) + hw, generator=g, dtype=torch.float64)
        v0, v1 = interpret(P0, W, x), interpret(P, W, x)
        assert all(torch.equal(v0[s_], v1[s_]) for s_ in taps) and torch.equal(v0[P0.out], v1[P0.out])
        for s_ in taps:
            v1[s_].retain_grad() if v1[s_].requires_grad else None
        finest = max(v1[s_].shape[2] for s_ in taps + [P0.out])
        assert v1[P.out].shape[1:] == (32, finest, finest * hw[1] // hw[0])
        # every tap feeds the output: perturbing it moves the output
        xs = {s_: v1[s_].detach().clone().requires_grad_(True) for s_ in taps}
        val = dict(v1)
        val.update(xs)
        for st in P.steps[len(P0.steps):]:
            if st[0] == 'conv':
                val[st[3]] = F.conv2d(val[st[2]], W[st[1] + '.weight'], None, st[4], st[5])
            else:
                val[st[3]] = sum(val[s_] if sh == 0 else F.interpolate(val[s_], scale_factor=2 ** sh, mode='nearest') for s_, sh in zip(st[1], st[2]))
        grads = torch.autograd.grad(val[P.out].sum(), list(xs.values()))
        assert all(float(g_.abs().max()) > 0 for g_ in grads)


def test_probe_tool_runs_without_a_gpu():
    """tools/probe_unet_grads.py (EXPERIMENTS K2) on the CPU: the fp64 / fp32 columns and the pinned pass (without a GPU the
    'hip' column repeats fp32) - the tool shares tests/unet_functional.py with test_generator_gradients_with_pinned_masks."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), 'tools', 'probe_unet_grads.py'), '64', '64', '2', '5'],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert '== against fp64 with the masks pinned to the HIP run' in out.stdout and 'L4 relu' in out.stdout
    rows = [ln.split() for ln in out.stdout.splitlines() if ln.startswith('model.') and ln.split()[0].endswith('.weight')]
    assert len(rows) == 2 * 10
    # first pass: fp32 against fp64, each with its OWN masks - on this container's CPU the fp32 evaluation itself flips a mask
    # at this size (weight gradients up to 2e-1 of scale off); second pass, masks pinned: rounding level on every weight
    print('own masks: worst %.2e   pinned: worst %.2e' % (max(float(r[1]) for r in rows[:10]), max(float(r[1]) for r in rows[10:])))
    assert all(float(r[1]) < 1e-4 for r in rows[10:]), rows[10:]


def test_pinning_an_evaluation_to_its_own_masks_changes_nothing():
    """tests/plan_functional.py: ``interpret(pin=..., pool_src=...)`` handed the slots of an un-pinned evaluation reproduces its
    values AND gradients exactly (ReLU, LeakyReLU, max-pool winners) - and handed the masks of a DIFFERENT input it stays on
    that input's linear piece (the output then differs from the un-pinned one, the gradient is finite and non-zero)."""
    from advmix_amd.plan import resnet_plan, unet_plan
    for P, cin, hw in ((resnet_plan(configs.RES18_TINY, 5), 3, (64, 64)), (unet_plan(9, 3, 5), 9, (64, 96))):
        g = torch.Generator().manual_seed(11)
        W = {n: (torch.randn(shape, generator=g, dtype=torch.float64) * (0.05 if len(shape) == 4 else 0.1) + (1.0 if kind == 'bn_w' else 0.0)).requires_grad_(True)
             for n, shape, kind in P.params}
        x = torch.randn((2, cin) + hw, generator=g, dtype=torch.float64)
        free = interpret(P, W, x)
        pin = {s_: free[s_].detach() for s_, _ in activated_slots(P)}
        pool = {dst: free[src].detach() for src, dst in pooled_slots(P)}
        assert len(pin) >= 8 and (len(pool) == 1) == (cin == 3)
        held = interpret(P, W, x, pin=pin, pool_src=pool)
        assert torch.equal(held[P.out], free[P.out])
        proj = torch.randn(free[P.out].shape, generator=g, dtype=torch.float64)
        ga = torch.autograd.grad((free[P.out] * proj).sum(), list(W.values()), allow_unused=True)
        gb = torch.autograd.grad((held[P.out] * proj).sum(), list(W.values()), allow_unused=True)
        for n, a_, b_ in zip(W, ga, gb):
            assert (a_ is None) == (b_ is None) and (a_ is None or float((a_ - b_).abs().max()) <= 1e-12 * (float(a_.abs().max()) + 1e-300)), n
        x2 = x + 0.05 * torch.randn(x.shape, generator=g, dtype=torch.float64)
        other = interpret(P, W, x2, pin=pin, pool_src=pool)[P.out]
        assert not torch.equal(other, interpret(P, W, x2)[P.out]) and bool(torch.isfinite(other).all())
