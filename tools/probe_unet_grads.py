#!/usr/bin/env python3
"""Where does the generator's gradient error come from?  The 6-down U-Net of tests' ``hrnet_w32_512`` case (512x512, B = 2)
built THREE times from the same functional description (tests/unet_functional.py) - torch CPU fp64 (the truth), torch CPU
fp32 (the oracle's arithmetic), and the HIP library through advmix_amd.ops' functional spellings - with every intermediate
tensor kept.  Prints, per intermediate and per parameter, max|value - fp64| and max|gradient - fp64| relative to the fp64
tensor's max, for the fp32 oracle and for HIP: first against fp64 with its own activation masks (the first line where a
column jumps is a flipped ReLU mask), then against fp64 with the masks PINNED to the HIP run's signs (the kernels' arithmetic
alone; the fp32 column then carries the oracle's flips against the device instead).  profiles/EXPERIMENTS.md K2.
usage: probe_unet_grads.py [H=512] [W=512] [B=2] [downs=6]     (ADVMIX_* switches apply)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
from oracle import detinit, configs
from helpers import build_states
from unet_functional import run

H, W, B, downs = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 512), (2, 512), (3, 2), (4, 6)))
_, _, G = build_states('pose_hrnet', configs.HRNET_W32, 17, unet_downs=downs)
x0 = torch.cat([detinit.normal('hrnet_w32_512.view%d' % k, (B, 3, H, W)) for k in range(3)], 1)
proj = detinit.normal('hrnet_w32_512.gproj', (B, 3, H, W))

v64, g64, p64, order = run('f64', G, x0, proj, downs)
v32, g32, p32, _ = run('f32', G, x0, proj, downs)
vh, gh, ph, _ = run('hip', G, x0, proj, downs) if torch.cuda.is_available() else (v32, g32, p32, None)   # (no GPU: the table's shape only)
rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-300))


def table(title, v64, g64, p64):
    print(title)
    print('%-36s %23s   %23s' % ('intermediate', 'value err (fp32 | hip)', 'gradient err (fp32 | hip)'))
    for n in order:
        print('%-36s %10.2e | %10.2e   %10.2e | %10.2e' % (n, rel(v32[n], v64[n]), rel(vh[n], v64[n]),
                                                           rel(g32[n], g64[n]) if n in g64 else float('nan'),
                                                           rel(gh[n], g64[n]) if n in gh and n in g64 else float('nan')))
    print('%-60s %s' % ('parameter gradient', 'fp32 | hip'))
    for k in G:
        print('%-60s %10.2e | %10.2e   (|g64| max %.2e)' % (k, rel(p32[k], p64[k]), rel(ph[k], p64[k]), float(p64[k].abs().max())))


table('== against fp64 with its OWN activation masks', v64, g64, p64)
pv, pg_, pp, _ = run('f64', G, x0, proj, downs, pin={n: t.float() for n, t in vh.items()})
table('== against fp64 with the masks pinned to the HIP run', pv, pg_, pp)
