"""Shared fixture builders (oracle side). Mirrors oracle/gen_golden.py's recipe."""
import json
import os
import numpy as np
import torch

from oracle import detinit, configs
from oracle.posenet import posenet_spec, calibrate, trainable
from oracle.unet import unet_spec, unet_transposed_names
from oracle.synth import synth_batch, strided, checksum

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

CASES = {
    'hrnet_tiny': ('pose_hrnet', configs.HRNET_TINY, 5, 2, 64, 64, 3),
    'resnet18_tiny': ('pose_resnet', configs.RES18_TINY, 5, 2, 64, 64, 3),
    'hrnet_w32': ('pose_hrnet', configs.HRNET_W32, 17, 2, 256, 192, 2),
    'resnet50': ('pose_resnet', configs.RES50, 17, 2, 256, 192, 2),
    # BASELINE.json configs[3] (C4): HRNet-W48 384x288 with UnetGenerator(9, 3, 5)
    'hrnet_w48': ('pose_hrnet', configs.HRNET_W48, 17, 2, 384, 288, 2),
}
# forward / backward vectors only (no loop fixtures): BASELINE.json configs[4] as far as the reference's code goes - its
# pose_hrnet + UnetGenerator(9, 3, 6) at 512x512 (128x128x32 ... 16x16x256 branch maps); HigherHRNet itself has no code
FORWARD_ONLY_CASES = {'hrnet_w32_512': ('pose_hrnet', configs.HRNET_W32, 17, 2, 512, 512, 0)}
ALL_FORWARD = dict(CASES, **FORWARD_ONLY_CASES)
DOWNS = {'hrnet_w48': 5}                        # U-Net depth per case (default 6; tools/_init_parse.py:132-134)
GOLD_FILES = {'hrnet_w48': ('c4_forward.npz', 'c4_advmix_steps.npz', 'c4_advmix_checksums.json'),
              'hrnet_w32_512': ('c5_trunk_forward.npz', None, None)}


def gold_files(tag):
    return GOLD_FILES.get(tag, ('forward.npz', 'advmix_steps.npz', 'advmix_checksums.json'))


def gold_json(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def gold_npz(name):
    return np.load(os.path.join(GOLD, name))


def build_states(net, extra, J, unet_downs=6, salt=0):
    """{key: tensor} dicts for D, teacher, G exactly as gen_golden.build_ref_models."""
    dspec = posenet_spec(net, extra, J)
    gspec = unet_spec(9, 3, unet_downs)
    detinit.mark_transposed(unet_transposed_names(9, 3, unet_downs))
    D = detinit.fill_state_dict(dspec, salt=salt)
    T = detinit.fill_state_dict(dspec, salt=salt + 1)
    G = detinit.fill_state_dict(gspec, salt=salt + 2, gain=0.5)
    return D, T, G


def close(a, b, atol=1e-3, rtol=1e-3):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), 'max abs err %.3e (tol %.3e) at %d/%d elements; max |ref| %.3e' % (
        err.max(), tol[bad].min() if bad.any() else 0, bad.sum(), bad.size, np.abs(b).max())


def checksum_close(got, want, rtol=2e-3):
    """sum / abs-sum pairs: tolerance relative to the abs-sum (never a hash)."""
    for k, (s, a) in want.items():
        gs, ga = got[k]
        assert abs(ga - a) <= rtol * max(a, 1e-6), (k, ga, a)
        assert abs(gs - s) <= rtol * max(a, 1e-6), (k, gs, s)
