"""Deterministic, name-keyed tensor generator shared by the fixture generator
and the tests (own counter-based stream: numpy Philox keyed by crc32(name)).

Used instead of the reference's ``init_weights`` N(0, 1e-3) because that init
makes eval-mode outputs ~1e-10 and an absolute 1e-3 parity bound vacuous
(SURVEY.md §7 "Parity signal is tiny under reference init").
"""
import zlib
import numpy as np
import torch


def _gen(name, salt=0):
    key = (zlib.crc32(name.encode()) + 0x9E3779B1 * salt) & 0xFFFFFFFFFFFFFFFF
    return np.random.Generator(np.random.Philox(key=key))


def normal(name, shape, std=1.0, mean=0.0, salt=0):
    a = _gen(name, salt).standard_normal(tuple(shape), dtype=np.float32)
    return torch.from_numpy(a * np.float32(std) + np.float32(mean))


def uniform(name, shape, lo=0.0, hi=1.0, salt=0):
    a = _gen(name, salt).random(tuple(shape), dtype=np.float32)
    return torch.from_numpy(a * np.float32(hi - lo) + np.float32(lo))


def fill_state_dict(spec, salt=0, gain=1.0):
    """spec: iterable of (name, shape). Returns {name: tensor} with a
    non-degenerate init: conv/deconv weights fan-out Kaiming-like, BN affine
    near (1, 0), running stats near (0, 1), biases small."""
    out = {}
    spec = list(spec)
    hrnet = any(n.startswith('stage2.') for n, _ in spec)      # (its head sees un-normalised fuse sums: larger inputs)
    for name, shape in spec:
        shape = tuple(shape)
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            out[name] = torch.zeros((), dtype=torch.int64)
        elif leaf == 'running_mean':
            out[name] = normal(name, shape, 0.1, salt=salt)
        elif leaf == 'running_var':
            out[name] = uniform(name, shape, 0.5, 1.5, salt=salt)
        elif len(shape) == 4:
            fan = shape[0] * shape[2] * shape[3]
            if 'deconv' in name or _is_transposed(name):
                fan = shape[1] * shape[2] * shape[3] / 4.0
            std = gain * (2.0 / fan) ** 0.5
            if name.startswith('final_layer'):
                # the head: fan-IN scaling, so that the heat-maps of the seeded networks are O(1) like real ones and
                # the north-star bound (1e-3 abs + 1e-3 rel per element) is meaningful.  (Fan-out scaling over the 17
                # joint channels gave heat-maps of magnitude 20-90, where an ABSOLUTE 1e-3 is 1e-5 of the scale:
                # below what any fp32 evaluation of a 100-layer network reproduces.)
                std = (0.045 if hrnet else 0.25) * (1.0 / (shape[1] * shape[2] * shape[3])) ** 0.5
            out[name] = normal(name, shape, std, salt=salt)
        elif leaf == 'weight':          # BN gamma
            out[name] = normal(name, shape, 0.1, 1.0, salt=salt)
        else:                           # biases / BN beta
            out[name] = normal(name, shape, 0.05, salt=salt)
    return out


_TRANSPOSED = set()


def mark_transposed(names):
    _TRANSPOSED.update(names)


def _is_transposed(name):
    return name in _TRANSPOSED
