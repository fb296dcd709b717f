"""TEST INFRASTRUCTURE ONLY - CPU restatement of the AutoAugment view of the three-view input pipeline
(SURVEY.md 8 f2 remainder): lib/dataset/advaug.py:10-108 (``ImageNetPolicy`` / ``SubPolicy``), applied per sample
at lib/dataset/JointsDataset.py:124.

The policy table (advaug.py:22-35) only ever reaches five Pillow operations - equalize, posterize, solarize, invert,
sharpness - so those are what is restated, in numpy, bit for bit:
  * ``ImageOps.equalize / posterize / solarize / invert``: 256-entry look-up tables (Pillow's ImageOps.py is Python;
    the integer arithmetic below is its published algorithm);
  * ``ImageEnhance.Sharpness(img).enhance(f)`` = ``Image.blend(img.filter(ImageFilter.SMOOTH), img, f)``: Pillow's C
    code (Filter.c 3x3 kernel in float32 with a +0.5 offset, border pixels copied; Blend.c float32 interpolation /
    clipped extrapolation, truncating).  Pillow is a third-party dependency (requirements.txt:11, unpinned); the
    container has 12.2.0, and ``tests/test_oracle_golden.py`` pins this file against it (random images, every operation
    and magnitude) and against what the REAL ``ImageNetPolicy`` returned for recorded draws (tests/golden/autoaug.npz).

``draw_policy`` replays the reference's use of Python's ``random`` module call for call, so a worker that seeds like
the reference draws the same sub-policy, the same coin flips and the same sharpness sign."""
import numpy as np

EQUALIZE, POSTERIZE, SOLARIZE, INVERT, SHARPNESS = 1, 2, 3, 4, 5
OP_CODE = {'equalize': EQUALIZE, 'posterize': POSTERIZE, 'solarize': SOLARIZE, 'invert': INVERT, 'sharpness': SHARPNESS}

# advaug.py:22-35 (data): (p1, op1, magnitude index 1, p2, op2, magnitude index 2)
POLICIES = (
    (0.8, 'equalize', 8, 0.6, 'equalize', 3), (0.6, 'posterize', 7, 0.6, 'posterize', 6),
    (0.4, 'equalize', 7, 0.2, 'solarize', 4), (0.6, 'solarize', 3, 0.6, 'equalize', 7),
    (0.8, 'posterize', 5, 1.0, 'equalize', 2), (0.6, 'equalize', 8, 0.4, 'posterize', 6),
    (0.0, 'equalize', 7, 0.8, 'equalize', 8), (0.6, 'invert', 4, 1.0, 'equalize', 8),
    (0.4, 'sharpness', 7, 0.6, 'invert', 8), (0.4, 'equalize', 7, 0.2, 'solarize', 4),
    (0.6, 'invert', 4, 1.0, 'equalize', 8), (0.8, 'equalize', 8, 0.6, 'equalize', 3),
)


def magnitude(op, idx):
    """advaug.py:50-65: the ``ranges`` table for the five reachable operations."""
    if op == 'posterize':
        return int(np.round(np.linspace(8, 4, 10), 0).astype(int)[idx])
    if op == 'solarize':
        return float(np.linspace(256, 0, 10)[idx])
    if op == 'sharpness':
        return float(np.linspace(0.0, 0.9, 10)[idx])
    return 0


def draw_policy(rng):
    """advaug.py:38-40 + :102-105 with ``rng`` = Python's ``random`` module (or a random.Random): returns the list of
    (op code, parameter) actually applied, in order.  Draw order: randint(0, 11); random() against p1; [sharpness:
    choice([-1, 1])]; random() against p2; [sharpness: choice([-1, 1])]."""
    p1, op1, m1, p2, op2, m2 = POLICIES[rng.randint(0, len(POLICIES) - 1)]
    out = []
    for p, op, m in ((p1, op1, m1), (p2, op2, m2)):
        if rng.random() < p:
            mag = magnitude(op, m)
            if op == 'sharpness':
                mag = 1 + mag * rng.choice([-1, 1])
            out.append((OP_CODE[op], mag))
    return out


def equalize_lut(hist):
    """ImageOps.equalize for one band: ``hist`` = its 256-bin histogram."""
    histo = [int(v) for v in hist if v]
    if len(histo) <= 1:
        return np.arange(256, dtype=np.uint8)
    step = (sum(histo) - histo[-1]) // 255
    if not step:
        return np.arange(256, dtype=np.uint8)
    lut, n = [], step // 2
    for i in range(256):
        lut.append(n // step)
        n += int(hist[i])
    return np.minimum(np.array(lut), 255).astype(np.uint8)  # Image.point clips table entries to 0..255 (n // step reaches 256+)


def smooth(a):
    """ImageFilter.SMOOTH (3x3 [1 1 1; 1 5 1; 1 1 1] / 13) on uint8 [H,W,3]: float32 accumulation in Pillow's order
    (0.5, then the row below, the row itself, the row above, each left to right), truncation, borders copied."""
    k = np.array([1, 1, 1, 1, 5, 1, 1, 1, 1], np.float32) / np.float32(13)
    H, W, _ = a.shape
    out = a.copy()
    if H < 3 or W < 3:
        return out
    f = a.astype(np.float32)

    def row(r0, kk):
        A = f[r0:r0 + H - 2]
        return (A[:, 0:W - 2] * kk[0] + A[:, 1:W - 1] * kk[1]) + A[:, 2:W] * kk[2]
    ss = np.full((H - 2, W - 2, 3), np.float32(0.5), np.float32)
    ss = ss + row(2, k[0:3])
    ss = ss + row(1, k[3:6])
    ss = ss + row(0, k[6:9])
    out[1:H - 1, 1:W - 1] = np.where(ss <= 0, 0, np.where(ss >= 255, 255, ss.astype(np.int32))).astype(np.uint8)
    return out


def blend(im1, im2, alpha):
    """Image.blend(im1, im2, alpha) on uint8 arrays (Blend.c): float32 arithmetic, truncation; clipped when alpha is
    outside [0, 1]."""
    if alpha == 0.0:
        return im1.copy()
    if alpha == 1.0:
        return im2.copy()
    al = np.float32(alpha)
    a, b = im1.astype(np.int32), im2.astype(np.int32)
    t = (a.astype(np.float32) + al * (b - a).astype(np.float32)).astype(np.float32)
    if 0 <= al <= 1:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def apply_op(a, code, param):
    """One operation on a uint8 [H,W,3] image."""
    if code == EQUALIZE:
        out = np.empty_like(a)
        for c in range(3):
            out[..., c] = equalize_lut(np.bincount(a[..., c].reshape(-1), minlength=256))[a[..., c]]
        return out
    if code == POSTERIZE:
        mask = ~(2 ** (8 - int(param)) - 1)
        return (a.astype(np.int32) & mask).astype(np.uint8)
    if code == SOLARIZE:
        i = np.arange(256)
        lut = np.where(i < param, i, 255 - i).astype(np.uint8)
        return lut[a]
    if code == INVERT:
        return (255 - a.astype(np.int32)).astype(np.uint8)
    if code == SHARPNESS:
        return blend(smooth(a), a, param)
    raise ValueError(code)


def autoaug(a, ops):
    """The AutoAugment view of one uint8 [H,W,3] crop for drawn operations ``ops`` (draw_policy)."""
    for code, param in ops:
        a = apply_op(a, code, param)
    return a


def autoaug_pil(a, ops):
    """The same view through Pillow itself, the way the reference computes it (SubPolicy.func, advaug.py:82-95) - what
    bench.py's cpu_baseline times for the input pipeline (the numpy restatement above is for checking, not for speed)."""
    from PIL import Image, ImageOps, ImageEnhance
    im = Image.fromarray(a)
    for code, param in ops:
        if code == EQUALIZE:
            im = ImageOps.equalize(im)
        elif code == POSTERIZE:
            im = ImageOps.posterize(im, int(param))
        elif code == SOLARIZE:
            im = ImageOps.solarize(im, param)
        elif code == INVERT:
            im = ImageOps.invert(im)
        elif code == SHARPNESS:
            im = ImageEnhance.Sharpness(im).enhance(param)
    return np.array(im)
