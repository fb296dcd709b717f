"""NMS oracle (test infra only): ctypes wrappers over nms_ref.c + numpy OKS NMS.

box NMS:  lib/nms/nms.py:35-72 (numpy), cpu_nms.pyx:20-71, gpu_nms.pyx:19-34 +
          nms_kernel.cu.  OKS: lib/nms/nms.py:75-177, float64 throughout.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def build():
    so = os.path.join(_HERE, 'liboracle_nms.so')
    src = os.path.join(_HERE, 'nms_ref.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, 'liboracle_nms.so'])
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def py_nms(dets, thresh):
    """nms.py:35-72: vectorised greedy; suppress iff ovr > thresh; dtype follows dets."""
    if dets.shape[0] == 0:
        return []
    x1, y1, x2, y2, sc = (dets[:, i] for i in range(5))
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = sc.argsort()[::-1]
    keep = []
    while order.size > 0:
        i, rest = order[0], order[1:]
        keep.append(int(i))
        w = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        order = rest[ovr <= thresh]
    return keep


def gpu_nms(dets, thresh, return_mask=False):
    """gpu_nms.pyx:19-34 around the restated device kernel + host greedy pass."""
    dets = np.ascontiguousarray(dets, np.float32)
    n = dets.shape[0]
    if n == 0:
        return []
    order = dets[:, 4].argsort()[::-1].astype(np.int32)
    sd = np.ascontiguousarray(dets[order])
    keep = np.zeros(n, np.int32)
    num = ctypes.c_int(0)
    cb = (n + 63) // 64
    mask = np.zeros(n * cb, np.uint64)
    rc = lib().oracle_gpu_nms(keep.ctypes.data_as(ctypes.c_void_p), ctypes.byref(num),
                              sd.ctypes.data_as(ctypes.c_void_p), n, ctypes.c_float(thresh),
                              mask.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    out = [int(i) for i in order[keep[:num.value]]]
    return (out, mask.reshape(n, cb)) if return_mask else out


def cpu_nms(dets, thresh):
    """cpu_nms.pyx:20-71."""
    dets = np.ascontiguousarray(dets, np.float32)
    n = dets.shape[0]
    if n == 0:
        return []
    order = np.ascontiguousarray(dets[:, 4].argsort()[::-1].astype(np.int32))
    keep = np.zeros(n, np.int32)
    num = ctypes.c_int(0)
    rc = lib().oracle_cpu_nms(keep.ctypes.data_as(ctypes.c_void_p), ctypes.byref(num),
                              dets.ctypes.data_as(ctypes.c_void_p),
                              order.ctypes.data_as(ctypes.c_void_p), n, ctypes.c_double(thresh))
    assert rc == 0
    return [int(i) for i in keep[:num.value]]


COCO_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62,
                        1.07, 1.07, .87, .87, .89, .89]) / 10.0          # nms.py:77


def oks_iou(g, d, a_g, a_d, sigmas=None, in_vis_thre=None):
    """nms.py:75-94. g:[51], d:[n,51].  With ``in_vis_thre`` only some joints count (nms.py:90-92): the reference
    writes ``ind = list(vg > t) and list(vd > t)`` - Python's ``and`` of two non-empty lists is the SECOND list, so the
    boolean mask is the detection's visibilities alone (``vd > t``); the ground-truth side never enters."""
    sig = COCO_SIGMAS if sigmas is None else sigmas
    var = (sig * 2) ** 2
    out = np.zeros(d.shape[0])
    for n in range(d.shape[0]):
        dx = d[n, 0::3] - g[0::3]
        dy = d[n, 1::3] - g[1::3]
        e = (dx ** 2 + dy ** 2) / var / ((a_g + a_d[n]) / 2 + np.spacing(1)) / 2
        if in_vis_thre is not None and g[2::3].shape[0] > 0:
            e = e[d[n, 2::3] > in_vis_thre]
        out[n] = np.sum(np.exp(-e)) / e.shape[0] if e.shape[0] else 0.0
    return out


def _unpack(db):
    sc = np.array([e['score'] for e in db])
    kp = np.array([np.asarray(e['keypoints']).flatten() for e in db])
    ar = np.array([e['area'] for e in db])
    return sc, kp, ar


def oks_nms(db, thresh, sigmas=None, in_vis_thre=None):
    """nms.py:97-125."""
    if len(db) == 0:
        return []
    sc, kp, ar = _unpack(db)
    order = sc.argsort()[::-1]
    keep = []
    while order.size > 0:
        i, rest = order[0], order[1:]
        keep.append(int(i))
        ov = oks_iou(kp[i], kp[rest], ar[i], ar[rest], sigmas, in_vis_thre)
        order = rest[ov <= thresh]
    return keep


def soft_oks_nms(db, thresh, sigmas=None, in_vis_thre=None, max_dets=20):
    """nms.py:139-177 (gaussian rescoring, re-sort every round, max_dets = 20)."""
    if len(db) == 0:
        return []
    sc, kp, ar = _unpack(db)
    order = sc.argsort()[::-1]
    sc = sc[order]
    keep = []
    while order.size > 0 and len(keep) < max_dets:
        i = order[0]
        ov = oks_iou(kp[i], kp[order[1:]], ar[i], ar[order[1:]], sigmas, in_vis_thre)
        order = order[1:]
        sc = sc[1:] * np.exp(-ov ** 2 / thresh)
        t = sc.argsort()[::-1]
        order, sc = order[t], sc[t]
        keep.append(int(i))
    return keep
