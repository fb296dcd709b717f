"""Mirror of lib/nms/nms.py on the HIP library: ``nms``, ``cpu_nms_wrapper``,
``gpu_nms_wrapper``, ``py_nms_wrapper``, ``oks_nms``, ``soft_oks_nms`` with the reference's
argument meaning and return values (lists / arrays of indices into the input).

* ``gpu_nms`` = gpu_nms.pyx:19-34 around ``advmix_nms_host`` (the `_nms` ABI): fp32 IoU,
  strict ``>`` against the fp32 threshold, 64-wide bitmask + host greedy pass.
* ``nms`` (numpy semantics, nms.py:35-72: keep ``ovr <= thresh``, compared in fp32) and
  ``cpu_nms`` (cpu_nms.pyx:68: suppress ``ovr >= thresh`` compared in double) reuse the same
  device bitmask; ``cpu_nms`` expresses its predicate as a strict fp32 ``>`` against the
  largest fp32 below the threshold, which is exact.
* OKS variants take the float64 similarity matrix from ``advmix_oks_matrix`` and run the
  (tiny, inherently sequential) greedy / rescoring loop on the host."""
import ctypes

import numpy as np
import torch

from .._lib import call


def _greedy_from_host_nms(dets, thresh):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = dets.shape[0]
    if n == 0:
        return []
    order = dets[:, 4].argsort()[::-1].astype(np.int32)
    sorted_dets = np.ascontiguousarray(dets[order, :])
    keep = np.zeros(n, dtype=np.int32)
    num_out = ctypes.c_int(0)
    dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
    call('advmix_nms_host', keep.ctypes.data_as(ctypes.c_void_p), ctypes.byref(num_out),
         sorted_dets.ctypes.data_as(ctypes.c_void_p), n, dets.shape[1], ctypes.c_float(thresh), dev)
    return list(order[keep[:num_out.value]])


def gpu_nms(dets, thresh, device_id=0):
    return _greedy_from_host_nms(dets, float(np.float32(thresh)))


def _below(x):
    return float(np.nextafter(np.float32(x), np.float32(-np.inf)))


def cpu_nms(dets, thresh):
    """Suppress iff fp32 ovr >= (double) thresh  ==  ovr > t' with t' the largest fp32 < thresh
    when thresh is exactly representable, else the fp32 value just below/at thresh."""
    t32 = np.float32(thresh)
    t = _below(t32) if float(t32) >= float(thresh) else float(t32)
    return [int(i) for i in _greedy_from_host_nms(dets, t)]


def nms(dets, thresh):
    """numpy semantics (nms.py:35-72) on fp32 dets: numpy compares the fp32 ``ovr`` array with
    the Python-float threshold in fp32 (the scalar is cast to the array dtype), so
    "keep ovr <= thresh" is exactly the device predicate with t = float32(thresh)."""
    dets = np.asarray(dets)
    if dets.shape[0] == 0:
        return []
    return [int(i) for i in _greedy_from_host_nms(dets, float(np.float32(thresh)))]


def py_nms_wrapper(thresh):
    def _nms(dets):
        return nms(dets, thresh)
    return _nms


def cpu_nms_wrapper(thresh):
    def _nms(dets):
        return cpu_nms(dets, thresh)
    return _nms


def gpu_nms_wrapper(thresh, device_id):
    def _nms(dets):
        return gpu_nms(dets, thresh, device_id)
    return _nms


_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0


def _oks_matrix(kpts, areas, sigmas, device=False):
    n, K = kpts.shape[0], kpts.shape[1] // 3
    sig = _SIGMAS if not isinstance(sigmas, np.ndarray) else sigmas
    k = torch.from_numpy(np.ascontiguousarray(kpts, dtype=np.float64)).cuda()
    a = torch.from_numpy(np.ascontiguousarray(areas, dtype=np.float64)).cuda()
    s = torch.from_numpy(np.ascontiguousarray(sig, dtype=np.float64)).cuda()
    out = torch.empty((n, n), dtype=torch.float64, device='cuda')
    P = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731
    call('advmix_oks_matrix', P(k), P(a), P(s), n, K, P(out),
         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return out if device else out.cpu().numpy()


def _unpack(kpts_db):
    scores = np.array([kpts_db[i]['score'] for i in range(len(kpts_db))])
    kpts = np.array([np.asarray(kpts_db[i]['keypoints']).flatten() for i in range(len(kpts_db))])
    areas = np.array([kpts_db[i]['area'] for i in range(len(kpts_db))])
    return scores, kpts, areas


def oks_nms(kpts_db, thresh, sigmas=None, in_vis_thre=None):
    """nms.py:97-125 (in_vis_thre is never passed by the caller, coco.py:356-364)."""
    if len(kpts_db) == 0:
        return []
    if in_vis_thre is not None:
        raise NotImplementedError('in_vis_thre is unused by the reference caller')
    scores, kpts, areas = _unpack(kpts_db)
    n = len(scores)
    M = _oks_matrix(kpts, areas, sigmas, device=True)      # [n, n] fp64, stays on the GPU
    # candidates best-first: numpy's argsort (its order among equal scores is part of the reference's result); the greedy
    # pass itself runs on the device (advmix_oks_greedy) and only the kept indices come back
    order = torch.from_numpy(np.ascontiguousarray(scores.argsort()[::-1], dtype=np.int32)).cuda()
    keep = torch.empty(n, dtype=torch.int32, device='cuda')
    cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
    P = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731
    call('advmix_oks_greedy', P(M), P(order), n, float(thresh), P(keep), P(cnt),
         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    out = torch.cat([cnt, keep]).cpu().numpy()
    return [int(i) for i in out[1:1 + int(out[0])]]


def soft_oks_nms(kpts_db, thresh, sigmas=None, in_vis_thre=None):
    """nms.py:139-177: gaussian rescoring, re-sort every round, max_dets = 20."""
    if len(kpts_db) == 0:
        return []
    if in_vis_thre is not None:
        raise NotImplementedError('in_vis_thre is unused by the reference caller')
    scores, kpts, areas = _unpack(kpts_db)
    M = _oks_matrix(kpts, areas, sigmas)
    order = scores.argsort()[::-1]
    scores = scores[order]
    max_dets = 20
    keep = np.zeros(max_dets, dtype=np.intp)
    keep_cnt = 0
    while order.size > 0 and keep_cnt < max_dets:
        i = order[0]
        oks_ovr = M[i, order[1:]]
        order = order[1:]
        scores = scores[1:] * np.exp(-oks_ovr ** 2 / thresh)
        tmp = scores.argsort()[::-1]
        order = order[tmp]
        scores = scores[tmp]
        keep[keep_cnt] = i
        keep_cnt += 1
    return keep[:keep_cnt]
