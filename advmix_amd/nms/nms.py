"""Mirror of lib/nms/nms.py on the HIP library: ``nms``, ``cpu_nms_wrapper``,
``gpu_nms_wrapper``, ``py_nms_wrapper``, ``oks_nms``, ``soft_oks_nms`` with the reference's
argument meaning and return values (lists / arrays of indices into the input).

* ``gpu_nms`` = gpu_nms.pyx:19-34 around ``advmix_nms_host`` (the `_nms` ABI): fp32 IoU,
  strict ``>`` against the fp32 threshold, 64-wide bitmask + host greedy pass.
* ``nms`` (numpy semantics, nms.py:35-72: keep ``ovr <= thresh``, compared in fp32) and
  ``cpu_nms`` (cpu_nms.pyx:68: suppress ``ovr >= thresh`` compared in double) reuse the same
  device bitmask; ``cpu_nms`` expresses its predicate as a strict fp32 ``>`` against the
  largest fp32 below the threshold, which is exact.
* OKS variants take the float64 similarity matrix from ``advmix_oks_iou`` and run the (tiny, inherently
  sequential) greedy / rescoring loop on the device too: one workgroup, only the kept indices cross PCIe."""
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


def _dev64(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def _oks(gk, ga, dk, da, sigmas, in_vis_thre):
    """ious[ng, nd] (fp64, on the GPU) of ng persons against nd detections: advmix_oks_iou."""
    sig = _SIGMAS if not isinstance(sigmas, np.ndarray) else sigmas
    K = dk.shape[1] // 3
    g, a, d, b, s = _dev64(gk), _dev64(ga), _dev64(dk), _dev64(da), _dev64(sig)
    out = torch.empty((g.shape[0], d.shape[0]), dtype=torch.float64, device='cuda')
    P = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731
    call('advmix_oks_iou', P(g), P(a), g.shape[0], P(d), P(b), d.shape[0], P(s), K,
         0 if in_vis_thre is None else 1, 0.0 if in_vis_thre is None else float(in_vis_thre), P(out),
         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return out


def oks_iou(g, d, a_g, a_d, sigmas=None, in_vis_thre=None):
    """nms.py:75-94: OKS of one person ``g`` [3K] (area a_g) with each of the detections ``d`` [n, 3K] (areas a_d) ->
    float64 [n]; with ``in_vis_thre`` only the joints the DETECTION shows above it count (nms.py:90-92)."""
    d = np.asarray(d, dtype=np.float64)
    if d.shape[0] == 0:
        return np.zeros(0)
    g = np.asarray(g, dtype=np.float64).reshape(1, -1)
    return _oks(g, np.asarray([a_g], dtype=np.float64), d, np.asarray(a_d, dtype=np.float64), sigmas,
                in_vis_thre)[0].cpu().numpy()


def rescore(overlap, scores, thresh, type='gaussian'):
    """nms.py:127-136: soft-NMS rescoring.  'linear' scales the scores whose overlap reaches ``thresh`` by
    (1 - overlap) IN PLACE (as the reference does); anything else is the Gaussian penalty, returned as a new array."""
    assert overlap.shape[0] == scores.shape[0]
    if type == 'linear':
        hit = np.where(overlap >= thresh)[0]
        scores[hit] = scores[hit] * (1 - overlap[hit])
        return scores
    return scores * np.exp(-overlap ** 2 / thresh)


def _oks_matrix(kpts, areas, sigmas, device=False, in_vis_thre=None):
    """Row i = oks_iou(kpts[i], kpts, areas[i], areas, sigmas, in_vis_thre): every pair the greedy passes may ask for."""
    out = _oks(kpts, areas, kpts, areas, sigmas, in_vis_thre)
    return out if device else out.cpu().numpy()


def _unpack(kpts_db):
    scores = np.array([kpts_db[i]['score'] for i in range(len(kpts_db))])
    kpts = np.array([np.asarray(kpts_db[i]['keypoints']).flatten() for i in range(len(kpts_db))])
    areas = np.array([kpts_db[i]['area'] for i in range(len(kpts_db))])
    return scores, kpts, areas


def oks_nms(kpts_db, thresh, sigmas=None, in_vis_thre=None):
    """nms.py:97-125 (the caller, coco.py:356-364, never passes in_vis_thre; it is served all the same)."""
    if len(kpts_db) == 0:
        return []
    scores, kpts, areas = _unpack(kpts_db)
    n = len(scores)
    M = _oks_matrix(kpts, areas, sigmas, device=True, in_vis_thre=in_vis_thre)      # [n, n] fp64, stays on the GPU
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
    """nms.py:139-177: gaussian rescoring, re-sort every round, max_dets = 20.  Matrix, rescoring and re-sorting all run
    on the device (advmix_soft_oks_greedy: one workgroup, <= 20 rounds); only the kept indices come back."""
    if len(kpts_db) == 0:
        return []
    scores, kpts, areas = _unpack(kpts_db)
    n = len(scores)
    max_dets = 20
    M = _oks_matrix(kpts, areas, sigmas, device=True, in_vis_thre=in_vis_thre)      # [n, n] fp64, stays on the GPU
    order = scores.argsort()[::-1]                          # numpy's initial order (ties: its own), as in the reference
    if n > 8192 or thresh == 0 or thresh != thresh:
        # what advmix_soft_oks_greedy refuses (one workgroup holds the candidates; exp(-oks^2 / 0)): the reference's own loop
        # (nms.py:157-175) on the host over the device's OKS matrix - any n, any threshold, numpy's own arithmetic
        Mh = M.cpu().numpy()
        sc, keep = scores[order], []
        while order.size > 0 and len(keep) < max_dets:
            i = order[0]
            ovr = Mh[i, order[1:]]
            order = order[1:]
            sc = rescore(ovr, sc[1:], thresh)
            tmp = sc.argsort()[::-1]
            order, sc = order[tmp], sc[tmp]
            keep.append(i)
        return np.asarray(keep, dtype=np.intp)
    od = torch.from_numpy(np.ascontiguousarray(order, dtype=np.int32)).cuda()
    sc = _dev64(scores[order])
    ws_s = torch.empty(2 * n, dtype=torch.float64, device='cuda')
    ws_o = torch.empty(2 * n, dtype=torch.int32, device='cuda')
    out = torch.zeros(1 + max_dets, dtype=torch.int32, device='cuda')
    P = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731
    call('advmix_soft_oks_greedy', P(M), P(od), P(sc), n, float(thresh), max_dets, P(ws_s), P(ws_o),
         ctypes.c_void_p(out.data_ptr() + 4), P(out), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    out = out.cpu().numpy()
    return out[1:1 + int(out[0])].astype(np.intp)
