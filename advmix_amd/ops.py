"""Autograd layer over the C ABI (include/advmix_hip.h).

PyTorch is plumbing here: device memory (caching allocator), HIP streams and the autograd
tape.  All arithmetic runs in libadvmix_hip.so.

Tensors are logical NCHW with channels_last strides, i.e. dense NHWC in HBM.  Parameter
gradients are ACCUMULATED by the kernels straight into ``param.grad`` (a view of the
optimizer's flat gradient buffer once ``FlatAdam`` owns the model); autograd sees None for
them.  ``needs_input_grad`` carries the reference's three gradient modes (full / input-only
after ``set_require_grad(model, False)`` / none), lib/core/function.py:98-104,140,158.

Every op is a *member* (a pair of plain functions ``fwd``/``bwd`` taking a raw HIP stream);
``GroupFn`` is the single autograd.Function.  A group runs its members CONCURRENTLY on forked
HIP streams and joins before returning: HRNet's 2-4 resolution branches, its fuse convs and
the residual downsample paths are independent, and individually the low-resolution ones only
launch 96-192 workgroups on a 256-CU chip.  All memory is allocated on the caller's stream
(allocation is host-side), the side streams only carry kernels between the fork and the join,
so the pattern is safe for the caching allocator and legal inside HIP-graph capture.
"""
import ctypes

import torch

from ._lib import call, lib

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
_CL = torch.channels_last
MAX_LANES = int(__import__('os').environ.get('ADVMIX_LANES', '4'))   # concurrent HIP streams per launch group


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_KEEP = []      # temporaries created while a group's lanes are in flight; dropped after the join


def keep(t):
    """A tensor freed before the lanes join could be handed by the caching allocator (which only
    orders reuse on the CALLER's stream) to the next member while a side-lane kernel still reads
    it - so every temporary lives until GroupFn has joined."""
    _KEEP.append(t)
    return t


def nhwc(x):
    """Dense NHWC view of a logical-NCHW fp32 CUDA tensor (copies only if needed)."""
    if x.dtype != torch.float32 or not x.is_cuda:
        raise TypeError('advmix_amd ops need fp32 CUDA tensors (no CPU fallback), got %s on %s'
                        % (x.dtype, x.device))
    if x.dim() != 4:
        raise ValueError('expected a 4-D NCHW tensor')
    if not x.is_contiguous(memory_format=_CL):
        x = keep(x.contiguous(memory_format=_CL))
    return x


def empty_nhwc(B, C, H, W, device):
    return torch.empty((B, H, W, C), device=device, dtype=torch.float32).permute(0, 3, 1, 2)


def _grad_buf(p, st):
    """param.grad, created on demand.  The zero-fill runs on the MEMBER's stream ``st``: a torch
    zeros_() would be enqueued on the caller's stream and race with a side lane's accumulation."""
    if p.grad is None:
        flat = getattr(p, '_flat_grad_view', None)         # owned by FlatAdam: something set the grads to None
        if flat is not None:                               # (nn.Module.zero_grad) - re-attach the flat view instead of
            p.grad = flat                                  # a standalone tensor the optimizer would never read
            return flat
        g = torch.empty_like(p, memory_format=torch.preserve_format)
        call('advmix_fill', _p(g), 0.0, g.numel(), st)
        p.grad = g
    return p.grad


_ws_cache = {}
WS_BYTES = 48 << 20      # fixed per-lane scratch: norm partials need <= 512*2*C*8 B (C = 2048: 16.8 MB)
STAT_SLOTS = 64          # capacity reserved per channel and statistic; a launch uses 16 of them (conv_direct.hip; ADVMIX_STAT_SLOTS)
STAT_SLOTS_ASK = int(__import__('os').environ.get('ADVMIX_STAT_SLOTS', '0'))   # 0 = the kernel's choice


# Side channels between members of DIFFERENT launch groups (the autograd graph only carries tensors):
#  _BNB_FWD: output tensor (data_ptr) of a train-mode ConvBN whose only consumer is a fuse sum -> what that sum's backward
#            needs to leave the BatchNorm-backward channel sums in the layer's slots (ConvBN.fwd -> FuseSum.fwd);
#  _BNB_PRE: (arena id, slot offset) -> (arena, pass id, slot count): "the gradient of THIS layer's output already carries
#            act' and its two channel sums are in your backward slots" (FuseSum.bwd -> ConvBN.bwd).  Keyed by the LAYER, not
#            by the gradient tensor's address (round 4): a layer registered here has the fuse sum as its ONLY consumer
#            (plan.PlanNet._fuse_only), so what reaches its backward is that one gradient - stolen, or copied by autograd on the
#            way (across a data-parallel cut it is: four hand-offs per pass used to be lost to the address check), never summed
#            with another; the pass id keeps a second backward through the same forward out.
_BNB_FWD, _BNB_PRE = {}, {}
_WARNED = {}


class StatArena:
    """The fp64 statistics slots of ONE network: for every conv -> train-mode BatchNorm pair a forward set
    (column sum / sum of squares of the conv output, written by the conv's epilogue) and a backward set (sum g,
    sum g * xhat, written by the epilogue of the input-gradient conv that produces g), each [2][STAT_SLOTS][C] (slot-major since round 4).
    The consumers (norm_apply_slots / norm_bwd_apply_slots) reduce the slots themselves - no finalize launch - and
    nobody re-zeroes them in a kernel: the whole arena is zero-filled ONCE at the start of every training forward
    (``begin_pass``, on the caller's stream before any lane forks; ~10 MB for HRNet-W32).  ``pass_id`` / ``dirty``
    guard the backward sets: a set may be accumulated into once per pass; any other pattern (two backwards through
    one forward, a backward from an older forward) takes the unfused path, which needs no slots."""

    def __init__(self):
        self.t, self.size, self.pass_id, self.dirty = None, 0, 0, set()

    def reserve(self, channels):
        off = self.size
        self.size += 2 * channels * STAT_SLOTS
        return off

    def begin_pass(self, device):
        if self.size == 0:
            return
        if self.t is None or self.t.device != device:
            self.t = torch.empty(self.size, device=device, dtype=torch.float64)
        call('advmix_fill', ctypes.c_void_p(self.t.data_ptr()), 0.0, 2 * self.size, _st())
        self.pass_id += 1
        self.dirty.clear()
        for reg in (_BNB_FWD, _BNB_PRE):                   # hand-offs of an earlier pass that nobody picked up
            stale = [k for k, v in reg.items() if v[0] is self]
            for k in stale:
                del reg[k]
            if stale and reg is _BNB_PRE and not _WARNED.get('bnb_pre'):
                # FuseSum.bwd left a layer's BatchNorm-backward sums in its slots and claimed them, but that layer's backward
                # never saw the gradient tensor it was keyed by (autograd copied or summed it on the way): the layer took the
                # unfused path - correct, but fusion was lost silently
                _WARNED['bnb_pre'] = True
                import warnings
                warnings.warn('advmix_amd: %d fuse-layer BatchNorm-backward hand-off(s) were not consumed in the last pass; '
                              'those layers ran the unfused backward' % len(stale))

    def ptr(self, off):
        return ctypes.c_void_p(self.t.data_ptr() + 8 * off)

    def claim_bwd(self, off, pass_id):
        """True (once) if the backward set at ``off`` is still clean for the pass the forward belonged to."""
        if self.t is None or pass_id != self.pass_id or off in self.dirty:
            return False
        self.dirty.add(off)
        return True


def _workspace(device, nbytes, lane):
    """Per-(device, lane) scratch of fixed size, allocated once (never re-allocated while another
    lane might still be using the old one).  Lanes never share scratch; within a lane kernels
    are stream-ordered."""
    if nbytes > WS_BYTES:
        raise ValueError('norm workspace of %d bytes exceeds the per-lane scratch (%d)' % (nbytes, WS_BYTES))
    key = (device.index, _LANE_SET[0], lane)
    w = _ws_cache.get(key)
    if w is None:
        w = torch.zeros(WS_BYTES // 4, device=device, dtype=torch.float32)   # ticket words start at 0
        _ws_cache[key] = w
    return w


def _ensure_workspaces(device, n):
    """Create (and zero) the scratch of lanes 0..n-1 on the CALLER's stream, before any fork."""
    for lane in range(n):
        _workspace(device, 0, lane)


_lane_streams = {}
_LANE_SET = [0]         # which set of launch-lane streams and per-lane scratch launch groups use right now (see lane_set)


def _lanes(device, n):
    ss = _lane_streams.setdefault((device.index, _LANE_SET[0]), [])
    while len(ss) < n:
        ss.append(torch.cuda.Stream(device=device))
    return ss[:n]


class lane_set:
    """``with ops.lane_set(k):`` - launch groups issued inside fork onto lane-stream set k and use its per-lane scratch
    (set 0 is the default).  For a network that runs on its OWN stream beside another one (round 6: the frozen teacher's
    forward beside the generator's and the student's, core.function.advmix_phase_a): sharing the lane streams would chain the
    two networks' launch groups behind each other, sharing the scratch would race."""

    def __init__(self, k):
        self.k = int(k)

    def __enter__(self):
        self.prev = _LANE_SET[0]
        _LANE_SET[0] = self.k

    def __exit__(self, *exc):
        _LANE_SET[0] = self.prev


_aux_streams = {}


def aux_stream(device, k=1):
    """A stream of its own for lane set k's network (created once per device)."""
    key = (device.index, k)
    if key not in _aux_streams:
        _aux_streams[key] = torch.cuda.Stream(device=device)
    return _aux_streams[key]


_DIRECT = {'ok': __import__('os').environ.get('ADVMIX_BT', '1') != '0'}


def _direct_ok():
    """False while a test forces the first-generation conv (advmix_set_option('direct', 0))."""
    return _DIRECT['ok']


def set_option(name, value):
    """advmix_set_option plus the Python-side dispatch it implies."""
    call('advmix_set_option', name.encode(), int(value))
    if name == 'direct':
        _DIRECT['ok'] = bool(value)


def _check_w(w):
    if not w.is_contiguous(memory_format=_CL):
        raise ValueError('conv weights must be channels_last ([O][R][S][I] in memory)')


def _wt(st, w, A, T, B):
    """[A][T][B] -> [B][T][A] weight re-layout for the transposed-gather kernel."""
    out = keep(torch.empty(w.numel(), device=w.device, dtype=torch.float32))
    call('advmix_transpose_w', _p(w), _p(out), A, T, B, st)
    return out


# =============================================================================================
# members: fwd(st, lane, tensors, meta, needs) -> (outputs, saved, extra)
#          bwd(st, lane, saved, extra, meta, grads, needs) -> input grads (aligned with tensors)
# =============================================================================================
BNB_FUSED = __import__('os').environ.get('ADVMIX_BNB', '1') != '0'
FUSE_BNB = __import__('os').environ.get('ADVMIX_FUSE_BNB', '1') != '0'
ACT_MASK = __import__('os').environ.get('ADVMIX_ACT_MASK', '1') != '0'       # 0: residual layers take the unfused BatchNorm backward (A/B)    # the fuse layers' BatchNorm-backward sums from FuseSum.bwd
DETERMINISTIC = False
REPLAY_ON_NULL = __import__('os').environ.get('ADVMIX_REPLAY_STREAM', 'own') == 'null'


def set_deterministic(on=True):
    """Bit-reproducible mode: every accumulation whose ORDER varies run to run takes its ordered
    form - weight / bias gradients and the loss sum store per-slice partials and add them in slice order
    (advmix_conv_wgrad_det, advmix_bias_grad_det, advmix_joints_loss_det) instead of fp32 atomics; convolutions never
    split K across the grid; the conv epilogues STORE one BatchNorm partial per row tile instead of adding fp64 atomics
    and advmix_stats_fold adds them in a fixed order (round 3: the epilogue fusion stays; rounds 1-2 fell back to the
    block-partial kernels and an extra pass over the tensor).  The fuse layers' sums (FuseSum.bwd) and shapes the
    epilogues do not serve take the block-partial kernels.  Slower (a fold launch per BatchNorm, two launches per weight
    gradient); same results to rounding."""
    global DETERMINISTIC
    DETERMINISTIC = bool(on)
    call('advmix_set_option', b'deterministic', 1 if on else 0)


if __import__('os').environ.get('ADVMIX_DETERMINISTIC', '0') == '1':
    set_deterministic(True)


# Weight gradients have no consumer before the optimizer step (lib/core/function.py:154-155): inside a launch chain they are
# not launched where autograd reaches them but collected, and at the end of the chain's backward the ones of ONE geometry - the
# eight 3x3 C -> C convs of an HRNet branch's four residual blocks - go out as a single launch (advmix_conv_wgrad_group: an
# eighth of the pixel slices per problem to merge with atomics, longer main loops, the 128 x 128 tile).  Round 4: a knock-out
# had shown the weight gradients cost the step 9.4 of 55.5 ms, i.e. they are NOT hidden behind the other lanes.
DECONV_GEMM = __import__('os').environ.get('ADVMIX_DECONV_GEMM', '1') != '0'   # narrow ConvTranspose2d forward: GEMM + gather (A/B switch)
WGRAD_GROUP = __import__('os').environ.get('ADVMIX_WGRAD_GROUP', '1') != '0'
_WG_DEFER = []          # a stack of pending lists: [(a, b, grad, geom)] of the Chain.bwd calls in progress


def _wgrad(st, lane, a, b, w, geom, park=False, v=None, xbn=None):
    """Weight gradient accumulated into w.grad (atomics, or ordered partials in deterministic mode).  ``park``: a and b stay
    as they are until the autograd pass ends (ConvBN.bwd: a fresh dc, a saved activation), so a SMALL problem may wait for
    a mixed launch (_wgrad_single); the U-Net's plain convs hand over gradients that later backward ops update in place.
    ``xbn`` (round 6): b is the RAW output of the preceding conv and (mean, invstd, gamma, beta, record) its BatchNorm - the
    forward applied it on load (ConvBN.fwd ``in_bn``); the Winograd weight gradient does the same while it stages b, every
    other kernel gets the activation materialised (ConvBN.materialize)."""
    g = _grad_buf(w, st)
    if xbn is not None and (DETERMINISTIC or geom[7:] != (3, 3, 1, 1)):
        b, xbn = keep(ConvBN.materialize(xbn[4], st)), None   # (no Winograd weight gradient will take it)
    if DETERMINISTIC:
        ws = _workspace(a.device, 0, lane)
        call('advmix_conv_wgrad_det', _p(a), _p(b), _p(g), *geom, _p(ws), WS_BYTES, st)
    elif geom[7:] == (4, 4, 2, 1) and _wgrad4x4s2_wino(st, a, b, g, geom, v):     # (the U-Net's convs: csrc/conv_wino4.hip; ``v``: the
        pass                                                                     #  input transform of b a preceding launch left)
    elif _WG_DEFER and WGRAD_GROUP and ((geom[3] % 64 == 0 and geom[6] % 4 == 0)
                                        or (geom[3] == 32 and geom[6] == 32 and geom[7:] == (3, 3, 1, 1))):
        _WG_DEFER[-1].append((a, b, g, geom, park, xbn))    # (a - a kept temporary - and b stay alive in the pending list)
    elif not _wgrad_wino(st, [(a, b, g, xbn)], geom):       # (a lone 3x3 with enough work, e.g. transition1's 256 -> 32 @64x48)
        if xbn is not None:
            b = keep(ConvBN.materialize(xbn[4], st))
        _wgrad_single(st, a, b, g, geom, park)


# The SMALL weight gradients - the strided 3x3 and 1x1 convs of HRNet's fuse layers and transitions, ~70 per backward pass at
# 7-23 us of launch latency each - wait in one list across launch groups and go out up to 16 at a time as ONE launch of
# mixed geometries (advmix_conv_wgrad_multi): at the start of the next multi-lane group's backward on its last lane (beside
# that group's members), the rest from a callback at the end of the autograd pass - before anything can read a gradient.
WGRAD_MULTI = __import__('os').environ.get('ADVMIX_WGRAD_MULTI', '1') != '0'
WGRAD_MULTI_MAX_FLOP = float(__import__('os').environ.get('ADVMIX_WGRAD_MULTI_MAX_GFLOP', '1.0')) * 1e9
WGRAD_MULTI_FLUSH = int(__import__('os').environ.get('ADVMIX_WGRAD_MULTI_FLUSH', '8'))    # pending problems that trigger a flush
_WG_SMALL = {}          # {autograd graph task id: [(a, b, grad, geom)]} across the launch groups of THAT pass


_graph_task_id = getattr(torch._C, '_current_graph_task_id', None)
if _graph_task_id is None:                                   # (a torch without the accessor: one list, as before round 6 - the
    def _graph_task_id():                                    #  forward-side clean-up then never fires, parking still works)
        return 0


def _pass_key():
    """The autograd pass in progress (-1: none).  ADVICE r5 (medium): the parked problems used to live in one process-global
    list, so what a pass that RAISED had parked (its end-of-pass callback never runs) went out with the next pass - stale
    dc / x products accumulated into live gradients - and a nested ``autograd.grad`` could flush its outer pass's problems.
    Graph-task ids grow monotonically and a nested pass runs inside one node of its outer pass.  (Not keyed by thread: the
    engine runs a pass's CUDA nodes on its device thread, the forward that cleans up runs on the caller's.)"""
    return _graph_task_id()


def _park_list(key, create=False):
    """This pass's parked problems; on the way, what can no longer belong to a live pass is dropped: lists of a LATER task
    id (a nested pass that has returned, or raised, before its outer pass reached this point)."""
    for k in [k for k in _WG_SMALL if k > key]:
        del _WG_SMALL[k]
    if create:
        return _WG_SMALL.setdefault(key, [])
    return _WG_SMALL.get(key, [])


def drop_stale_parked():
    """Outside any autograd pass (forward side of a launch group): whatever is still held was parked by a pass that raised -
    never launched, only released.  (One process drives one GPU from one thread at a time; a forward on one thread beside
    another thread's backward pass is not a supported way to use the library.)"""
    if _WG_SMALL and _graph_task_id() < 0:
        _WG_SMALL.clear()


def _wgrad_single(st, a, b, g, geom, park):
    """One weight gradient on its own: launched now, or - small, ``park``, inside an autograd pass - parked for a mixed launch."""
    B, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad = geom
    if (park and WGRAD_MULTI and not DETERMINISTIC and Ca % 4 == 0 and Cb % 4 == 0
            and 2.0 * B * Ha * Wa * Ca * Cb * R * S <= WGRAD_MULTI_MAX_FLOP):
        key = _pass_key()
        if key >= 0:                                         # (not inside an autograd pass: nobody would flush)
            lst = _park_list(key, create=True)
            try:
                if not lst:                                  # one callback per pass, bound to THIS pass's list
                    torch.autograd.Variable._execution_engine.queue_callback(lambda: _flush_small_wgrads_at_end(key))
                lst.append((a, b, g, geom))
                return
            except RuntimeError:                             # (no pass in progress after all: launch it now)
                if not lst:
                    _WG_SMALL.pop(key, None)
    call('advmix_conv_wgrad', _p(a), _p(b), _p(g), *geom, st)


def _flush_small_wgrads(st, key, count=None, keep_alive=True):
    """The first ``count`` (default: all) weight gradients parked by pass ``key`` as mixed launches of up to 16 on stream
    ``st``; ``keep_alive``: their operands stay referenced until the group in progress has joined its lanes (keep())."""
    lst = _WG_SMALL.get(key, [])
    count = len(lst) if count is None else count
    pending = lst[:count]
    del lst[:count]
    if not lst:
        _WG_SMALL.pop(key, None)
    for i in range(0, len(pending), 16):
        grp = pending[i:i + 16]
        n = len(grp)
        arr = ctypes.c_void_p * n
        geoms = (ctypes.c_int * (11 * n))(*[v for x in grp for v in x[3]])
        rc = lib.advmix_conv_wgrad_multi(n, arr(*[x[0].data_ptr() for x in grp]), arr(*[x[1].data_ptr() for x in grp]),
                                         arr(*[x[2].data_ptr() for x in grp]), geoms, st) if n >= 2 else 1
        if rc == 0:
            COUNTERS['wgrad_multi'] = COUNTERS.get('wgrad_multi', 0) + 1
        elif rc == 1:
            for a, b, g, geom in grp:
                call('advmix_conv_wgrad', _p(a), _p(b), _p(g), *geom, st)
        else:
            raise RuntimeError('advmix_conv_wgrad_multi failed: %d' % rc)
    if keep_alive:
        _KEEP.extend(t for x in pending for t in x[:2])


def _flush_small_wgrads_at_end(key):
    """End of the autograd pass ``key`` (queue_callback): what it still has parked goes out on the caller's stream - every
    lane has joined it."""
    if _WG_SMALL.get(key):
        _flush_small_wgrads(_st(), key, keep_alive=False)
    _WG_SMALL.pop(key, None)


WGRAD_WINO = __import__('os').environ.get('ADVMIX_WGRAD_WINO', '1') != '0'   # Winograd F(3x3,2x2) weight gradients (A/B switch)
WGRAD_WINO_MIN_UNITS = int(__import__('os').environ.get('ADVMIX_WGRAD_WINO_MIN_UNITS', '2048'))   # (a lone 32 -> 32 @64x48 problem - 768 units - is no faster than wgrad3x3_c32)


def _wgrad_wino(st, grp, geom):
    """The group's weight gradients [(dy, x, grad[, xbn])] through the Winograd kernel (csrc/wgrad_wino.hip) when it serves the
    geometry and there is enough work to fill the chip; False = not taken.  ``xbn`` per problem: x is the raw output of the
    preceding conv, its BatchNorm + ReLU (saved statistics) is applied while x is staged (see _wgrad)."""
    B, Ha, Wa, Ca, Hb, Wb, Cb, R, S, stride, pad = geom       # a = dy [B,Ha,Wa,Ca], b = x [B,Hb,Wb,Cb]
    if not (WGRAD_WINO and WINO and (R, S, stride, pad) == (3, 3, 1, 1)) or DETERMINISTIC:
        return False
    if lib.advmix_wgrad_wino_config(B, Hb, Wb, Cb, Ca) * len(grp) < WGRAD_WINO_MIN_UNITS:
        return False
    n = len(grp)
    arr = ctypes.c_void_p * n
    xb = [x[3] if len(x) > 3 else None for x in grp]
    if any(v is not None for v in xb):
        col = lambda k: arr(*[(v[k].data_ptr() if v is not None else None) for v in xb])      # noqa: E731
        rc = lib.advmix_conv3x3_wgrad_wino_group_bn(n, arr(*[x[0].data_ptr() for x in grp]), arr(*[x[1].data_ptr() for x in grp]),
                                                    arr(*[x[2].data_ptr() for x in grp]), col(0), col(1), col(2), col(3),
                                                    B, Hb, Wb, Ca, Cb, st)
        if rc == 0:
            COUNTERS['wgrad_wino_bn'] = COUNTERS.get('wgrad_wino_bn', 0) + sum(1 for v in xb if v is not None)
    else:
        rc = lib.advmix_conv3x3_wgrad_wino_group(n, arr(*[x[0].data_ptr() for x in grp]), arr(*[x[1].data_ptr() for x in grp]),
                                                 arr(*[x[2].data_ptr() for x in grp]), B, Hb, Wb, Ca, Cb, st)
    if rc == 0:
        COUNTERS['wgrad_wino'] = COUNTERS.get('wgrad_wino', 0) + 1
        return True
    if rc != 1:
        raise RuntimeError('advmix_conv3x3_wgrad_wino_group failed: %d' % rc)
    return False


def _flush_wgrads(st, pending):
    """Launch the collected weight gradients: groups of 2-8 of one geometry as one launch, the rest one by one."""
    by = {}
    for a, b, g, geom, park, xbn in pending:
        by.setdefault(geom, []).append((a, b, g, xbn, park))
    for geom, items in by.items():
        for i in range(0, len(items), 8):
            grp = items[i:i + 8]
            n = len(grp)
            if _wgrad_wino(st, grp, geom):
                continue
            # no kernel below applies a BatchNorm on load: the activation as a tensor for the problems that deferred theirs
            grp = [(a, b if xbn is None else keep(ConvBN.materialize(xbn[4], st)), g, None, park) for a, b, g, xbn, park in grp]
            if n >= 2:
                arr = ctypes.c_void_p * n
                rc = lib.advmix_conv_wgrad_group(n, arr(*[x[0].data_ptr() for x in grp]), arr(*[x[1].data_ptr() for x in grp]),
                                                 arr(*[x[2].data_ptr() for x in grp]), *geom, st)
                if rc == 0:
                    COUNTERS['wgrad_group'] = COUNTERS.get('wgrad_group', 0) + 1
                    continue
                if rc != 1:
                    raise RuntimeError('advmix_conv_wgrad_group failed: %d' % rc)
            for a, b, g, _x, park in grp:
                _wgrad_single(st, a, b, g, geom, park)


def _bias_grad(st, lane, dy, bias, rows, C):
    g = _grad_buf(bias, st)
    if DETERMINISTIC:
        ws = _workspace(dy.device, 0, lane)
        call('advmix_bias_grad_det', _p(dy), _p(g), rows, C, _p(ws), WS_BYTES, st)
    else:
        call('advmix_bias_grad', _p(dy), _p(g), rows, C, st)
COUNTERS = {'bnb': 0}       # launches whose epilogue carried a BatchNorm backward (tests assert the path is taken)


def _det_stats(C, device, lane):
    """Deterministic mode: where a conv epilogue stores its per-tile partial sums (the lane's scratch; stream-ordered
    with the fold that follows) and how many tiles fit per (statistic, channel)."""
    ws = _workspace(device, 0, lane)
    return ws, WS_BYTES // 8 // (2 * C)


# ---- Winograd F(2x2,3x3) path (csrc/conv_wino.hip) ---------------------------------------------------------------------
WINO = __import__('os').environ.get('ADVMIX_WINO', '1') != '0'       # A/B switch: 0 = every conv on the direct kernels
WINO_MIN_WGS = int(__import__('os').environ.get('ADVMIX_WINO_MIN_WGS', '96'))   # workgroups (32 tiles x 32 channels) below which the direct kernel stays
WINO_ASYNC = __import__('os').environ.get('ADVMIX_WINO_ASYNC', '1') != '0'   # filter transforms beside the stem (plan.PlanNet._wino_refresh; 0 = on the caller's stream)
PW = __import__('os').environ.get('ADVMIX_PW', '1') != '0'           # A/B switch: 0 = the 64 -> 256 1x1 convs of the bottlenecks on the direct kernel (csrc/conv_pw.hip)
WINO4 = __import__('os').environ.get('ADVMIX_WINO4', '1') != '0'     # A/B switch: 0 = the U-Net's 4x4 / stride-2 convs on the direct kernel (csrc/conv_wino4.hip)
WINO4_T = __import__('os').environ.get('ADVMIX_WINO4_T', '1') != '0'   # A/B switch: 0 = the transposed form (ConvTranspose2d forward, Conv2d input gradient) on the direct kernel
WINO4_T_MIN_TILES = int(__import__('os').environ.get('ADVMIX_WINO4_T_MIN_TILES', '192'))
WINO4_KEEP_V = __import__('os').environ.get('ADVMIX_WINO4_KEEP_V', '1') != '0'   # a Conv2d's forward keeps its input transform for its weight gradient (0 = made again)
WINO4_WGRAD = __import__('os').environ.get('ADVMIX_WINO4_WGRAD', '1') != '0'   # A/B switch: 0 = their weight gradients on the direct kernel
WINO4_WGRAD_MIN_TILES = int(__import__('os').environ.get('ADVMIX_WINO4_WGRAD_MIN_TILES', '512'))   # (8 x 6 maps, 192 tiles at B = 32: the 16 dU planes - up to 134 MB - cost more than the multiplies saved)
WINO4_MIN_TILES = int(__import__('os').environ.get('ADVMIX_WINO4_MIN_TILES', '128'))   # 3x3 output tiles below which the direct kernel stays (B = 32: the 4x3 bottleneck has 64)
SMAP = __import__('os').environ.get('ADVMIX_SMAP', '1') != '0'       # A/B switch: 0 = the small 256-channel maps on the direct kernel
SMAP_C = 256                                                         # (csrc/conv_smap.hip: one workgroup per image, K split over its eight waves)
SMAP_WINO = __import__('os').environ.get('ADVMIX_SMAP_WINO', '1') != '0'   # Winograd F(2x2,3x3) inside that workgroup shape (conv_smapw; 0 = the direct form)


class WinoBank:
    """The transformed-filter images (U = G g G^T, MFMA fragment order) of a set of 3x3 / stride 1 / pad 1 conv weights: one
    side buffer, one launch to refresh all of it (``refresh``: at the start of every forward pass of the owning network, on
    the caller's stream before any lane forks - the filters change once per optimizer step and the launch is ~4 us, so
    nothing tracks versions).  Each weight TENSOR OBJECT is tagged with its two images and the device address they were
    made for (``w._wino``): a conv finds them on the tensor it was handed, and a tensor whose storage has moved since
    (``.to()``, FlatAdam's flat buffer) - or any other tensor that merely lives at a recycled address - has no valid tag and
    takes the direct kernel.  (A registry keyed by address alone served stale images to a later test's weights.)"""

    def __init__(self, weights):
        import numpy as np
        import weakref
        dev = weights[0].device
        self.key = tuple(w.data_ptr() for w in weights)
        pad32 = lambda c: (c + 31) // 32 * 32                # noqa: E731  (the n dimension is padded to whole column tiles)
        # kind 'smap' (csrc/conv_smap.hip): filters with 256 input channels, plain re-layout in that kernel's fragment order (9 Co Ci floats
        # per image); everything else 'wino'
        kinds = ['w4' if tuple(w.shape[2:]) == (4, 4) else 'pw' if tuple(w.shape[2:]) == (1, 1) else
                 ('smapw' if SMAP_WINO else 'smap') if (w.shape[1] == SMAP_C and w.shape[0] % 32 == 0) else 'wino' for w in weights]   # (input-gradient images only where Cout == 256 too)
        sizes = [(64 * w.shape[0] * w.shape[1], 64 * w.shape[0] * w.shape[1] if WINO4_T and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0 else 0) if k == 'w4' else (9 * w.shape[0] * w.shape[1],) * 2 if k == 'smap' else (16 * w.shape[0] * w.shape[1],) * 2 if k == 'smapw' else
                 (w.shape[0] * w.shape[1],) * 2 if k == 'pw' else
                 (16 * pad32(w.shape[0]) * w.shape[1], 16 * pad32(w.shape[1]) * w.shape[0]) for w, k in zip(weights, kinds)]
        self.buf = torch.empty(sum(a + b for a, b in sizes), device=dev, dtype=torch.float32)
        rec = np.dtype([('w', '<u8'), ('u', '<u8'), ('Cn', '<i4'), ('Ck', '<i4'), ('role', '<i4'), ('blk0', '<i4')])
        ents = {'wino': [], 'smap': [], 'smapw': [], 'pw': [], 'w4': [], 'w4t': []}
        blk = {'wino': 0, 'smap': 0, 'smapw': 0, 'pw': 0, 'w4': 0, 'w4t': 0}
        owner = {'wino': [], 'smap': [], 'smapw': [], 'pw': [], 'w4': [], 'w4t': []}
        off = 0
        self._tagged = []
        for i, (w, kind) in enumerate(zip(weights, kinds)):
            Co, Ci, R, S = w.shape
            if (R, S) not in ((3, 3), (1, 1), (4, 4)) or Co % 16 or Ci % 16 or not w.is_contiguous(memory_format=_CL):
                raise ValueError('WinoBank: 3x3 / 1x1 / 4x4 channels_last weights with Cout, Cin multiples of 16')
            ptrs = []
            for role, (Cn, Ck) in enumerate(((Co, Ci), (Ci, Co))):
                u = self.buf.data_ptr() + 4 * off
                off += sizes[i][role]
                if (kind in ('smap', 'smapw') and Ck != SMAP_C) or (kind == 'pw' and (Cn, Ck) != (256, 64)) or (kind == 'w4' and role == 1 and not sizes[i][1]):
                    # (the small-map kernels read exactly 256 channels, conv_pw reads 64 and writes 256: no such image)
                    ptrs.append(None)
                    continue
                tk = kind
                if kind == 'w4':                            # filters [Co][4][4][Ci] in memory (a Conv2d's own, a ConvTranspose2d's [Cin][4][4][Cout]): role 0 = the
                    Cn, Ck = Co, Ci                         # forward-form image (csrc/conv_wino4.hip), role 1 = the transposed form's, a table and a launch of its own
                    tk = 'w4t' if role == 1 else 'w4'
                nb = {'wino': ((Cn + 31) // 32) * (Ck // 8), 'smap': (Cn // 32) * (Ck // 32) * 36, 'smapw': (Cn // 32) * 32, 'pw': 64,
                      'w4': Cn * 4 * Ck // 256, 'w4t': (Cn // 32) * (Ck // 32) * 4}[tk]
                owner[tk] += [len(ents[tk])] * nb
                ents[tk].append((w.data_ptr(), u, Cn, Ck, role, blk[tk]))
                blk[tk] = len(owner[tk])
                ptrs.append(ctypes.c_void_p(u))
            w._wino = (self.buf, ptrs[0], ptrs[1], w.data_ptr(), kind)     # (the tag keeps the side buffer alive)
            self._tagged.append(weakref.ref(w))
        self.tables = {}
        for kind in ('wino', 'smap', 'smapw', 'pw', 'w4', 'w4t'):
            if ents[kind]:
                ent = np.array(ents[kind], dtype=rec)
                self.tables[kind] = (torch.from_numpy(ent.view(np.uint8).copy()).to(dev),
                                     torch.tensor(owner[kind], dtype=torch.int32).to(dev), blk[kind])

    def matches(self, weights):
        return self.key == tuple(w.data_ptr() for w in weights) and all(_wino_tag(w) is not None for w in weights)

    def refresh(self, st=None):
        st = st if st is not None else _st()
        for kind, (ents, blk_ent, blocks) in self.tables.items():
            call('advmix_%s_weights' % kind, _p(ents), _p(blk_ent), blocks, st)

    def images(self, w):
        """(forward image, input-gradient image) pointers of one of the bank's weights."""
        tag = _wino_tag(w)
        if tag is None or tag[0] is not self.buf:
            raise KeyError('not a weight of this bank (or its storage has moved)')
        return tag[1], tag[2]

    def release(self):
        for r in self._tagged:
            w = r()
            if w is not None and getattr(w, '_wino', (None,))[0] is self.buf:
                del w._wino
        self._tagged = []


def _wino_tag(w):
    tag = getattr(w, '_wino', None)
    return tag if tag is not None and tag[3] == w.data_ptr() else None


_W3 = {'wino': (lib.advmix_conv3x3_wino_fwd, lib.advmix_conv3x3_wino_dgrad),      # (forward, input gradient) entry points per image kind
       'smap': (lib.advmix_conv3x3_smap_fwd, lib.advmix_conv3x3_smap_dgrad),
       'smapw': (lib.advmix_conv3x3_smapw_fwd, lib.advmix_conv3x3_smapw_dgrad),
       'pw': (lib.advmix_conv1x1_pw_fwd, lib.advmix_conv1x1_pw_dgrad)}


def _wino_images(w, B, H, W, Ci, Co, R, S, stride, pad):
    """(forward image, input-gradient image) of ``w`` when this problem goes to the Winograd kernel, else None."""
    if not WINO or DETERMINISTIC or not _direct_ok() or stride != 1:
        return None
    tag = _wino_tag(w)
    if tag is None:
        return None
    if (R, S, pad) == (1, 1, 0):                            # 64 -> 256 over many pixels: the streaming kernel (csrc/conv_pw.hip)
        if tag[4] != 'pw' or not PW or lib.advmix_conv_pw_config(B, H, W, Ci, Co) < WINO_MIN_WGS:
            return None
        return tag[1], tag[2], 'pw'
    if (R, S, pad) != (3, 3, 1) or tag[4] == 'pw':
        return None
    if tag[4] != 'wino':                                    # 256 -> 256 on a map of <= 48 pixels: the image-per-workgroup kernels
        config = lib.advmix_conv_smapw_config if tag[4] == 'smapw' else lib.advmix_conv_smap_config
        if not SMAP or config(B, H, W, Ci, Co) < WINO_MIN_WGS:
            return None
        return tag[1], tag[2], tag[4]
    if lib.advmix_conv_wino_config(B, H, W, Ci, Co) < WINO_MIN_WGS:
        return None
    return tag[1], tag[2], 'wino'


def _conv4x4s2_wino(st, x, w, bias, y, B, Hi, Wi, Ci, Co):
    """The forward-form 4x4 / stride 2 / pad 1 conv x[B,Hi,Wi,Ci] -> y[B,Hi/2,Wi/2,Co] with filters w (memory [Co][4][4][Ci]) on
    csrc/conv_wino4.hip when ``w`` carries a 'w4' image and the shape is served; None = nothing launched, else the launch's
    scratch tensor - the input transform of x sits at its start (the weight gradient of a transposed conv reads it again)."""
    if not (WINO and WINO4) or DETERMINISTIC or not _direct_ok():
        return None
    tag = _wino_tag(w)
    if tag is None or tag[4] != 'w4' or B * (-(-Hi // 6)) * (-(-Wi // 6)) < WINO4_MIN_TILES:
        return None
    wsf = lib.advmix_conv4x4s2_wino_ws_floats(B, Hi, Wi, Ci, Co)
    if wsf <= 0:
        return None
    ws = keep(torch.empty(wsf, device=x.device, dtype=torch.float32))
    rc = lib.advmix_conv4x4s2_wino_fwd(_p(x), tag[1], _p(bias), _p(y), _p(ws), wsf, B, Hi, Wi, Ci, Co, st)
    if rc == 0:
        COUNTERS['w4'] = COUNTERS.get('w4', 0) + 1
        return ws
    if rc != 1:
        raise RuntimeError('advmix_conv4x4s2_wino_fwd failed: %d' % rc)
    return None


def _deconv4x4s2_wino(st, x, w, bias, add_to, y, B, Hl, Wl, Cl, Ch):
    """The transposed form, x[B,Hl,Wl,Cl] -> y[B,2Hl,2Wl,Ch] (+ bias + add_to) with filters w (memory [Cl][4][4][Ch]), on
    csrc/conv_wino4.hip when ``w`` carries a transposed-form image and the shape is served; False = nothing launched."""
    if not (WINO and WINO4 and WINO4_T) or DETERMINISTIC or not _direct_ok():
        return False
    tag = _wino_tag(w)
    if tag is None or tag[4] != 'w4' or tag[2] is None or B * (Hl // 3 + 1) * (Wl // 3 + 1) < WINO4_T_MIN_TILES:
        return False
    wsf = lib.advmix_deconv4x4s2_wino_ws_floats(B, Hl, Wl, Cl, Ch)
    if wsf <= 0:
        return False
    ws = keep(torch.empty(wsf, device=x.device, dtype=torch.float32))
    rc = lib.advmix_deconv4x4s2_wino_fwd(_p(x), tag[2], _p(bias), _p(add_to), _p(y), _p(ws), wsf, B, Hl, Wl, Cl, Ch, st)
    if rc == 0:
        COUNTERS['w4t'] = COUNTERS.get('w4t', 0) + 1
        return True
    if rc != 1:
        raise RuntimeError('advmix_deconv4x4s2_wino_fwd failed: %d' % rc)
    return False


def _wgrad4x4s2_wino(st, lo, hi, g, geom, v=None):
    """Weight gradient of a 4x4 / stride 2 / pad 1 conv (filters [Cl][4][4][Ch]) on csrc/conv_wino4.hip, accumulated into ``g``:
    lo [B,Ha,Wa,Cl] (the conv's output gradient / a transposed conv's input), hi [B,2Ha,2Wa,Ch].  ``v``: the scratch tensor of a
    preceding _conv4x4s2_wino(hi, ...) on the same stream (its input transform is reused).  False = nothing launched."""
    B, Ha, Wa, Cl, Hb, Wb, Ch = geom[:7]
    if not (WINO and WINO4 and WINO4_WGRAD) or DETERMINISTIC or not _direct_ok() or B * (-(-Hb // 6)) * (-(-Wb // 6)) < WINO4_WGRAD_MIN_TILES:
        return False
    wsf = lib.advmix_conv4x4s2_wino_wgrad_ws_floats(B, Hb, Wb, Ch, Cl, 1 if v is not None else 0)
    if wsf <= 0:
        return False
    ws = keep(torch.empty(wsf, device=hi.device, dtype=torch.float32))
    rc = lib.advmix_conv4x4s2_wino_wgrad(_p(hi), _p(lo), _p(g), _p(v), _p(ws), wsf, B, Hb, Wb, Ch, Cl, st)
    if rc == 0:
        COUNTERS['w4_wgrad'] = COUNTERS.get('w4_wgrad', 0) + 1
        return True
    if rc != 1:
        raise RuntimeError('advmix_conv4x4s2_wino_wgrad failed: %d' % rc)
    return False


def _conv_dgrad(st, dy, w, x, B, Ho, Wo, Co, Hi, Wi, Ci, R, S, stride, pad, add_to=None, bnb=None, lane=0):
    """Input gradient of a conv (+ ``add_to``, another gradient of the same input: summed in the kernel's
    epilogue on the conv_direct path, by advmix_add otherwise).

    ``bnb`` (dict): the conv's input is y = act(BN(c) + res) of a train-mode ConvBN earlier in the same chain and
    this is the LAST contribution to dL/dy.  Where the kernel serves the shape, the epilogue also multiplies by
    act'(y) - the sign of y from the producer's bit mask (``bnb['mask']``) or, without a residual, recomputed from c -
    and accumulates the two BatchNorm-backward channel sums into the producer's slots; ``bnb['done']`` is
    then set to the slot count and the result is g = dL/dy * act'(y) instead of dL/dy."""
    dx = empty_nhwc(B, Ci, Hi, Wi, x.device)
    if Co % 16 == 0 and Ci % 4 == 0 and _direct_ok():          # weights consumed in their own layout
        if add_to is not None and (add_to.shape != dx.shape or add_to.stride() != dx.stride()):
            raise RuntimeError('advmix_amd: gradient fan-in of differently laid out tensors')
        if (R, S, stride, pad) == (4, 4, 2, 1) and bnb is None and (Hi, Wi) == (2 * Ho, 2 * Wo) and \
                _deconv4x4s2_wino(st, dy, w, None, add_to, dx, B, Ho, Wo, Co, Ci):       # (the U-Net's down convs: csrc/conv_wino4.hip)
            return dx
        wu = _wino_images(w, B, Hi, Wi, Co, Ci, R, S, stride, pad)      # (the gradient conv reads Co channels, writes Ci)
        if bnb is not None and BNB_FUSED:
            if tuple(bnb['c'].shape) != tuple(dx.shape) or bnb['c'].stride() != dx.stride():
                raise RuntimeError('advmix_amd: BatchNorm-backward epilogue on a differently laid out tensor')
            if wu is not None:
                nsv = ctypes.c_int(STAT_SLOTS_ASK)
                rc = _W3[wu[2]][1](_p(dy), wu[1], _p(add_to), _p(dx), B, Hi, Wi, Co, Ci, _p(bnb['mask']),
                                                   _p(bnb['c']), _p(bnb['mean']), _p(bnb['invstd']), _p(bnb['gamma']),
                                                   _p(bnb['beta']), bnb['act'], bnb['slots'], ctypes.byref(nsv), st)
                if rc == 0:
                    bnb['done'] = nsv.value
                    COUNTERS['bnb'] += 1
                    COUNTERS[wu[2][:4]] = COUNTERS.get(wu[2][:4], 0) + 1     # ('wino' / 'smap' / 'pw': both small-map kernels count as smap)
                    return dx
                if rc != 1:
                    raise RuntimeError('advmix_conv3x3_wino_dgrad failed: %d' % rc)
            if DETERMINISTIC:                               # per-tile partials (plain stores) + an ordered fold: no atomics
                ws, cap = _det_stats(Ci, x.device, lane)
                nsv, target = ctypes.c_int(-cap), _p(ws)
            else:
                nsv, target = ctypes.c_int(STAT_SLOTS_ASK), bnb['slots']
            rc = lib.advmix_conv_tr_w_bnb(_p(dy), _p(w), _p(add_to), _p(dx), B, Ho, Wo, Co, Hi, Wi, Ci, R, S, stride,
                                          pad, _p(bnb['mask']), _p(bnb['c']), _p(bnb['mean']), _p(bnb['invstd']),
                                          _p(bnb['gamma']), _p(bnb['beta']), bnb['act'], target,
                                          ctypes.byref(nsv), st)
            if rc == 0:
                if DETERMINISTIC:
                    call('advmix_stats_fold', target, nsv.value, Ci, bnb['slots'], st)
                    nsv.value = 1
                bnb['done'] = nsv.value
                COUNTERS['bnb'] += 1
                return dx
            if rc != 1:
                raise RuntimeError('advmix_conv_tr_w_bnb failed: %d' % rc)
        if wu is not None:
            rc = _W3[wu[2]][1](_p(dy), wu[1], _p(add_to), _p(dx), B, Hi, Wi, Co, Ci, None, None, None, None,
                                               None, None, 0, None, None, st)
            if rc == 0:
                COUNTERS[wu[2][:4]] = COUNTERS.get(wu[2][:4], 0) + 1     # ('wino' / 'smap' / 'pw': both small-map kernels count as smap)
                return dx
            if rc != 1:
                raise RuntimeError('advmix_conv3x3_wino_dgrad failed: %d' % rc)
        rc = lib.advmix_conv_tr_w_add(_p(dy), _p(w), _p(add_to), _p(dx), B, Ho, Wo, Co, Hi, Wi, Ci, R, S,
                                      stride, pad, st)
        if rc == 0:
            return dx
        if rc != 1:                                        # 1 = EINVAL: not served (a tensor of 2 GiB or more) -> first-generation kernel
            raise RuntimeError('advmix_conv_tr_w_add failed: %d' % rc)
    if Ci <= 4:                                               # a network's first conv: 3 image channels
        rc = lib.advmix_conv_tr_narrow(_p(dy), _p(w), _p(dx), B, Ho, Wo, Co, Hi, Wi, Ci, R, S, stride, pad, st)
        if rc == 0:
            return dx if add_to is None else _add(st, keep(dx), add_to)
        if rc != 1:
            raise RuntimeError('advmix_conv_tr_narrow failed: %d' % rc)
    wt = _wt(st, w, Co, R * S, Ci)
    call('advmix_conv_tr', _p(dy), _p(wt), None, _p(dx), B, Ho, Wo, Co, Hi, Wi, Ci, R, S, stride, pad, st)
    return dx if add_to is None else _add(st, keep(dx), add_to)


class Conv:
    NHWC = (0,)      # inputs made dense NHWC BEFORE the lanes fork (None = all)
    """nn.Conv2d (square stride / padding, dilation 1, groups 1).  tensors = (x, w, bias|None)."""

    @staticmethod
    def fwd(st, lane, t, meta, needs):
        x, w, bias = t
        stride, pad = meta
        x = nhwc(x)
        _check_w(w)
        B, Ci, Hi, Wi = x.shape
        Co, _, R, S = w.shape
        Ho = (Hi + 2 * pad - R) // stride + 1
        Wo = (Wi + 2 * pad - S) // stride + 1
        y = empty_nhwc(B, Co, Ho, Wo, x.device)
        v = _conv4x4s2_wino(st, x, w, bias, y, B, Hi, Wi, Ci, Co) if (R, S, stride, pad) == (4, 4, 2, 1) else None
        if v is None:
            call('advmix_conv_fwd', _p(x), _p(w), _p(bias), _p(y), B, Hi, Wi, Ci, Ho, Wo, Co, R, S,
                 stride, pad, st)
        # (the Winograd launch's scratch starts with the input transform of x: the weight gradient multiplies the same one)
        return (y,), (x, w, bias), (v if needs[1] and WINO4_KEEP_V else None)

    ADD_TO = True    # bwd(..., add_to): another gradient of the input, summed in the dgrad epilogue

    @staticmethod
    def bwd(st, lane, saved, extra, meta, grads, needs, add_to=None, pre=0, bnb=None):
        x, w, bias = saved
        stride, pad = meta
        dy = nhwc(grads[0])
        B, Ci, Hi, Wi = x.shape
        Co, _, R, S = w.shape
        Ho, Wo = dy.shape[2], dy.shape[3]
        dx = None
        if needs[0]:
            dx = _conv_dgrad(st, dy, w, x, B, Ho, Wo, Co, Hi, Wi, Ci, R, S, stride, pad, add_to, bnb, lane)
        if needs[1]:
            _wgrad(st, lane, dy, x, w, (B, Ho, Wo, Co, Hi, Wi, Ci, R, S, stride, pad), v=extra if torch.is_tensor(extra) else None)
        if bias is not None and needs[2]:
            _bias_grad(st, lane, dy, bias, B * Ho * Wo, Co)
        return dx, None, None


class Deconv:
    NHWC = (0,)      # inputs made dense NHWC BEFORE the lanes fork (None = all)
    """nn.ConvTranspose2d (output_padding 0); weight logical [Cin, Cout, R, S]."""

    @staticmethod
    def fwd(st, lane, t, meta, needs):
        x, w, bias = t
        stride, pad = meta
        x = nhwc(x)
        _check_w(w)
        B, Ci, Hi, Wi = x.shape
        _, Co, R, S = w.shape
        Ho = (Hi - 1) * stride - 2 * pad + R
        Wo = (Wi - 1) * stride - 2 * pad + S
        y = empty_nhwc(B, Co, Ho, Wo, x.device)
        if Co <= 4 and R == 4 and S == 4 and stride == 2 and pad == 1 and Ci % 8 == 0 and _direct_ok():   # the U-Net's tail
            if DECONV_GEMM and Ci % 16 == 0:               # products of every input pixel on the matrix pipe + a gather
                ws = keep(torch.empty(B * Hi * Wi * 16 * Co, device=x.device, dtype=torch.float32))
                rc = lib.advmix_deconv4x4s2_narrow_gemm(_p(x), _p(w), _p(bias), _p(y), _p(ws), ws.numel() * 4, B, Hi, Wi, Ci, Co, st)
                if rc == 0:
                    return (y,), (x, w, bias), None
                if rc != 1:
                    raise RuntimeError('advmix_deconv4x4s2_narrow_gemm failed: %d' % rc)
            call('advmix_deconv4x4s2_narrow', _p(x), _p(w), _p(bias), _p(y), B, Hi, Wi, Ci, Co, st)
            return (y,), (x, w, bias), None
        if (R, S, stride, pad) == (4, 4, 2, 1) and _deconv4x4s2_wino(st, x, w, bias, None, y, B, Hi, Wi, Ci, Co):
            return (y,), (x, w, bias), None
        if Ci % 16 == 0 and Co % 4 == 0 and _direct_ok():
            rc = lib.advmix_conv_tr_w(_p(x), _p(w), _p(bias), _p(y), B, Hi, Wi, Ci, Ho, Wo, Co, R, S,
                                      stride, pad, st)
        else:
            rc = 1
        if rc == 1:                                        # not served by the second-generation kernel
            wt = _wt(st, w, Ci, R * S, Co)                # [Co][R][S][Ci]
            call('advmix_conv_tr', _p(x), _p(wt), _p(bias), _p(y), B, Hi, Wi, Ci, Ho, Wo, Co, R, S,
                 stride, pad, st)
        elif rc != 0:
            raise RuntimeError('advmix_conv_tr_w failed: %d' % rc)
        return (y,), (x, w, bias), None

    @staticmethod
    def bwd(st, lane, saved, extra, meta, grads, needs):
        x, w, bias = saved
        stride, pad = meta
        dy = nhwc(grads[0])
        B, Ci, Hi, Wi = x.shape
        _, Co, R, S = w.shape
        Ho, Wo = dy.shape[2], dy.shape[3]
        dx, v = None, None
        if needs[0]:
            dx = empty_nhwc(B, Ci, Hi, Wi, x.device)
            v = _conv4x4s2_wino(st, dy, w, None, dx, B, Ho, Wo, Co, Ci) if (R, S, stride, pad) == (4, 4, 2, 1) else None
            if v is None:
                call('advmix_conv_fwd', _p(dy), _p(w), None, _p(dx), B, Ho, Wo, Co, Hi, Wi, Ci, R, S,
                     stride, pad, st)
        if needs[1]:
            _wgrad(st, lane, x, dy, w, (B, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride, pad), v=v)
        if bias is not None and needs[2]:
            _bias_grad(st, lane, dy, bias, B * Ho * Wo, Co)
        return dx, None, None


class BatchNorm:
    NHWC = (0, 6)      # inputs made dense NHWC BEFORE the lanes fork (None = all)
    """y = act(BN(x) + residual).  tensors = (x, gamma, beta, rmean, rvar, nbt, residual|None);
    meta = (act, training, momentum, eps).  training: batch stats + running-stat update."""

    @staticmethod
    def fwd(st, lane, t, meta, needs):
        x, gamma, beta, rmean, rvar, nbt, residual = t
        act, training, momentum, eps = meta
        x = nhwc(x)
        B, C, H, W = x.shape
        rows = B * H * W
        res = nhwc(residual) if residual is not None else None
        y = empty_nhwc(B, C, H, W, x.device)
        if not training:
            call('advmix_bn_eval', _p(x), _p(gamma), _p(beta), _p(rmean), _p(rvar), eps, _p(res),
                 _p(y), rows, C, act, st)
            return (y,), (), None
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        ws = _workspace(x.device, lib.advmix_norm_ws_bytes(1, C), lane)
        call('advmix_norm_stats', _p(x), 1, rows, C, eps, _p(mean), _p(invstd), _p(rmean), _p(rvar),
             _p(nbt), momentum, _p(ws), st)
        call('advmix_norm_apply', _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(res), _p(y),
             C, 1, rows, C, act, st)
        return (y,), (x, y, mean, invstd, gamma, beta), residual is not None

    @staticmethod
    def bwd(st, lane, saved, has_res, meta, grads, needs):
        act, training = meta[0], meta[1]
        if not training:
            raise RuntimeError('advmix_amd: backward through eval-mode BatchNorm is not on the hot path')
        x, y, mean, invstd, gamma, beta = saved
        dy = nhwc(grads[0])
        B, C, H, W = x.shape
        rows = B * H * W
        need_res = has_res and needs[6]
        dx = empty_nhwc(B, C, H, W, x.device)
        dres = None
        if need_res:
            dres = dy if act == ACT_NONE else empty_nhwc(B, C, H, W, x.device)
        ws = _workspace(x.device, lib.advmix_norm_ws_bytes(1, C), lane)
        dg = _grad_buf(gamma, st) if needs[1] else None
        db = _grad_buf(beta, st) if needs[2] else None
        call('advmix_norm_bwd', _p(dy), _p(y), C, _p(x), _p(mean), _p(invstd), _p(gamma), _p(dx),
             _p(dres) if (need_res and act != ACT_NONE) else None, _p(dg), _p(db), 1, rows, C, act,
             _p(ws), st)
        return dx, None, None, None, None, None, dres


class ConvBN:
    """conv (bias-free) -> BatchNorm -> (+residual) -> activation as ONE member (pose_hrnet.py:41-57).
    tensors = (x, w, gamma, beta, rmean, rvar, nbt, residual|None); meta = (stride, pad, act, training, momentum,
    eps[, arena, fwd_off, bwd_off]) - the network's StatArena and this layer's two slot sets.
    eval: a single launch (BN folded into the conv epilogue).  training: TWO launches - the conv, whose epilogue
    leaves the column sums in the layer's slots, and norm_apply_slots, which reduces the slots itself, publishes
    mean / invstd, updates the running statistics and applies BN + residual + activation (the one-wave-per-channel
    finalize launch in between is gone).  Shapes the fused epilogue does not serve fall back to the separate kernels.
    backward: when the gradient arrives already multiplied by act'(y) with its two channel sums in the layer's
    backward slots (``pre`` = slot count; produced by the epilogue of the input-gradient conv of the consumer, see
    Chain.bwd), ONE launch (norm_bwd_apply_slots) replaces the statistics pass + finalize + apply."""
    NHWC = (0, 7)

    @staticmethod
    def materialize(pend, st):
        """The activation act(BN(c)) of a DEFERRED train-mode ConvBN (see ``fwd``) as a tensor after all: by norm_apply_slots
        when nobody has derived the batch statistics yet (it publishes them), by norm_apply from the published ones otherwise.
        Cached in the record: the producer's and the consumer's backward may both ask."""
        if pend.get('y') is not None:
            return pend['y']
        c = pend['c']
        B, Co, Ho, Wo = c.shape
        y = empty_nhwc(B, Co, Ho, Wo, c.device)
        rows = B * Ho * Wo
        if not pend['published']:
            call('advmix_norm_apply_slots', _p(c), pend['slots'], pend['ns'], rows, Co, pend['eps'], _p(pend['gamma']),
                 _p(pend['beta']), None, _p(y), pend['act'], _p(pend['mean']), _p(pend['invstd']), _p(pend['rmean']),
                 _p(pend['rvar']), _p(pend['nbt']), pend['momentum'], None, st)
            pend['published'] = True
        else:
            call('advmix_norm_apply', _p(c), _p(pend['mean']), _p(pend['invstd']), _p(pend['gamma']), _p(pend['beta']), None,
                 _p(y), Co, 1, rows, Co, pend['act'], st)
        pend['y'] = y
        COUNTERS['inbn_materialized'] = COUNTERS.get('inbn_materialized', 0) + 1
        return y

    @staticmethod
    def fwd(st, lane, t, meta, needs, defer=False, in_bn=None):
        """``defer`` (round 6; ops.Chain decides): a train-mode layer whose activation has ONE reader - the 3x3 Winograd conv
        that follows it in the same chain - stops after the conv and its column sums: the output handed on is the RAW conv
        output c, and ``extra[6]`` is the record ('pend') of everything norm_apply_slots would have needed.  ``in_bn``: such a
        record of THIS layer's input; the conv applies that BatchNorm + ReLU while it stages its input
        (advmix_conv3x3_wino_fwd_inbn: +0.3 ... +1.3 us on the conv instead of a 6 ... 10 us launch and a tensor), derives and
        publishes the statistics on the way; where the kernel refuses, the activation is materialised first."""
        x, w, gamma, beta, rmean, rvar, nbt, residual = t
        stride, pad, act, training, momentum, eps = meta[:6]
        arena, fwd_off, bwd_off = meta[6:9] if len(meta) > 6 else (None, 0, 0)
        x = nhwc(x)
        _check_w(w)
        res = nhwc(residual) if residual is not None else None
        B, Ci, Hi, Wi = x.shape
        Co, _, R, S = w.shape
        Ho = (Hi + 2 * pad - R) // stride + 1
        Wo = (Wi + 2 * pad - S) // stride + 1
        rows = B * Ho * Wo
        geom = (B, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride, pad)
        fused_ok = _direct_ok() and Ci % 16 == 0
        wu = _wino_images(w, B, Hi, Wi, Ci, Co, R, S, stride, pad) if fused_ok else None
        xbn = None                                          # (mean, invstd, gamma, beta) of the BatchNorm this conv applied on load
        if in_bn is not None and not (training and wu is not None and wu[2] == 'wino' and not DETERMINISTIC
                                      and arena is not None and arena.t is not None):
            x = keep(ConvBN.materialize(in_bn, st))         # (not the fused kernel's case after all)
            in_bn = None
        y = empty_nhwc(B, Co, Ho, Wo, x.device) if not defer else None
        if not training:
            rc = 1
            if y is None:
                y = empty_nhwc(B, Co, Ho, Wo, x.device)
            if wu is not None:                              # Winograd F(2x2,3x3): same fused epilogue, 2.25x fewer MFMAs
                rc = _W3[wu[2]][0](_p(x), wu[0], _p(y), B, Hi, Wi, Ci, Co, _p(gamma), _p(beta), _p(rmean),
                                                 _p(rvar), eps, _p(res), act, None, None, st)
                if rc == 0:
                    COUNTERS[wu[2][:4]] = COUNTERS.get(wu[2][:4], 0) + 1     # ('wino' / 'smap' / 'pw': both small-map kernels count as smap)
            if rc == 1:
                rc = lib.advmix_conv_fwd_ex(_p(x), _p(w), None, _p(y), *geom, _p(gamma), _p(beta), _p(rmean),
                                            _p(rvar), eps, _p(res), act, None, None, st) if fused_ok else 1
            if rc == 1:                                     # not served by the fused epilogue
                c = keep(empty_nhwc(B, Co, Ho, Wo, x.device))
                call('advmix_conv_fwd', _p(x), _p(w), None, _p(c), *geom, st)
                call('advmix_bn_eval', _p(c), _p(gamma), _p(beta), _p(rmean), _p(rvar), eps, _p(res),
                     _p(y), rows, Co, act, st)
            elif rc != 0:
                raise RuntimeError('advmix_conv_fwd_ex failed: %d' % rc)
            return (y,), (), None
        c = empty_nhwc(B, Co, Ho, Wo, x.device)
        mean = torch.empty(Co, device=x.device, dtype=torch.float32)
        invstd = torch.empty(Co, device=x.device, dtype=torch.float32)
        rc = 1
        mask, slot_fwd = None, False       # activation bit mask for the backward epilogue; "y came from norm_apply_slots"
        if fused_ok:
            if arena is not None and arena.t is not None:
                slots = arena.ptr(fwd_off)
            else:                                           # functional use outside a network: private slots,
                tmp = keep(torch.empty(2 * Co * STAT_SLOTS, device=x.device, dtype=torch.float64))
                call('advmix_fill', _p(tmp), 0.0, 2 * tmp.numel(), st)     # zeroed on THIS member's stream
                slots = _p(tmp)
            if DETERMINISTIC:                               # no fp64 atomics: the epilogue stores one partial per row tile
                ws, cap = _det_stats(Co, x.device, lane)    # in the lane's scratch, an ordered fold leaves the totals in
                nbg, target = ctypes.c_int(-cap), _p(ws)    # the layer's slots (ns = 1) - no extra pass over c
            else:
                nbg, target = ctypes.c_int(STAT_SLOTS_ASK), slots
            if in_bn is not None:
                rc = lib.advmix_conv3x3_wino_fwd_inbn(
                    _p(x), wu[0], _p(c), B, Hi, Wi, Ci, Co, in_bn['slots'], in_bn['ns'], _p(in_bn['gamma']), _p(in_bn['beta']),
                    in_bn['eps'], _p(in_bn['mean']), _p(in_bn['invstd']), _p(in_bn['rmean']), _p(in_bn['rvar']), _p(in_bn['nbt']),
                    in_bn['momentum'], target, ctypes.byref(nbg), st)
                if rc == 0:
                    in_bn['published'] = True
                    xbn = (in_bn['mean'], in_bn['invstd'], in_bn['gamma'], in_bn['beta'], in_bn)
                    COUNTERS['wino'] = COUNTERS.get('wino', 0) + 1
                    COUNTERS['inbn'] = COUNTERS.get('inbn', 0) + 1
                elif rc == 1:                               # refused (nothing launched): the activation as a tensor, then as ever
                    x = keep(ConvBN.materialize(in_bn, st))
                    in_bn = None
                else:
                    raise RuntimeError('advmix_conv3x3_wino_fwd_inbn failed: %d' % rc)
            if rc == 1 and wu is not None and not DETERMINISTIC:
                rc = _W3[wu[2]][0](_p(x), wu[0], _p(c), B, Hi, Wi, Ci, Co, None, None, None, None, 0.0, None, 0,
                                                 target, ctypes.byref(nbg), st)
                if rc == 0:
                    COUNTERS[wu[2][:4]] = COUNTERS.get(wu[2][:4], 0) + 1     # ('wino' / 'smap' / 'pw': both small-map kernels count as smap)
            if rc == 1:
                rc = lib.advmix_conv_fwd_ex(_p(x), _p(w), None, _p(c), *geom, None, None, None, None, 0.0, None, 0,
                                            target, ctypes.byref(nbg), st)
            if rc == 0 and DETERMINISTIC:
                call('advmix_stats_fold', target, nbg.value, Co, slots, st)
                nbg.value = 1
            if rc == 0 and defer and res is None and not DETERMINISTIC and nbg.value <= 16 and arena is not None and arena.t is not None:
                # deferred: the reader applies BatchNorm + activation itself (or materialises); everything it needs, in one record
                pend = {'c': c, 'slots': slots, 'ns': nbg.value, 'gamma': gamma, 'beta': beta, 'eps': eps, 'mean': mean,
                        'invstd': invstd, 'rmean': rmean, 'rvar': rvar, 'nbt': nbt, 'momentum': momentum, 'act': act,
                        'published': False, 'y': None}
                extra = (False, arena, arena.pass_id, bwd_off, None, True, pend, xbn)     # (slot_fwd: the sign of y can be recomputed from c)
                return (c,), (x, w, c, None, mean, invstd, gamma, beta), extra
            if y is None:
                y = empty_nhwc(B, Co, Ho, Wo, x.device)
            if rc == 0:
                if res is not None and act != ACT_NONE and Co % 16 == 0 and any(needs) and BNB_FUSED and ACT_MASK:
                    # y = act(BN(c) + residual): the consumer's BatchNorm-backward epilogue needs the SIGN of y only - a bit
                    # per element here (1/16 of y's bytes) instead of reading the fp32 tensor again (without a residual the
                    # sign is recomputed from c: nothing is written)
                    mask = torch.empty(rows * Co // 4, device=x.device, dtype=torch.uint8)
                rc2 = lib.advmix_norm_apply_slots(_p(c), slots, nbg.value, rows, Co, eps, _p(gamma), _p(beta), _p(res),
                                                  _p(y), act, _p(mean), _p(invstd), _p(rmean), _p(rvar), _p(nbt),
                                                  momentum, _p(mask), st)
                slot_fwd = rc2 == 0
                if rc2 == 1:                                # e.g. Co % 4 != 0: separate finalize + apply
                    mask = None
                    call('advmix_norm_finalize', slots, nbg.value, rows, Co, eps, _p(mean), _p(invstd), _p(rmean),
                         _p(rvar), _p(nbt), momentum, st)
                    call('advmix_norm_apply', _p(c), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(res), _p(y),
                         Co, 1, rows, Co, act, st)
                elif rc2 != 0:
                    raise RuntimeError('advmix_norm_apply_slots failed: %d' % rc2)
            elif rc != 1:
                raise RuntimeError('advmix_conv_fwd_ex failed: %d' % rc)
        if rc == 1:
            ws = _workspace(x.device, lib.advmix_norm_ws_bytes(1, Co), lane)
            call('advmix_conv_fwd', _p(x), _p(w), None, _p(c), *geom, st)
            call('advmix_norm_stats', _p(c), 1, rows, Co, eps, _p(mean), _p(invstd), _p(rmean), _p(rvar),
                 _p(nbt), momentum, _p(ws), st)
            call('advmix_norm_apply', _p(c), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(res), _p(y),
                 Co, 1, rows, Co, act, st)
        extra = (residual is not None, arena, arena.pass_id if arena is not None else 0, bwd_off, mask, slot_fwd, None, xbn)
        if len(meta) > 9 and meta[9] and arena is not None and arena.t is not None and act == ACT_NONE and any(needs) \
                and BNB_FUSED and FUSE_BNB and not DETERMINISTIC:
            # the only consumer is a fuse sum in another launch group: its backward can produce this layer's sums
            _BNB_FWD[y.data_ptr()] = (arena, arena.pass_id, bwd_off, c, mean, invstd)
        return (y,), (x, w, c, y, mean, invstd, gamma, beta), extra

    ADD_TO = True

    @staticmethod
    def bnb_target(saved, extra, meta):
        """What the input-gradient conv of this layer's CONSUMER needs to fold this layer's BatchNorm-backward
        statistics into its epilogue (None if the layer's backward slots cannot be used for this pass)."""
        has_res, arena, pass_id, bwd_off, mask, slot_fwd = extra[:6]
        act = meta[2]
        if arena is None or act not in (ACT_NONE, ACT_RELU):
            return None
        if act != ACT_NONE and (mask is None if has_res else not slot_fwd):
            return None                                     # no mask was written / y did not come from norm_apply_slots
        if not arena.claim_bwd(bwd_off, pass_id):
            return None
        x, w, c, y, mean, invstd, gamma, beta = saved
        return {'mask': mask, 'c': c, 'mean': mean, 'invstd': invstd, 'gamma': gamma, 'beta': beta, 'act': act,
                'slots': arena.ptr(bwd_off)}

    @staticmethod
    def bwd(st, lane, saved, extra, meta, grads, needs, add_to=None, pre=0, bnb=None):
        stride, pad, act, training = meta[0], meta[1], meta[2], meta[3]
        if not training:
            raise RuntimeError('advmix_amd: backward through eval-mode BatchNorm is not on the hot path')
        has_res, arena, _pass, bwd_off = extra[:4]
        pend = extra[6] if len(extra) > 6 else None         # this layer's activation was deferred to its reader (see fwd)
        xbn = extra[7] if len(extra) > 7 else None          # this layer's conv applied its INPUT's BatchNorm on load: x is raw
        x, w, c, y, mean, invstd, gamma, beta = saved
        dy = nhwc(grads[0])
        B, Ci, Hi, Wi = x.shape
        Co, _, R, S = w.shape
        Ho, Wo = c.shape[2], c.shape[3]
        rows = B * Ho * Wo
        need_res = has_res and needs[7]
        if not pre and arena is not None:                   # sums left by the backward of a fuse sum in another group
            hit = _BNB_PRE.pop((id(arena), bwd_off), None)
            if hit is not None and hit[0] is arena and hit[1] == _pass:
                pre = hit[2]
        dc = keep(empty_nhwc(B, Co, Ho, Wo, x.device))
        dres = None
        dg = _grad_buf(gamma, st) if needs[2] else None
        db = _grad_buf(beta, st) if needs[3] else None
        if pre:                                             # dy is g = dL/dy * act'(y); its sums are in the slots
            call('advmix_norm_bwd_apply_slots', _p(dy), _p(c), _p(mean), _p(invstd), _p(gamma), arena.ptr(bwd_off),
                 pre, rows, Co, _p(dc), _p(dg), _p(db), st)
            if need_res:
                dres = dy
        else:
            if need_res:
                dres = dy if act == ACT_NONE else empty_nhwc(B, Co, Ho, Wo, x.device)
            if y is None:                                   # deferred, and no input-gradient epilogue carried its sums: the
                y = keep(ConvBN.materialize(pend, st))      # statistics pass below wants the activation as a tensor
            ws = _workspace(x.device, lib.advmix_norm_ws_bytes(1, Co), lane)
            call('advmix_norm_bwd', _p(dy), _p(y), Co, _p(c), _p(mean), _p(invstd), _p(gamma), _p(dc),
                 _p(dres) if (need_res and act != ACT_NONE) else None, _p(dg), _p(db), 1, rows, Co, act,
                 _p(ws), st)
        dx = None
        if needs[0]:
            dx = _conv_dgrad(st, dc, w, x, B, Ho, Wo, Co, Hi, Wi, Ci, R, S, stride, pad, add_to, bnb, lane)
        if needs[1]:
            _wgrad(st, lane, dc, x, w, (B, Ho, Wo, Co, Hi, Wi, Ci, R, S, stride, pad), park=True, xbn=xbn)
        return dx, None, None, None, None, None, None, dres


class InstanceNorm:
    NHWC = (0,)      # inputs made dense NHWC BEFORE the lanes fork (None = all)
    """y = act(InstanceNorm2d(x)), affine=False, no running stats.  meta = (act, eps)."""

    @staticmethod
    def fwd(st, lane, t, meta, needs):
        x = nhwc(t[0])
        act, eps = meta
        B, C, H, W = x.shape
        mean = torch.empty(B * C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(B * C, device=x.device, dtype=torch.float32)
        ws = _workspace(x.device, lib.advmix_norm_ws_bytes(B, C), lane)
        y = empty_nhwc(B, C, H, W, x.device)
        call('advmix_norm_stats', _p(x), B, H * W, C, eps, _p(mean), _p(invstd), None, None, None,
             0.0, _p(ws), st)
        call('advmix_norm_apply', _p(x), _p(mean), _p(invstd), None, None, None, _p(y), C, B, H * W,
             C, act, st)
        return (y,), (x, y, mean, invstd), None

    @staticmethod
    def bwd(st, lane, saved, extra, meta, grads, needs):
        x, y, mean, invstd = saved
        dy = nhwc(grads[0])
        B, C, H, W = x.shape
        dx = empty_nhwc(B, C, H, W, x.device)
        ws = _workspace(x.device, lib.advmix_norm_ws_bytes(B, C), lane)
        call('advmix_norm_bwd', _p(dy), _p(y), C, _p(x), _p(mean), _p(invstd), None, _p(dx), None,
             None, None, B, H * W, C, meta[0], _p(ws), st)
        return (dx,)


class Act:
    NHWC = (0,)      # inputs made dense NHWC BEFORE the lanes fork (None = all)
    @staticmethod
    def fwd(st, lane, t, meta, needs):
        x = nhwc(t[0])
        B, C, H, W = x.shape
        y = empty_nhwc(B, C, H, W, x.device)
        call('advmix_act_copy', _p(x), C, _p(y), C, B * H * W, C, meta, st)
        return (y,), (y,), None

    @staticmethod
    def bwd(st, lane, saved, extra, meta, grads, needs):
        (y,) = saved
        dy = nhwc(grads[0])
        B, C, H, W = y.shape
        dx = empty_nhwc(B, C, H, W, y.device)
        call('advmix_act_bwd', _p(dy), C, _p(y), C, _p(dx), C, B * H * W, C, meta, st)
        return (dx,)


class CatAct:
    NHWC = (0, 1)      # inputs made dense NHWC BEFORE the lanes fork (None = all)
    """y = act(cat([a, b], dim=1))  (Unet_generator.py:83 followed by the parent's uprelu)."""

    @staticmethod
    def fwd(st, lane, t, meta, needs):
        a, b = nhwc(t[0]), nhwc(t[1])
        B, Ca, H, W = a.shape
        Cb = b.shape[1]
        C = Ca + Cb
        y = empty_nhwc(B, C, H, W, a.device)
        rows = B * H * W
        base = y.data_ptr()
        call('advmix_act_copy', _p(a), Ca, ctypes.c_void_p(base), C, rows, Ca, meta, st)
        call('advmix_act_copy', _p(b), Cb, ctypes.c_void_p(base + 4 * Ca), C, rows, Cb, meta, st)
        return (y,), (y,), (Ca, Cb)

    @staticmethod
    def bwd(st, lane, saved, extra, meta, grads, needs):
        (y,) = saved
        Ca, Cb = extra
        dy = nhwc(grads[0])
        B, C, H, W = y.shape
        rows = B * H * W
        da = empty_nhwc(B, Ca, H, W, y.device) if needs[0] else None
        db = empty_nhwc(B, Cb, H, W, y.device) if needs[1] else None
        if da is not None:
            call('advmix_act_bwd', _p(dy), C, _p(y), C, _p(da), Ca, rows, Ca, meta, st)
        if db is not None:
            call('advmix_act_bwd', ctypes.c_void_p(dy.data_ptr() + 4 * Ca), C,
                 ctypes.c_void_p(y.data_ptr() + 4 * Ca), C, _p(db), Cb, rows, Cb, meta, st)
        return da, db


class FuseSum:
    NHWC = None      # inputs made dense NHWC BEFORE the lanes fork (None = all)
    """y = act(sum_j nearest_up_{2^shift_j}(in_j))  (pose_hrnet.py:206,254-265). meta=(act, shifts).
    Backward: ONE launch writes g = dy * act'(y) and the block-summed gradients of the up-sampled sources; for sources
    that are train-mode conv + BN outputs (the fuse layers, registered by ConvBN.fwd in ``_BNB_FWD``) it also leaves
    their BatchNorm-backward channel sums in their slots, so that ConvBN.bwd needs norm_bwd_apply_slots only."""

    @staticmethod
    def fwd(st, lane, t, meta, needs):
        act, shifts = meta
        ins = [nhwc(v) for v in t]
        n = len(ins)
        B, C, H, W = ins[shifts.index(0)].shape
        y = empty_nhwc(B, C, H, W, ins[0].device)
        ptrs = (ctypes.c_void_p * n)(*[v.data_ptr() for v in ins])
        sh = (ctypes.c_int * n)(*shifts)
        call('advmix_fuse_sum', ptrs, sh, n, _p(y), B, H, W, C, act, st)
        targets = [None] * n
        for j, v in enumerate(ins):
            tg = _BNB_FWD.pop(v.data_ptr(), None)
            if tg is not None and needs[j] and tg[0].pass_id == tg[1] and tuple(tg[3].shape) == tuple(v.shape) \
                    and tg[3].stride() == v.stride():
                targets[j] = tg
        return (y,), (y,), (targets if any(tg is not None for tg in targets) else None)

    @staticmethod
    def bwd(st, lane, saved, extra, meta, grads, needs):
        (y,) = saved
        act, shifts = meta
        dy = nhwc(grads[0])
        B, C, H, W = y.shape
        n = len(shifts)
        need_g = any(needs[j] and s == 0 for j, s in enumerate(shifts))
        g = keep(empty_nhwc(B, C, H, W, y.device)) if need_g else None
        outs = []
        for j, s in enumerate(shifts):
            if not needs[j]:
                outs.append(None)
            elif s == 0:
                outs.append(g)
            else:
                outs.append(empty_nhwc(B, C, H >> s, W >> s, y.device))
        ptrs = (ctypes.c_void_p * n)(*[(o.data_ptr() if (o is not None and s > 0) else None)
                                      for o, s in zip(outs, shifts)])
        sh = (ctypes.c_int * n)(*shifts)
        if BNB_FUSED and not DETERMINISTIC and act in (ACT_NONE, ACT_RELU, ACT_LEAKY):
            tgs = [None] * n
            for j, tg in enumerate(extra or ()):
                if tg is not None and outs[j] is not None and tg[0].claim_bwd(tg[2], tg[1]):
                    tgs[j] = tg
            NS = 16
            vp = ctypes.c_void_p * n
            rc = lib.advmix_fuse_sum_bwd_bnb(
                _p(dy), _p(y), _p(g), ptrs, sh, n, B, H, W, C, act,
                vp(*[(tg[3].data_ptr() if tg else None) for tg in tgs]), vp(*[(tg[4].data_ptr() if tg else None) for tg in tgs]),
                vp(*[(tg[5].data_ptr() if tg else None) for tg in tgs]),
                vp(*[(tg[0].t.data_ptr() + 8 * tg[2] if tg else None) for tg in tgs]), NS, st)
            if rc == 0:
                for j, tg in enumerate(tgs):
                    if tg is not None:
                        _BNB_PRE[(id(tg[0]), tg[2])] = (tg[0], tg[1], NS)
                        COUNTERS['fuse_bnb'] = COUNTERS.get('fuse_bnb', 0) + 1
                return tuple(outs)
            for tg in tgs:                                  # not served (rc 1): release the slot claims, separate kernels
                if tg is not None:
                    tg[0].dirty.discard(tg[2])
            if rc != 1:
                raise RuntimeError('advmix_fuse_sum_bwd_bnb failed: %d' % rc)
        if g is None:
            g = keep(empty_nhwc(B, C, H, W, y.device))
        call('advmix_fuse_sum_bwd', _p(dy), _p(y), _p(g), ptrs, sh, n, B, H, W, C, act, st)
        return tuple(outs)


class MaxPool:
    NHWC = (0,)      # inputs made dense NHWC BEFORE the lanes fork (None = all)
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1)  (pose_resnet.py:115)."""

    @staticmethod
    def fwd(st, lane, t, meta, needs):
        x = nhwc(t[0])
        B, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = empty_nhwc(B, C, Ho, Wo, x.device)
        idx = torch.empty(B * Ho * Wo * C, device=x.device, dtype=torch.uint8)
        call('advmix_maxpool3x3s2', _p(x), _p(y), _p(idx), B, H, W, C, Ho, Wo, st)
        return (y,), (idx,), (B, C, H, W, Ho, Wo)

    @staticmethod
    def bwd(st, lane, saved, extra, meta, grads, needs):
        (idx,) = saved
        B, C, H, W, Ho, Wo = extra
        dy = nhwc(grads[0])
        dx = empty_nhwc(B, C, H, W, dy.device)
        call('advmix_maxpool3x3s2_bwd', _p(dy), _p(idx), _p(dx), B, H, W, C, Ho, Wo, st)
        return (dx,)


def _nchw3(v):
    if v.dim() != 4 or v.shape[1] != 3 or not v.is_contiguous() or v.dtype != torch.float32 or not v.is_cuda:
        raise ValueError('views must be contiguous NCHW fp32 CUDA tensors [B,3,H,W]')
    return v


class Mix:
    NHWC = (0,)      # inputs made dense NHWC BEFORE the lanes fork (None = all)
    """tmp = sum_k views[k] * softmax(logits, 1)[:, k:k+1]  (function.py:138-144)."""

    @staticmethod
    def fwd(st, lane, t, meta, needs):
        logits = nhwc(t[0])
        v0, v1, v2 = _nchw3(t[1]), _nchw3(t[2]), _nchw3(t[3])
        B, _, H, W = v0.shape
        tmp = empty_nhwc(B, 3, H, W, v0.device)
        call('advmix_mix_fwd', _p(v0), _p(v1), _p(v2), _p(logits), _p(tmp), B, H, W, st)
        return (tmp,), (logits, v0, v1, v2), None

    @staticmethod
    def bwd(st, lane, saved, extra, meta, grads, needs):
        logits, v0, v1, v2 = saved
        dtmp = nhwc(grads[0])
        B, _, H, W = v0.shape
        dl = empty_nhwc(B, 3, H, W, v0.device)
        call('advmix_mix_bwd', _p(v0), _p(v1), _p(v2), _p(logits), _p(dtmp), _p(dl), B, H, W, st)
        return dl, None, None, None


class JointsLoss:
    NHWC = (0,)      # inputs made dense NHWC BEFORE the lanes fork (None = all)
    """JointsMSELoss.forward (lib/core/loss.py:25-65); fused forward + gradient.
    tensors = (pred, target, tw|None[, target_b|None]); meta = (use_tw, mse[, scale_a, scale_b]).
    With scales (round 6) the op is the BLEND ``scale_a L(pred, target) + scale_b L(pred, target_b)`` in one pass over pred
    (advmix_joints_loss_blend): the student's heat-map + distillation loss (function.py:151-153), the generator's negated
    loss (:161) - no torch arithmetic around the criterion."""

    @staticmethod
    def _target(target, pred):
        if target.shape != pred.shape:
            raise ValueError('target shape %s != output shape %s' % (tuple(target.shape), tuple(pred.shape)))
        target = keep(target.float())
        if target.is_contiguous():
            return target, 0
        if target.is_contiguous(memory_format=_CL):
            return target, 1
        return keep(target.contiguous()), 0

    @staticmethod
    def fwd(st, lane, t, meta, needs):
        pred, target, tw = t[:3]
        target_b = t[3] if len(t) > 3 else None
        use_tw, mse = meta[:2]
        blend = len(meta) > 2
        sa, sb = (float(meta[2]), float(meta[3])) if blend else (1.0, 0.0)
        pred = nhwc(pred)
        B, J, H, W = pred.shape
        target, t_nhwc = JointsLoss._target(target, pred)
        tb, b_nhwc = JointsLoss._target(target_b, pred) if target_b is not None else (None, 0)
        w = keep(tw.float().reshape(B, J).contiguous()) if use_tw else None
        loss = torch.empty((), device=pred.device, dtype=torch.float32)
        call('advmix_fill', _p(loss), 0.0, 1, st)
        grad = empty_nhwc(B, J, H, W, pred.device) if needs[0] else None
        ws = _p(_workspace(pred.device, 0, lane)) if DETERMINISTIC else None
        if blend:
            call('advmix_joints_loss_blend', _p(pred), _p(target), t_nhwc, _p(tb), b_nhwc, _p(w), _p(loss), _p(grad), sa, sb,
                 B, J, H * W, 1 if mse else 0, ws, st)
        elif DETERMINISTIC:
            call('advmix_joints_loss_det', _p(pred), _p(target), t_nhwc, _p(w), _p(loss), _p(grad), 1.0,
                 B, J, H * W, 1 if mse else 0, ws, st)
        else:
            call('advmix_joints_loss', _p(pred), _p(target), t_nhwc, _p(w), _p(loss), _p(grad), 1.0,
                 B, J, H * W, 1 if mse else 0, st)
        return (loss,), ((grad,) if needs[0] else ()), None

    @staticmethod
    def bwd(st, lane, saved, extra, meta, grads, needs):
        (grad,) = saved
        unit = _UNIT.get(grad.device.index)
        if unit is not None and grads[0].data_ptr() == unit.data_ptr():
            return (grad,) + (None,) * (len(needs) - 1)     # d loss = 1 (unit_grad): the gradient the forward left IS the result
        out = torch.empty_like(grad, memory_format=torch.preserve_format)
        dl = keep(grads[0].float().contiguous())
        call('advmix_scale_dev', _p(out), _p(grad), _p(dl), 1.0, grad.numel(), st)
        return (out,) + (None,) * (len(needs) - 1)


_UNIT = {}


def unit_grad(device):
    """The constant 1.0 on ``device`` to seed a backward pass with: ``loss.backward(ops.unit_grad(loss.device))``.  The
    engine's own seed is a ones_like() - a torch fill launch per pass on the critical path between forward and backward -
    and JointsLoss.bwd recognises THIS tensor and hands the gradient its forward computed on without the scaling launch.
    Created on first use (the runners' eager warm-up steps: never inside a capture)."""
    t = _UNIT.get(device.index)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            return None                                     # (the engine's default seed: correct, one launch more)
        t = _UNIT[device.index] = torch.ones((), device=device, dtype=torch.float32)
    return t


FANIN_FUSED = __import__('os').environ.get('ADVMIX_FANIN', '1') != '0'
INBN = __import__('os').environ.get('ADVMIX_INBN', '1') != '0'        # BatchNorm + ReLU of an inner edge applied by the reading conv on load (A/B switch)


def _add(st, a, b):
    """a + b for two dense tensors of identical layout, on the member's stream (gradient fan-in
    inside a Chain; a torch add would run on the caller's stream)."""
    if a.shape != b.shape or a.stride() != b.stride():
        raise RuntimeError('advmix_amd: gradient fan-in of differently laid out tensors')
    out = keep(torch.empty_like(a, memory_format=torch.preserve_format))
    call('advmix_add', _p(a), _p(b), _p(out), a.numel(), st)
    return out


# Observation hook of the parity tests (never set by the product): callable(owner, slot, tensor, stream handle) called for
# every slot a plan step produces as soon as it has been launched - inside a chain (owner = the chain's ``subs`` tuple) and by
# plan.PlanRun for single steps (owner = the PlanNet).  The pinned-mask tests read the signs of the activations of the step's OWN
# forward passes through it (tests/plan_functional.py); the callee synchronises before it reads.  Not legal during graph capture.
SLOT_TAP = None


class Chain:
    """A dependent run of members executed back to back on ONE lane: an HRNet branch with its fuse
    convolutions, a bottleneck stack, a whole U-Net.  Inside a chain nothing waits for the other
    lanes, so branches drift apart and one branch's tiny kernels (norm finalize, fuse sums) overlap
    another's convolutions; the level-synchronous schedule (one group per plan level) joined all
    lanes after every conv+BN instead.

    tensors = (external slot tensors..., then every other tensor of every sub-member)
    meta    = (subs, ext_slots, out_slots[, bnb_plan]); subs[i] = (op, refs, sub_meta, dst_slot) with refs[j] =
              ('s', slot) | ('i', flat index into tensors) | None for the sub-member's j-th tensor.
    The sub-members are the ordinary member classes; their gradients w.r.t. a slot that has several
    consumers are summed here (``advmix_add``), which is what autograd's fan-in adds did before."""

    @staticmethod
    def NHWC(meta):
        return tuple(range(len(meta[1])))

    @staticmethod
    def n_out(meta):
        return len(meta[2])

    @staticmethod
    def inbn_plan(subs, out_slots):
        """{producer index j: consumer index k}: sub j is a train-mode conv + BatchNorm + ReLU without a residual whose output
        slot has exactly ONE reader - the conv input of sub k, a 3x3 / stride 1 / pad 1 conv + BatchNorm - and does not leave the
        chain: conv1 -> bn1 -> relu -> conv2 of a BasicBlock / Bottleneck (lib/models/pose_hrnet.py:41-57, :77-88).  For these
        pairs the activation need never be a tensor: the reader applies it while staging its input (ConvBN.fwd ``in_bn``)."""
        uses = {}
        for k, (_o, refs, _m, _d) in enumerate(subs):
            for pos, r in enumerate(refs):
                if r is not None and r[0] == 's':
                    uses.setdefault(r[1], []).append((k, pos))
        plan = {}
        for j, (op, refs, m, dst) in enumerate(subs):
            if op is not ConvBN or not m[3] or m[2] != ACT_RELU or (len(refs) > 7 and refs[7] is not None) or dst in out_slots:
                continue
            u = uses.get(dst, [])
            if len(u) != 1 or u[0][1] != 0:
                continue
            k = u[0][0]
            kop, _kr, km, _kd = subs[k]
            if kop is ConvBN and km[3] and km[0] == 1 and km[1] == 1 and k > j:
                plan[j] = k
        return plan

    @staticmethod
    def fwd(st, lane, t, meta, needs):
        subs, ext_slots, out_slots = meta[:3]
        val = {s_: t[i] for i, s_ in enumerate(ext_slots)}
        need = {s_: bool(needs[i]) for i, s_ in enumerate(ext_slots)}
        saved, rec = [], []
        plan = (meta[4] if len(meta) > 4 else Chain.inbn_plan(subs, out_slots)) if INBN else {}
        pending = {}                                       # consumer index -> the deferred producer's record
        for si, (op, refs, smeta, dst) in enumerate(subs):
            tin = tuple(None if r is None else (val[r[1]] if r[0] == 's' else t[r[1]]) for r in refs)
            nin = tuple(False if r is None else (need[r[1]] if r[0] == 's' else bool(needs[r[1]])) for r in refs)
            kw = {}
            if si in plan:                                 # defer this layer's BatchNorm + ReLU to its reader - if that reader is
                kref = subs[plan[si]][1][1]                # one the Winograd kernel will take at this batch size (its filters carry images)
                wk = val[kref[1]] if kref[0] == 's' else t[kref[1]]
                c_shape = tin[1].shape
                x_shape = tin[0].shape
                Ho = (x_shape[2] + 2 * smeta[1] - c_shape[2]) // smeta[0] + 1
                Wo = (x_shape[3] + 2 * smeta[1] - c_shape[3]) // smeta[0] + 1
                hit = _wino_images(wk, x_shape[0], Ho, Wo, wk.shape[1], wk.shape[0], wk.shape[2], wk.shape[3], 1, 1) \
                    if (wk.shape[1] == c_shape[0] and tuple(wk.shape[2:]) == (3, 3) and _direct_ok()) else None
                if hit is not None and hit[2] == 'wino' and not DETERMINISTIC:
                    kw['defer'] = True
            if si in pending:
                kw['in_bn'] = pending.pop(si)
            o, sv, ex = op.fwd(st, lane, tin, smeta, nin, **kw)
            pend = ex[6] if (kw.get('defer') and ex is not None and len(ex) > 6) else None
            if pend is not None:
                pending[plan[si]] = pend
            wants = any(nin)
            if not wants:                                  # nothing to differentiate: drop the references, but only
                _KEEP.extend(v for v in sv if torch.is_tensor(v))   # after the lanes have joined (see keep())
                sv = ()
            keep(o[0])                                     # an intermediate nobody saved must outlive the lanes
            val[dst], need[dst] = o[0], wants
            if SLOT_TAP is not None:
                if pend is None:
                    SLOT_TAP(subs, dst, o[0], st)
                if 'in_bn' in kw:                           # the activation this layer applied on load, as a tensor for the observer
                    src = refs[0][1]                        # (the reader has published the statistics by now; observation only)
                    n_mat = COUNTERS.get('inbn_materialized', 0)
                    SLOT_TAP(subs, src, ConvBN.materialize(dict(kw['in_bn'], y=None), st), st)
                    COUNTERS['inbn_materialized'] = n_mat   # (the observer's copy is not the product's)
            rec.append((len(saved), len(sv), ex, nin))
            saved += list(sv)
        return tuple(val[s_] for s_ in out_slots), tuple(saved), rec

    @staticmethod
    def bnb_plan(subs):
        """{consumer index k: producer index j}: sub k is a conv whose dgrad can carry the BatchNorm-backward
        epilogue of sub j.  Holds when k reads slot s = output of ConvBN j through its conv input only and k is the
        FIRST consumer of s in forward order - in the reversed backward sweep every other contribution to dL/ds has
        then already been summed (and rides into k's dgrad epilogue as the addend), so k's epilogue sees the
        complete gradient."""
        prod = {dst: j for j, (_o, _r, _m, dst) in enumerate(subs)}
        first_use = {}
        for k, (_o, refs, _m, _d) in enumerate(subs):
            for r in refs:
                if r is not None and r[0] == 's':
                    first_use.setdefault(r[1], k)
        plan = {}
        for k, (op, refs, _m, _d) in enumerate(subs):
            r0 = refs[0] if refs else None
            if not getattr(op, 'ADD_TO', False) or r0 is None or r0[0] != 's':
                continue
            j = prod.get(r0[1])
            if j is None or first_use.get(r0[1]) != k or sum(1 for r in refs if r == r0) != 1:
                continue
            if subs[j][0] is ConvBN and subs[j][2][3]:                # a train-mode conv + BatchNorm
                plan[k] = j
        return plan

    @staticmethod
    def bwd(st, lane, saved, rec, meta, grads, needs):
        subs, ext_slots, out_slots = meta[:3]
        plan = (meta[3] if len(meta) > 3 else Chain.bnb_plan(subs)) if BNB_FUSED else {}
        _WG_DEFER.append([])                                # this chain's weight gradients are collected ...
        try:
            return Chain._bwd(st, lane, saved, rec, subs, ext_slots, out_slots, plan, grads, needs)
        finally:
            _flush_wgrads(st, _WG_DEFER.pop())              # ... and go out grouped, on this chain's lane

    @staticmethod
    def _bwd(st, lane, saved, rec, subs, ext_slots, out_slots, plan, grads, needs):
        grad, pre = {}, {}
        for s_, g in zip(out_slots, grads):
            if g is not None:
                grad[s_] = g if s_ not in grad else _add(st, grad[s_], g)
        flat = [None] * len(needs)
        for k in range(len(subs) - 1, -1, -1):
            (op, refs, smeta, dst), (sp, sc, ex, nin) = subs[k], rec[k]
            go = grad.pop(dst, None)
            if go is None or not any(nin):
                continue
            kw = {}
            if dst in pre:                                 # go is already g = dL/dy * act'(y), sums in the slots
                kw['pre'] = pre.pop(dst)
            bnb = None
            # The epilogue must see the COMPLETE dL/dy: every other contribution rides in as the addend, which only
            # happens on the fan-in-fused path.  With ADVMIX_FANIN=0 a pending gradient of the slot would be added
            # AFTER the epilogue had multiplied this conv's share by act'(y) and summed it - so no epilogue then.
            if k in plan and nin[0] and any(rec[plan[k]][3]) and rec[plan[k]][1] \
                    and (FANIN_FUSED or refs[0][1] not in grad):
                jsp, jsc, jex, _jn = rec[plan[k]]
                bnb = ConvBN.bnb_target(saved[jsp:jsp + jsc], jex, subs[plan[k]][2])
                if bnb is not None:
                    kw['bnb'] = bnb
            if FANIN_FUSED and getattr(op, 'ADD_TO', False) and refs[0] is not None and refs[0][0] == 's' \
                    and nin[0] and refs[0][1] in grad:
                # the input already has a gradient from another consumer: the conv's dgrad epilogue adds it
                r = op.bwd(st, lane, saved[sp:sp + sc], ex, smeta, [go], nin, grad.pop(refs[0][1]), **kw)
            else:
                r = op.bwd(st, lane, saved[sp:sp + sc], ex, smeta, [go], nin, **kw)
            if bnb is not None:
                if 'done' in bnb:
                    pre[refs[0][1]] = bnb['done']
                else:                                      # shape not served: release the slot set again
                    jex = rec[plan[k]][2]
                    jex[1].dirty.discard(jex[3])
            for ref, g in zip(refs, r):
                if g is None or ref is None:
                    continue
                keep(g)
                if ref[0] == 's':
                    grad[ref[1]] = g if ref[1] not in grad else _add(st, grad[ref[1]], g)
                else:
                    flat[ref[1]] = g if flat[ref[1]] is None else _add(st, flat[ref[1]], g)
        for i, s_ in enumerate(ext_slots):
            if needs[i]:
                flat[i] = grad.get(s_)
        return flat



class GraphSeq:
    """What the runners capture into: plain torch.cuda.graph captures (one multi-branch HIP graph per ``capture`` call: the
    launch lanes' streams fork and join inside the capture) sharing one memory pool.  ``capture(fn)`` returns (segment
    id, fn's result); ``replay(segment id)`` replays that segment.  (Rounds 1-2 also had per-lane single-chain graphs
    with device-side hand-offs, "lane tapes": 2 ms instead of 46 ms of host time per step and a SLOWER step, 63.2 vs
    59.2 ms - DESIGN.md section 3; removed in round 3, the code is in history at 72f4536.)"""

    def __init__(self, device):
        self.graphs, self.pool = [], None
        self._sink = torch.zeros(64, device=device, dtype=torch.float32)
        self.device = device
        # Replays never go onto the NULL stream (round 4): with two processes sharing one GPU, the
        # SECOND and later launches of an instantiated graph on the null stream - the launches the runtime serves from its
        # captured AQL packets - computed garbage gradients in 21 of 21 two-rank runs (host fully synchronised, before any
        # exchange); on a created stream, or with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, 0 of 16.  ADVMIX_REPLAY_STREAM=null
        # restores the old behaviour for A/B runs.
        self.stream = None if REPLAY_ON_NULL else torch.cuda.Stream(device=device)

    def replay_stream(self):
        """(stream, hop): the stream a step's replays (and everything the runner enqueues between them) run on - the
        caller's current stream unless that is the NULL stream - and whether that is another stream than the current one
        (the caller then orders the two with wait_stream on both sides)."""
        cur = torch.cuda.current_stream(self.device)
        if self.stream is None or cur.cuda_stream != 0:
            return cur, False
        return self.stream, True

    def capture(self, fn):
        g = torch.cuda.CUDAGraph()
        kw = dict(capture_error_mode='thread_local')
        if self.pool is not None:
            kw['pool'] = self.pool
        with torch.cuda.graph(g, **kw):
            r = fn()
        if self.pool is None:
            self.pool = g.pool()
        self.graphs.append(g)
        return len(self.graphs) - 1, r

    def replay(self, seg):
        rs, hop = self.replay_stream()
        if not hop:
            self.graphs[seg].replay()
            return
        cur = torch.cuda.current_stream(self.device)       # a lone replay from the NULL stream: hop onto the replay stream
        rs.wait_stream(cur)
        with torch.cuda.stream(rs):
            self.graphs[seg].replay()
        cur.wait_stream(rs)

    @property
    def n_graphs(self):
        return len(self.graphs)


def _lane_order(n, nl):
    """Members in execution order: lane by lane (member i runs on lane i % nl)."""
    return [i for l in range(nl) for i in range(l, n, nl)]


def _run_lanes(dev, n, nl, run_member):
    """Fork, run member i on lane i % nl via ``run_member(i, stream handle, lane)``, join (real stream fork / join,
    eager or inside a capture)."""
    cur = torch.cuda.current_stream(dev)
    side = _lanes(dev, nl - 1)
    for s_ in side:
        s_.wait_stream(cur)
    handles = [ctypes.c_void_p(cur.cuda_stream)] + [ctypes.c_void_p(s_.cuda_stream) for s_ in side]
    for i in _lane_order(n, nl):
        run_member(i, handles[i % nl], i % nl)
    for s_ in side:
        cur.wait_stream(s_)


# =============================================================================================
class GroupFn(torch.autograd.Function):
    """Runs a list of independent members; member i on lane i % MAX_LANES (lane 0 = the
    caller's stream).  Inputs/outputs are the members' tensors, flattened."""

    @staticmethod
    def forward(ctx, members, *flat):
        dev = next(t for t in flat if t is not None).device
        n = len(members)
        cur = torch.cuda.current_stream(dev)
        nl = min(n, MAX_LANES)
        if _WG_SMALL:
            drop_stale_parked()                             # (what a backward pass that raised left parked: released, not run)
        # layout conversions are torch kernels on the caller's stream: do them BEFORE the fork so
        # no side lane can start ahead of a copy it depends on
        flat = list(flat)
        pos = 0
        for op, cnt, meta in members:
            idx = op.NHWC(meta) if callable(op.NHWC) else op.NHWC
            for k in (range(cnt) if idx is None else idx):
                if flat[pos + k] is not None:
                    flat[pos + k] = nhwc(flat[pos + k])
            pos += cnt
        _ensure_workspaces(dev, nl)
        needs_all = ctx.needs_input_grad[1:]
        starts, pos = [], 0
        for op, cnt, meta in members:
            starts.append(pos)
            pos += cnt
        results = [None] * n

        def run_member(i, handle, lane):
            op, cnt, meta = members[i]
            p0 = starts[i]
            results[i] = op.fwd(handle, lane, flat[p0:p0 + cnt], meta, needs_all[p0:p0 + cnt])
        _run_lanes(dev, n, nl, run_member)
        outs, saved, spans, extras, nondiff = [], [], [], [], []
        for i, (op, cnt, meta) in enumerate(members):
            o, sv, ex = results[i]
            if not any(needs_all[starts[i]:starts[i] + cnt]):   # frozen member (e.g. the teacher riding along): nothing
                _KEEP.extend(v for v in sv if torch.is_tensor(v))   # to differentiate; its temporaries were alive until
                sv = ()                                    # the join (``results``), which is all they need
                nondiff += list(o)
            spans.append((starts[i], cnt, len(outs), len(o), len(saved), len(sv)))
            outs += list(o)
            saved += list(sv)
            extras.append(ex)
        del _KEEP[:]
        ctx.members, ctx.spans, ctx.extras = members, spans, extras
        ctx.n_in = len(flat)
        ctx.save_for_backward(*saved)
        ctx.set_materialize_grads(False)
        if nondiff:
            ctx.mark_non_differentiable(*nondiff)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        saved = ctx.saved_tensors
        dev = next(g for g in gouts if g is not None).device
        members = ctx.members
        nl = min(len(members), MAX_LANES)
        cur = torch.cuda.current_stream(dev)
        gouts = [nhwc(g) if (g is not None and g.dim() == 4) else g for g in gouts]   # before the fork
        _ensure_workspaces(dev, nl)
        grads = [None] * ctx.n_in
        needs_all = ctx.needs_input_grad[1:]
        spans, extras = ctx.spans, ctx.extras

        def run_member(i, handle, lane):
            op, cnt, meta = members[i]
            ipos, icnt, opos, ocnt, spos, scnt = spans[i]
            g = gouts[opos:opos + ocnt]
            needs = needs_all[ipos:ipos + icnt]
            if all(x is None for x in g) or not any(needs):
                return
            r = op.bwd(handle, lane, saved[spos:spos + scnt], extras[i], meta, g, needs)
            grads[ipos:ipos + icnt] = list(r) + [None] * (icnt - len(r))
        key = _pass_key()
        ready = len(_park_list(key))                        # small weight gradients parked by EARLIER groups of THIS pass (their
        if nl > 1 and ready >= WGRAD_MULTI_FLUSH:           # lanes have joined): one mixed launch on this group's last lane,
            inner = run_member                              # beside its members

            def run_member(i, handle, lane):
                if i == nl - 1:                             # (the first member that runs on lane nl - 1)
                    _flush_small_wgrads(handle, key, ready)
                inner(i, handle, lane)
        _run_lanes(dev, len(members), nl, run_member)
        del _KEEP[:]
        return (None,) + tuple(grads)


def run_group(members):
    """members: list of (op, tensors tuple, meta).  Returns one entry per member: its output
    tensor, or for a Chain the tuple of its output slots."""
    flat, spec = [], []
    for op, tensors, meta in members:
        spec.append((op, len(tensors), meta))
        flat += list(tensors)
    outs = GroupFn.apply(spec, *flat)
    res, pos = [], 0
    for op, tensors, meta in members:
        if hasattr(op, 'n_out'):                           # a Chain returns the tuple of its output slots
            n = op.n_out(meta)
            res.append(tuple(outs[pos:pos + n]))
            pos += n
        else:
            res.append(outs[pos])
            pos += 1
    return res


def _one(op, tensors, meta):
    return run_group([(op, tuple(tensors), meta)])[0]


# ---- functional spellings ---------------------------------------------------------------------
def conv2d(x, w, bias=None, stride=1, pad=0):
    return _one(Conv, (x, w, bias), (stride, pad))


def conv_transpose2d(x, w, bias=None, stride=2, pad=1):
    return _one(Deconv, (x, w, bias), (stride, pad))


def batch_norm(x, gamma, beta, rmean, rvar, nbt, residual=None, act=ACT_NONE, training=True,
               momentum=0.1, eps=1e-5):
    return _one(BatchNorm, (x, gamma, beta, rmean, rvar, nbt, residual), (act, training, momentum, eps))


def conv_bn(x, w, gamma, beta, rmean, rvar, nbt, residual=None, stride=1, pad=0, act=ACT_NONE, training=True,
            momentum=0.1, eps=1e-5):
    return _one(ConvBN, (x, w, gamma, beta, rmean, rvar, nbt, residual), (stride, pad, act, training, momentum, eps))


def instance_norm(x, act=ACT_NONE, eps=1e-5):
    return _one(InstanceNorm, (x,), (act, eps))


def activation(x, act):
    return _one(Act, (x,), act)


def cat_act(a, b, act=ACT_NONE):
    return _one(CatAct, (a, b), act)


def fuse_sum(ins, shifts, act=ACT_RELU):
    return _one(FuseSum, tuple(ins), (act, list(shifts)))


def max_pool3x3s2(x):
    return _one(MaxPool, (x,), None)


def softmax_mix(logits, views):
    return _one(Mix, (logits, views[0], views[1], views[2]), None)


def joints_loss(pred, target, tw, use_target_weight=True, mse=False):
    return _one(JointsLoss, (pred, target, tw), (use_target_weight, mse))


def joints_loss_blend(pred, target_a, scale_a, target_b, scale_b, tw, use_target_weight=True, mse=False):
    """scale_a L(pred, target_a) + scale_b L(pred, target_b) as ONE op (``target_b`` None: a scaled single loss)."""
    return _one(JointsLoss, (pred, target_a, tw, target_b), (use_target_weight, mse, float(scale_a), float(scale_b)))


def cat_views(views):
    """torch.cat(inputs, dim=1) for the 3 NCHW views -> NHWC [B,9,H,W] (function.py:137)."""
    v0, v1, v2 = (_nchw3(v) for v in views)
    B, _, H, W = v0.shape
    out = empty_nhwc(B, 9, H, W, v0.device)
    call('advmix_cat_views', _p(v0), _p(v1), _p(v2), _p(out), B, H, W, _st())
    return out


def heatmap_argmax(hm, out=None):
    """(idx int32 [B,J], max [B,J]) of a [B,J,H,W] heat-map; first occurrence (numpy.argmax).  ``out`` = (idx, max)
    tensors to write into (contiguous, B * J elements each) instead of fresh ones."""
    if not hm.is_cuda or hm.dtype != torch.float32:
        raise TypeError('heatmap_argmax needs an fp32 CUDA tensor')
    B, J, H, W = hm.shape
    if hm.is_contiguous():
        fmt = 0
    else:
        hm, fmt = nhwc(hm), 1
    if out is None:
        idx = torch.empty((B, J), device=hm.device, dtype=torch.int32)
        mx = torch.empty((B, J), device=hm.device, dtype=torch.float32)
    else:
        idx, mx = out
        if idx.dtype != torch.int32 or mx.dtype != torch.float32 or idx.numel() != B * J or mx.numel() != B * J \
                or not idx.is_contiguous() or not mx.is_contiguous():
            raise TypeError('heatmap_argmax: out = (int32, float32) contiguous tensors of B * J elements')
    call('advmix_heatmap_argmax', _p(hm), fmt, _p(idx), _p(mx), B, J, H * W, _st())
    return idx, mx


# ---- validate(): flip test and final predictions ---------------------------------------------

def _dense_layout(t):
    """(tensor, nhwc flag) for a logical-NCHW fp32 CUDA tensor that is dense in NCHW or NHWC memory."""
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4:
        raise TypeError('needs a 4-D fp32 CUDA tensor')
    if t.is_contiguous():
        return t, 0
    return nhwc(t), 1


def flip_w(x, channels_last=True):
    """``x.flip(3)`` (function.py:241) of a dense NCHW batch, written straight into the layout the
    networks consume (NHWC when ``channels_last``)."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        raise TypeError('flip_w needs a 4-D fp32 CUDA tensor')
    x = x.contiguous()
    B, C, H, W = x.shape
    y = empty_nhwc(B, C, H, W, x.device) if channels_last else torch.empty_like(x)
    call('advmix_flip_w', _p(x), _p(y), B, C, H, W, 1 if channels_last else 0, _st())
    return y


_partner_cache = {}


def flip_partner(flip_pairs, J, device):
    """int32 [J] permutation of ``flip_back``'s pairwise swaps (transforms.py:36-39), cached per device."""
    key = (tuple(tuple(int(v) for v in p) for p in flip_pairs), J, device.index)
    t = _partner_cache.get(key)
    if t is None:
        perm = list(range(J))
        for a, b in key[0]:                      # sequential swaps, exactly like the reference loop
            perm[a], perm[b] = perm[b], perm[a]
        t = torch.tensor(perm, dtype=torch.int32, device=device)
        _partner_cache[key] = t
    return t


def flip_merge(output, output_flipped, flip_pairs, shift, merge=True):
    """Flip-test merge on the device: ``(output + shift(flip_back(output_flipped))) * 0.5``
    (function.py:249-261); ``merge=False`` returns the flipped-back (and shifted) maps alone."""
    of, fmt = _dense_layout(output_flipped.detach())
    B, J, H, W = of.shape
    o = None
    if merge:
        o = output.detach()
        o = nhwc(o) if fmt else o.contiguous()
        if o.shape != of.shape:
            raise ValueError('flip_merge: shape mismatch')
    y = empty_nhwc(B, J, H, W, of.device) if fmt else torch.empty_like(of)
    call('advmix_flip_merge', _p(o), _p(of), _p(flip_partner(flip_pairs, J, of.device)), _p(y), B, J, H, W, fmt,
         1 if shift else 0, _st())
    return y


def final_preds(hm, center, scale, post_process):
    """Device ``get_final_preds`` (inference.py:52-95): returns (coords [B,J,2] heat-map space,
    preds [B,J,2] image space, maxvals [B,J]) as CUDA tensors.  center/scale: [B,2] (any float dtype/host)."""
    hm, fmt = _dense_layout(hm.detach())
    B, J, H, W = hm.shape
    c = torch.as_tensor(center, dtype=torch.float32).reshape(B, 2).to(hm.device).contiguous()
    s = torch.as_tensor(scale, dtype=torch.float32).reshape(B, 2).to(hm.device).contiguous()
    coords = torch.empty((B, J, 2), device=hm.device, dtype=torch.float32)
    preds = torch.empty((B, J, 2), device=hm.device, dtype=torch.float32)
    mx = torch.empty((B, J), device=hm.device, dtype=torch.float32)
    call('advmix_final_preds', _p(hm), fmt, _p(c), _p(s), B, J, H, W, 1 if post_process else 0, _p(coords),
         _p(preds), _p(mx), _st())
    return coords, preds, mx
