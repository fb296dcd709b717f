"""Data parallelism for the AdvMix step: one process per GPU, full replicas, and exactly two
gradient all-reduces per step (D grads before optimizer.step(), G grads before
optimizer_G.step()) on the optimizers' flat fp32 gradient buffers - RCCL over xGMI on the
GPU (backend "nccl"), gloo in the CPU tests.

Replaces the reference's single-process nn.DataParallel (tools/train.py:69,106,109), which
re-broadcasts ~110 MiB of weights and scatters/gathers activations through GPU 0 four times
per step.  BatchNorm statistics stay per replica, as they are per shard under DataParallel.
The loss is a local-batch mean, so averaging gradients over equal shards reproduces the
reference's global-batch mean (lib/core/function.py:151-153)."""
import contextlib

import torch
import torch.distributed as dist
import torch.utils.data
import torch.utils.data.distributed


class GradSync:
    def __init__(self, bucket_mb=64, group=None, force=False):
        self.group = group
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.force = force          # run the collectives even with one rank (single-GPU test of the path)
        self._side = None
        self.pieces = 3             # the backward pass is cut into this many pieces per network (see reduce_async)
        self._cuts = {}
        self.trace = None           # a list: reduce_async records (flat, lo, hi, copy before, copy after) of every exchange
        # self-diagnosis of an N-rank run: with ``meter`` on, finish() brackets its wait with two events on
        # the compute stream and reduce_async counts the bytes it hands to the collective; exchange_report() reads them
        self.meter = False
        self._waits, self._bytes, self._exchanges = [], 0, 0
        # RCCL ("nccl") averages in the collective; any other backend (gloo: the CPU tests, and the two-ranks-on-one-GPU
        # test of the whole path) sums and scales
        self._avg = self.world > 1 and dist.get_backend(group) == 'nccl' or (self.world == 1 and force and
                                                                            dist.is_initialized() and dist.get_backend(group) == 'nccl')

    @property
    def active(self):
        return self.world > 1 or self.force

    def cuts_for(self, net):
        """Level boundaries at which ``net``'s backward pass is cut (PlanNet.plan_cuts), chosen once per network."""
        key = id(net)
        if key not in self._cuts or self._cuts[key][0] is not net:
            self._cuts[key] = (net, net.plan_cuts(self.pieces) if hasattr(net, 'plan_cuts') else [])
        return self._cuts[key][1]

    def reduce_async(self, flat, lo, hi):
        """Average flat[lo:hi] over the ranks on the side stream, ordered after everything enqueued so far on the
        current stream - and WITHOUT making the current stream wait: the caller goes on with the next piece of the
        backward pass (whose kernels write other ranges of ``flat``) and calls ``finish()`` before the optimizer."""
        if not self.active or hi <= lo:
            return
        chunk = flat[lo:hi]
        if self.meter:
            self._bytes += (hi - lo) * flat.element_size()
            self._exchanges += 1
        if flat.is_cuda:
            cur = torch.cuda.current_stream(flat.device)
            side = self._side_stream(flat.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                pre = chunk.clone() if self.trace is not None else None     # what THIS rank handed to the exchange
                for b in range(0, hi - lo, self.bucket_elems):
                    self._mean_(chunk[b:b + self.bucket_elems])
                if pre is not None:
                    self.trace.append((flat, lo, hi, pre, chunk.clone()))
        else:
            pre = chunk.clone() if self.trace is not None else None
            dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
            chunk.div_(self.world)
            if pre is not None:
                self.trace.append((flat, lo, hi, pre, chunk.clone()))

    def _mean_(self, t):
        """In-place mean over the ranks of a device tensor, on the current (side) stream."""
        if self._avg:
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.mul_(1.0 / self.world)

    def finish(self):
        """The current stream waits for every reduce_async issued so far."""
        if self._side is not None:
            cur = torch.cuda.current_stream(self._side.device)
            if self.meter and not torch.cuda.is_current_stream_capturing():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)                              # done when the compute stream has reached the wait ...
                cur.wait_stream(self._side)
                e1.record(cur)                              # ... done when the exchange has let it pass
                self._waits.append((e0, e1))
            else:
                cur.wait_stream(self._side)

    def exchange_report(self, steps=1, reset=True):
        """{'exchange_wait_ms': time per step the compute stream stood in finish() waiting for the gradient exchange (what an
        all-reduce that is NOT hidden behind the backward pass costs the step), 'exchange_bytes': bytes handed to the
        collectives per step, 'exchanges': collective ranges per step} since ``meter`` was switched on / the last report.
        Synchronises the device (reads event times)."""
        wait = 0.0
        if self._waits:
            torch.cuda.synchronize(self._side.device)
            wait = sum(a.elapsed_time(b) for a, b in self._waits)
        out = {'exchange_wait_ms': wait / max(steps, 1), 'exchange_bytes': self._bytes // max(steps, 1),
               'exchanges': self._exchanges / max(steps, 1)}
        if reset:
            self._waits, self._bytes, self._exchanges = [], 0, 0
        return out

    def _side_stream(self, device):
        if self._side is None and device.type == 'cuda':
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def all_reduce_mean(self, flat):
        """Average ``flat`` (1-D fp32) over ranks in place.  On CUDA the bucketed collectives
        run on a side HIP stream (so bucket k+1 overlaps the scaling of bucket k) and the
        current stream waits for them before the optimizer reads the buffer."""
        if self.world == 1 and not self.force:
            return
        n = flat.numel()
        if flat.is_cuda:
            cur = torch.cuda.current_stream(flat.device)
            side = self._side_stream(flat.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for lo in range(0, n, self.bucket_elems):
                    self._mean_(flat[lo:min(n, lo + self.bucket_elems)])
            cur.wait_stream(side)
        else:
            for lo in range(0, n, self.bucket_elems):
                chunk = flat[lo:min(n, lo + self.bucket_elems)]
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
                chunk.div_(self.world)

    def sync(self, optimizer):
        """Average the optimizer's gradients over ranks: ONE flat buffer for FlatAdam; any other
        torch.optim.Optimizer (host tests, the SGD branch) falls back to its per-parameter gradients."""
        flat = getattr(optimizer, 'flat_grads', None)
        if flat is not None:
            self.all_reduce_mean(flat)
            return
        for g in optimizer.param_groups:
            for p in g['params']:
                if p.grad is not None:
                    self.all_reduce_mean(p.grad.view(-1))

    # ---- in-band evidence that an N-rank run was correct ------------------------------------------------------------
    # nn.DataParallel re-broadcasts GPU 0's weights before every forward (tools/train.py:69,106,109; lib/core/function.py:
    # 138,146,160), so the reference's replicas CANNOT drift.  Separate processes can - silently, through a bad exchange -
    # so the N-rank entry points check: what came out of an exchange is the mean of what the ranks put in
    # (``verify_trace``), and the replicas' parameters / optimizer state are still the same bits (``replicas_state``).
    @contextlib.contextmanager
    def off_null(self, t):
        """Issue a collective on a device tensor with a NON-NULL stream current (the side stream, ordered behind the caller's
        stream and the caller's stream behind it).  Round 4 (tools/dp_graph_repro.py): with two ranks
        over gloo on one GPU, ANY collective issued while the NULL stream was current - the test's own all_gather checks, the
        exchange itself in one variant - made later HIP-graph replays (the runtime's captured-packet launches) compute
        garbage, 28 of 29 runs; with everything between the replays under a created stream, 0 of 11, whichever stream the replays ran on.
        Not the collectives alone: the kernels of ``replicas_state``'s fold on the NULL stream (its all_gather already off
        it) did the same, 2 of 2, a synchronous loss.item() there did not.  The exchange proper always ran on the side
        stream; this keeps every OTHER piece of data-parallel bookkeeping off the NULL stream too."""
        if not (torch.is_tensor(t) and t.is_cuda):
            yield
            return
        cur = torch.cuda.current_stream(t.device)
        if cur.cuda_stream != 0:                            # (also the nested case: already on the side stream)
            yield
            return
        side = self._side_stream(t.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            yield
        cur.wait_stream(side)

    def _collective_device(self):
        """Where a tensor must live for this group's collectives: the current GPU for RCCL, the host otherwise."""
        if dist.is_available() and dist.is_initialized() and dist.get_backend(self.group) == 'nccl' and torch.cuda.is_available():
            return torch.device('cuda', torch.cuda.current_device())
        return torch.device('cpu')

    def _gather(self, t):
        if self.world == 1 or not (dist.is_available() and dist.is_initialized()):
            return [t]
        t = t.contiguous()
        got = [torch.empty_like(t) for _ in range(self.world)]      # (allocated on the caller's stream)
        with self.off_null(t):
            dist.all_gather(got, t, group=self.group)
        return got

    def verify_trace(self, rtol=1e-4):
        """For every exchange recorded in ``self.trace``: all-gather the ranks' inputs and compare their mean with what
        the exchange left in the buffer.  Returns (ok on EVERY rank, worst |got - mean| / max|mean| over the ranks).  Exact for the sum-and-scale
        transports at two ranks; RCCL's AVG and larger rings add in another order (a few ulps of the largest operand per
        element), hence ``rtol`` of the range's largest element - a wrong or partial exchange is off by O(1) of it."""
        ok, worst = True, 0.0
        for flat, lo, hi, pre, post in (self.trace or ()):
            parts = self._gather(pre)
            want = parts[0].double()
            for q in parts[1:]:
                want += q.double()
            want /= len(parts)
            scale = float(want.abs().max())
            err = float((post.double() - want).abs().max()) if post.numel() else 0.0
            fin = bool(torch.isfinite(post).all()) and bool(torch.isfinite(pre).all())
            rel = err / scale if scale > 0 else (0.0 if err == 0 else float('inf'))
            worst = max(worst, rel)
            ok = ok and fin and rel <= rtol
        dev = self.trace[0][3].device if self.trace else self._collective_device()      # (an NCCL group cannot gather a CPU tensor)
        verdict = torch.stack(self._gather(torch.tensor([0.0 if ok else 1.0, min(worst, 3e38)], device=dev,
                                                        dtype=torch.float32))).cpu()
        return bool((verdict[:, 0] == 0).all()), float(verdict[:, 1].max())          # the same answer on every rank

    @staticmethod
    def state_fold(tensors):
        """Three int64 numbers over the raw bits of ``tensors``: wrapping sum, position-weighted wrapping sum, count of
        non-finite elements.  Equal bits -> equal folds; one flipped bit changes the first two."""
        acc = torch.zeros(3, dtype=torch.int64, device=tensors[0].device)
        CH = 1 << 22                                        # fixed-size pieces: a flat optimizer buffer of 30-60 M
        w = None                                            # elements used to be widened to int64 three times over in one go
        for t in tensors:                                   # (~24 B / parameter of transient memory in the middle of training)
            src = t.detach().contiguous().view(-1)          # the ORIGINAL elements: the finiteness count reads these (ADVICE r5:
            if src.element_size() % 4:                      # the widened copy of an fp16 / bf16 buffer is integer, so its NaNs
                # were never counted); 1- and 2-byte elements are widened to one int32 word each (raw bit pattern for floats)
                bits = (src.view(torch.int16) if src.is_floating_point() else src).to(torch.int32)
            else:
                bits = src.view(torch.int32)
            wide = max(1, src.element_size() // 4)          # int32 words per source element (fp64 / int64: 2)
            for lo in range(0, bits.numel(), CH):
                piece = bits[lo:lo + CH]
                if src.is_floating_point():                 # (an element is counted with the piece that holds its first word)
                    acc[2] += (~torch.isfinite(src[-(-lo // wide):-(-(lo + piece.numel()) // wide)])).sum()
                if w is None or w.device != piece.device:
                    w = torch.arange(CH, device=piece.device, dtype=torch.int64)
                v = piece.to(torch.int64)
                acc[0] += v.sum()
                acc[1] += (v * ((w[:piece.numel()] + lo) % 8191 + 1)).sum()
        return acc

    def replicas_state(self, optimizers):
        """{'identical': every rank's parameters / optimizer moments / step counters fold to the same numbers,
        'finite': none of them holds an inf / NaN on any rank}.  One 24-byte all-gather per call."""
        tensors = []
        for opt in optimizers:
            if opt is None:
                continue
            if hasattr(opt, 'flat_state'):
                tensors += opt.flat_state()
            else:
                for g in opt.param_groups:
                    for p in g['params']:
                        tensors.append(p.data)
                        st = opt.state.get(p, {})
                        tensors += [st[k] for k in sorted(st) if torch.is_tensor(st[k])]
        if not tensors:
            return {'identical': True, 'finite': True}
        with self.off_null(tensors[0]):                     # the fold's kernels too: nothing of this on the NULL stream
            fold = self.state_fold(tensors)
            got = torch.stack(self._gather(fold)).cpu()
        return {'identical': bool((got[:, :2] == got[0, :2]).all()), 'finite': bool((got[:, 2] == 0).all())}

    def assert_replicas(self, optimizers, where=''):
        st = self.replicas_state(optimizers)
        if not (st['identical'] and st['finite']):
            raise RuntimeError('advmix_amd: data-parallel replicas %s%s - the ranks no longer train the same model'
                               % ('diverged' if not st['identical'] else 'hold non-finite state', where and ' (' + where + ')'))

    def broadcast_state(self, models, optimizers, src=0):
        """Make every replica start from rank ``src``'s state.  The reference's single-process nn.DataParallel
        re-broadcasts GPU 0's weights before every forward (tools/train.py:69,106,109) and never seeds torch, so
        with one process per GPU each rank would otherwise start from its own random G / final_layer / D and the
        averaged gradients would be applied to different weights.  Sent once, after ``get_optimizer``: the flat
        parameter buffer, Adam moments and step counter of each optimizer, then every buffer (BatchNorm running
        statistics, num_batches_tracked) and every parameter no optimizer owns (the teacher) of each model."""
        if self.world == 1 and not self.force:
            return
        owned = set()
        tensors = []
        for opt in optimizers:
            if opt is None:
                continue
            for g in opt.param_groups:
                owned.update(id(p) for p in g['params'])
            if hasattr(opt, 'flat_state'):
                tensors += opt.flat_state()
            else:
                for g in opt.param_groups:
                    for p in g['params']:
                        tensors.append(p.data)
                        st = opt.state.get(p, {})
                        tensors += [st[k] for k in sorted(st) if torch.is_tensor(st[k])]
        for m in models:
            if m is None:
                continue
            tensors += [p.data for p in m.parameters() if id(p) not in owned]
            tensors += list(m.buffers())
        for t in tensors:
            with self.off_null(t):
                dist.broadcast(t, src, group=self.group)
        for m in models:                                    # (the collectives wrote the parameters behind torch's back: a frozen
            if m is not None and hasattr(getattr(m, 'module', m), 'invalidate_filter_images'):   # network's filter images are keyed on their
                getattr(m, 'module', m).invalidate_filter_images()                               # version counters - plan.PlanNet._wino_refresh)
        if tensors and tensors[0].is_cuda:
            torch.cuda.current_stream(tensors[0].device).synchronize()

    def checkpoint_rank(self):
        """The reference saves ``model.module.state_dict()`` = GPU 0's replica (tools/train.py:311-337), whose
        BatchNorm running statistics come from GPU 0's shard only (nn.DataParallel updates replica 0's buffers in
        place and discards the others).  Here rank 0's buffers ARE that shard's statistics: only rank 0 writes
        checkpoints, and nothing is averaged."""
        return (dist.get_rank(self.group) if self.world > 1 else 0) == 0


class Replica(torch.nn.Module):
    """What ``torch.nn.DataParallel(model, device_ids=cfg.GPUS)`` becomes with one process per GPU: a wrapper that
    keeps DataParallel's SHAPE - ``.module``, ``module.``-prefixed state-dict keys, ``forward`` delegating to the wrapped
    network - and none of its mechanics (no scatter / replicate / gather: this process owns one GPU and one replica;
    gradients are exchanged by GradSync).  The reference's wiring relies on that shape: tools/train.py:198-235 prefixes
    the keys of ``--load_from_D`` / ``--load_from_G`` checkpoints with ``module.`` before matching them against
    ``model.state_dict()``, :315,325,337 save ``model.module.state_dict()``, and AUTO_RESUME reloads the prefixed
    ``state_dict`` (:244).  Binding it as ``torch.nn.DataParallel`` lets tools/train.py:69,
    106,109 run unchanged.  Attributes the wrapper does not have (``plan_cuts``, ``init_weights``, ...) resolve on
    the wrapped network."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids) if device_ids is not None else None      # recorded, not used: one GPU per process

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if name == 'module':
                raise
            return getattr(super().__getattr__('module'), name)


def unwrap(model):
    """The network inside a Replica / DataParallel-shaped wrapper (or the model itself)."""
    return model.module if isinstance(model, Replica) else model


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class ShardedDataLoader(torch.utils.data.DataLoader):
    """``torch.utils.data.DataLoader`` as tools/train.py:165-178 constructs it, for one process per GPU.  The reference's
    single process loads ``BATCH_SIZE_PER_GPU * len(GPUS)`` shuffled samples per iteration and nn.DataParallel scatters
    them; with N ranks every rank would load that GLOBAL batch from its own private shuffle - N times the work on
    overlapping samples.  Under an initialised process group of N > 1 ranks a SHUFFLED loader (the training one) instead
    draws from a ``DistributedSampler`` (disjoint shards of one common permutation, re-drawn every epoch: ``set_epoch`` is
    called per ``__iter__``, which the reference loop has no line for) with ``batch_size // N`` samples per iteration -
    the per-GPU batch when N = len(GPUS).  Unshuffled loaders are left whole - core.function.validate shards a validation loader's
    batches over the ranks itself, and the training loops ask ``for_training()`` for a sharded twin of an unshuffled one.  Without a process group this IS DataLoader.
    Bound by the INTEGRATION.md recipe as ``torch.utils.data.DataLoader``."""

    def __init__(self, dataset, batch_size=1, shuffle=False, sampler=None, batch_sampler=None, **kw):
        rank, world = _world()
        self._epoch = 0
        self._ctor = (dataset, batch_size, kw) if (sampler is None and batch_sampler is None) else None
        self._train_copy = None
        self._sharded = world > 1 and shuffle and sampler is None and batch_sampler is None and batch_size is not None
        if self._sharded:
            if batch_size % world:
                raise ValueError('global batch %d does not divide over %d ranks' % (batch_size, world))
            sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True)
            shuffle, batch_size = False, batch_size // world
        super().__init__(dataset, batch_size=batch_size, shuffle=shuffle, sampler=sampler, batch_sampler=batch_sampler, **kw)

    def set_epoch(self, epoch):
        """The permutation the NEXT ``__iter__`` draws: the training loops pass their ``epoch`` argument, so a run
        resumed at epoch k (AUTO_RESUME, tools/train.py:238-269) continues with permutation k instead of replaying 0..k."""
        self._epoch = int(epoch)

    def for_training(self):
        """What a TRAINING loop should iterate.  A shuffled loader is already this rank's shard; an UNSHUFFLED one
        (TRAIN.SHUFFLE false) was left whole for validation's sake - every rank would load the full global batch of identical
        samples - so the training loops get a sharded twin of it (DistributedSampler(shuffle=False), batch // N), built once."""
        rank, world = _world()
        if self._sharded or world <= 1 or self._ctor is None:
            return self
        if self._train_copy is None:
            dataset, batch_size, kw = self._ctor
            if batch_size is None or batch_size % world:
                import warnings
                warnings.warn('advmix_amd: an unshuffled training loader whose batch does not divide over %d ranks stays whole: '
                              'every rank loads every sample' % world)
                return self
            smp = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=False)
            self._train_copy = torch.utils.data.DataLoader(dataset, batch_size=batch_size // world, shuffle=False, sampler=smp, **kw)
        return self._train_copy

    def __iter__(self):
        if self._sharded:
            self.sampler.set_epoch(self._epoch)
            self._epoch += 1
        return super().__iter__()


def rank0_only(fn):
    """``fn`` on rank 0, a no-op returning None on the other ranks of an initialised process group (for the reference's
    unguarded writers, e.g. ``torch.save(model.module.state_dict(), final_state.pth)`` at tools/train.py:337, which every
    rank would otherwise write to the same path)."""
    import functools

    @functools.wraps(fn)
    def guarded(*a, **k):
        if _world()[0] != 0:
            return None
        return fn(*a, **k)
    return guarded


def atomic_copy(fn):
    """``shutil.copy2``-shaped ``fn`` whose destination appears in ONE step: the copy goes to a private temporary name in
    the destination's directory and is renamed over the target.  For the reference's three ``shutil.copy2`` calls into
    ``final_output_dir`` (tools/train.py:73-84), which N ranks would otherwise write through the same path at the same
    time (open-for-write truncates: a reader - or the other writer's ``copystat`` - can meet a half-written file).  Every
    caller keeps ``copy2``'s semantics and return value; without a process group the rename is the only difference."""
    import functools
    import os

    @functools.wraps(fn)
    def copy(src, dst, *a, **k):
        target = os.path.join(dst, os.path.basename(src)) if os.path.isdir(dst) else dst
        tmp = '%s.tmp.%d.%d' % (target, os.getpid(), _world()[0])
        try:
            fn(src, tmp, *a, **k)
            os.replace(tmp, target)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        return target
    return copy
