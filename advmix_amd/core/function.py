"""Mirror of lib/core/function.py: ``train`` (:30-95), ``set_require_grad`` (:98-104),
``train_advmix`` (:107-197), ``validate`` (:200-358), ``AverageMeter`` (:383-398) with
identical signatures.  The per-batch bodies are factored into ``plain_step`` / ``advmix_step``
so bench.py, the HIP-graph runner and the loops share one implementation.

What changed underneath (MI355X-first): models run hand-written HIP kernels; the 3-view
concat and the softmax-mix are single fused kernels; the accuracy argmax runs on the device;
gradients of per-GPU replicas are averaged with two flat RCCL all-reduces (``dp.GradSync``)
instead of nn.DataParallel's broadcast/scatter/gather."""
import logging
import os
import time

import numpy as np
import torch

from .. import ops
from .evaluate import accuracy
from .inference import get_final_preds

logger = logging.getLogger(__name__)


def set_require_grad(nets, requires_grad=True):
    if not isinstance(nets, list):
        nets = [nets]
    for net in nets:
        if net is not None:
            for param in net.parameters():
                param.requires_grad = requires_grad


def _cuda(t):
    return t.cuda(non_blocking=True) if not t.is_cuda else t


def _backward(loss):
    """``loss.backward()`` seeded with the cached constant 1.0 (ops.unit_grad) instead of the engine's ones_like launch."""
    seed = ops.unit_grad(loss.device) if loss.is_cuda and loss.dim() == 0 and loss.dtype == torch.float32 else None
    loss.backward(seed)


def _blend(criterion, output, target_a, scale_a, target_b, scale_b, target_weight):
    """scale_a L(output, target_a) + scale_b L(output, target_b) (function.py:151-153, :161): one fused op when the criterion
    is this package's (core.loss.JointsMSELoss.blend), the reference's arithmetic around any other criterion."""
    if hasattr(criterion, 'blend'):
        return criterion.blend(output, target_a, scale_a, target_b, scale_b, target_weight)
    loss = criterion(output, target_a, target_weight) * scale_a
    return loss if target_b is None else loss + criterion(output, target_b, target_weight) * scale_b


def plain_step(model, criterion, optimizer, input, target, target_weight, grad_sync=None):
    """function.py:48-59: forward, loss, zero_grad, backward, step.  Returns (loss, outputs)."""
    outputs = model(_cuda(input).float())
    loss = criterion(outputs, target, target_weight)
    optimizer.zero_grad()
    _backward(loss)
    if grad_sync is not None:
        grad_sync.sync(optimizer)
    optimizer.step()
    return loss.detach(), outputs.detach()


def _backward_pieces(loss, net, cuts, pairs):
    """The backward pass from ``loss`` as a list of closures, one per piece of ``net``.  ``cuts`` = net.plan_cuts();
    ``pairs`` = net.last_cuts of the forward that ran with ``net.cut_levels`` armed: per cut the (original, detached
    twin) activations alive across it.  Piece 0 runs from the loss down to the twins of the highest cut (autograd
    stops there: they are leaves), piece k restarts from that cut's originals with the gradients the twins collected,
    the last one reaches the network's leaves.  Each closure returns the range [lo, hi) of the flat gradient buffer
    that is COMPLETE once it has run (parameters are laid out in execution order, utils.FlatAdam; weight gradients
    are side effects of the autograd nodes, ops.GroupFn), so the data-parallel all-reduce of that range can start
    while the next piece computes."""
    assert len(pairs) == len(cuts), (len(pairs), len(cuts))
    total = net._level_elems[-1]
    bounds = [c[1] for c in cuts] + [total]

    def make(k):                                            # k counts DOWN from the top piece (k = len(cuts))
        def run():
            if k == len(cuts):
                _backward(loss)
            else:
                live = [(o, t.grad) for o, t in pairs[k] if t.grad is not None]
                torch.autograd.backward([o for o, _ in live], [g for _, g in live])
                for _, t in pairs[k]:
                    t.grad = None
            return (bounds[k - 1] if k > 0 else 0), bounds[k]
        return run
    return [make(k) for k in range(len(cuts), -1, -1)]


# The teacher's forward rides in the student's launch groups (advmix_phase_a).  Round 1 measured this as a loss (86.3 vs
# 85.8 ms); with round 3's chains it is a small, repeatable gain on the same box (ResNet-50 41.78 -> 41.13 ms, HRNet-W32
# 58.22 -> 57.67 ms, profiles/r03_pair_*.log).  Starting the teacher beside the GENERATOR's forward instead and letting it
# run on into the student's is no better (41.3 / 57.9 ms, profiles/r03_pair2_*.log); with the teacher's forward removed
# altogether (not a valid step: a bound) the step takes 37.3 / 51.8 ms - the teacher costs 3.8 / 5.6 ms and sharing
# launch groups hides 0.6 of them.  ADVMIX_PAIR_TEACHER=0 restores the two separate forwards.
_PAIR_TEACHER = int(os.environ.get('ADVMIX_PAIR_TEACHER', '1'))


def _advance(runs):
    """One level of each in-flight forward (plan.PlanRun) as ONE launch group."""
    groups = [r.members() for r in runs]
    outs = ops.run_group([m for g in groups for m in g])
    pos = 0
    for r, g in zip(runs, groups):
        r.consume(outs[pos:pos + len(g)])
        pos += len(g)


def teacher_forward(model_teacher, clean):
    """function.py:148-149: the frozen teacher's heat-maps for the clean view, under no_grad, marked for the static filter
    images (plan.PlanNet._wino_refresh re-makes a frozen eval-mode network's Winograd images only when its filters change)."""
    getattr(model_teacher, 'module', model_teacher).wino_static = True
    for p_ in model_teacher.parameters():                                 # the teacher only ever runs under no_grad - nothing
        if p_.requires_grad:                                              # reads its gradients (checked per call: a state
            p_.requires_grad = False                                      # restore may re-arm the flags)
    with torch.no_grad():
        return model_teacher(clean).detach()


def advmix_phase_a1(args, model, model_G, model_teacher, optimizer, inputs, cuts=None, teacher_output=None):
    """First half of advmix_phase_a, function.py:137-149: G forward, softmax-mix, D forward on the detached mix and the
    teacher's forward.  Returns the state advmix_phase_a2 finishes the phase from.  ``teacher_output``: the teacher's
    heat-maps when the caller computes them itself (graph.AdvMixGraphRunner replays the teacher's forward as a graph of its
    own beside this half, ADVMIX_PAIR_TEACHER=2); otherwise the teacher runs here - its levels as members of the student's
    launch groups (1, the default), on a stream and lane set of its own from the start of the phase (2, eager), or after the
    student (0)."""
    G_input = ops.cat_views(inputs)                                       # :137
    getattr(model_teacher, 'module', model_teacher).wino_static = True    # (frozen, eval mode: see teacher_forward)
    pair = _PAIR_TEACHER if all(hasattr(m, 'begin') for m in (model, model_G, model_teacher)) else 0
    if teacher_output is not None:
        pair = -1
    if pair > 0:
        # the (frozen) teacher's levels as members of the student's launch groups: a sequential network is one chain per
        # pass, so the teacher shares the chip with it instead of running on its own afterwards
        for p_ in model_teacher.parameters():                             # function.py:148: the teacher only ever runs under
            if p_.requires_grad:                                          # no_grad - nothing reads its gradients (checked per
                p_.requires_grad = False                                  # call: a state restore may re-arm the flags)
    t_side = None
    if pair == 2 and inputs[0].is_cuda and not torch.cuda.is_current_stream_capturing():
        # eager form of mode 2: the teacher's whole forward on a stream and a lane set of its own, started BEFORE the
        # generator's - it needs the clean view only - so that it runs beside the generator's single-lane forward and on into
        # the student's; joined before the loss reads its heat-maps.  (NOT inside a capture: tensors allocated under a second
        # current stream while capturing raced / crashed hipStreamEndCapture - the graph runner replays a teacher graph instead.)
        cur = torch.cuda.current_stream(inputs[0].device)
        t_side = ops.aux_stream(inputs[0].device)
        t_side.wait_stream(cur)
        with torch.cuda.stream(t_side), ops.lane_set(1), torch.no_grad():
            teacher_output = model_teacher(inputs[0])                     # :148-149
    if cuts:
        model_G.cut_levels = tuple(c[0] for c in cuts[1])
    logits = model_G(G_input)                                             # :138 (softmax fused below)
    pairs_G = model_G.last_cuts if cuts else None
    model_G.cut_levels = ()
    set_require_grad(model, True)                                         # :140
    optimizer.zero_grad()
    tmp = ops.softmax_mix(logits, inputs)                                 # :138,142-144
    if cuts:
        model.cut_levels = tuple(c[0] for c in cuts[0])
    if t_side is not None:
        D_output_detach = model(tmp.detach())                             # :146
        torch.cuda.current_stream(inputs[0].device).wait_stream(t_side)
        teacher_output = teacher_output.detach()
    elif pair < 0:
        D_output_detach = model(tmp.detach())                             # :146 (the teacher's heat-maps come from the caller)
    elif pair:
        rt = model_teacher.begin(inputs[0])
        rd = model.begin(tmp.detach())                                    # :146
        while not rd.done:
            _advance([rd] if rt.done else [rd, rt])
        while not rt.done:
            _advance([rt])
        D_output_detach, teacher_output = rd.result, rt.result.detach()   # :148-149
    else:
        D_output_detach = model(tmp.detach())                             # :146
        with torch.no_grad():
            teacher_output = model_teacher(inputs[0])                     # :148-149
    pairs_D = model.last_cuts if cuts else None
    model.cut_levels = ()
    return {'tmp': tmp, 'out': D_output_detach, 'teacher': teacher_output, 'pairs_D': pairs_D, 'pairs_G': pairs_G}


def advmix_phase_a2(args, model, criterion, state, target, target_weight, cuts=None):
    """Second half of advmix_phase_a, function.py:151-154: the heat-map + distillation loss and D's backward pass."""
    loss_D = _blend(criterion, state['out'], target, 1 - args.alpha, state['teacher'], args.alpha, target_weight)   # :151-153
    if cuts:
        return loss_D.detach(), state['tmp'], _backward_pieces(loss_D, model, cuts[0], state['pairs_D']), state['pairs_G']
    _backward(loss_D)
    return loss_D.detach(), state['tmp']


def advmix_phase_a(args, model, model_G, model_teacher, criterion, optimizer, inputs, target, target_weight,
                   cuts=None):
    """function.py:137-154: G forward, softmax-mix, D forward on the detached mix, teacher forward,
    heat-map + KD loss, D backward.  Leaves D's gradients in ``optimizer.flat_grads``.
    ``cuts`` (data parallel) = (model.plan_cuts(), model_G.plan_cuts()): both forwards run with their autograd graph
    cut at those levels and the backward pass is NOT run here; it is returned as closures (see _backward_pieces) so
    that the caller can all-reduce each finished gradient range beside the next piece, together with G's cut record
    for advmix_phase_b."""
    state = advmix_phase_a1(args, model, model_G, model_teacher, optimizer, inputs, cuts)
    return advmix_phase_a2(args, model, criterion, state, target, target_weight, cuts)


def advmix_phase_b(args, model, criterion, optimizer, optimizer_G, tmp, target, target_weight, cuts_G=None,
                   model_G=None, pairs_G=None):
    """function.py:155-163: D update, then the adversarial pass through the frozen, updated D.
    ``cuts_G`` / ``pairs_G`` (data parallel): G's backward is returned in pieces like D's in advmix_phase_a; the first
    piece contains the whole input-gradient pass through D (whose graph is not cut in this phase)."""
    optimizer.step()                                                      # :155
    set_require_grad(model, False)                                        # :158
    optimizer_G.zero_grad()
    output = model(tmp)                                                   # :160
    loss_G = _blend(criterion, output, target, -args.adv_loss_weight, None, 0.0, target_weight)                     # :161
    if cuts_G is not None:
        return output.detach(), _backward_pieces(loss_G, model_G, cuts_G, pairs_G)
    _backward(loss_G)
    return output.detach()


def advmix_step(args, model, model_G, model_teacher, criterion, optimizer, optimizer_G,
                inputs, target, target_weight, grad_sync=None):
    """function.py:137-164, one batch.  ``inputs``: 3 contiguous NCHW fp32 CUDA views.
    Returns (loss_D, output) with output = D(tmp) after the D update (the tensor the reference
    feeds to ``accuracy``).  Data parallel (``grad_sync``): the two gradient exchanges sit exactly where the
    reference's optimizers consume the gradients, and each is OVERLAPPED with the backward pass that produces it - the
    backward runs in pieces, and the finished range of the flat gradient buffer is all-reduced on a side HIP stream
    while the next piece computes (dp.GradSync.reduce_async)."""
    if grad_sync is None or not grad_sync.active:
        loss_D, tmp = advmix_phase_a(args, model, model_G, model_teacher, criterion, optimizer,
                                     inputs, target, target_weight)
        output = advmix_phase_b(args, model, criterion, optimizer, optimizer_G, tmp, target, target_weight)
        optimizer_G.step()                                                # :164
        return loss_D, output
    cuts_D, cuts_G = grad_sync.cuts_for(model), grad_sync.cuts_for(model_G)
    loss_D, tmp, pieces, pairs_G = advmix_phase_a(args, model, model_G, model_teacher, criterion, optimizer,
                                                  inputs, target, target_weight, (cuts_D, cuts_G))
    for piece in pieces:
        lo, hi = piece()
        grad_sync.reduce_async(optimizer.flat_grads, lo, hi)
    grad_sync.finish()
    output, pieces = advmix_phase_b(args, model, criterion, optimizer, optimizer_G, tmp, target, target_weight,
                                    cuts_G, model_G, pairs_G)
    for piece in pieces:
        lo, hi = piece()
        grad_sync.reduce_async(optimizer_G.flat_grads, lo, hi)
    grad_sync.finish()
    optimizer_G.step()                                                    # :164
    return loss_D, output


# ---- the loops' fast path: HIP-graph replay + one-batch-ahead device prefetch ---------------------------------------
# The step has static shapes, so train_advmix / train capture it ONCE per (models, optimizers, batch shape) into HIP
# graphs (graph.AdvMixGraphRunner / PlainGraphRunner) and replay it per batch; a batch of another shape (the ragged
# last batch of an epoch) runs eagerly.  ADVMIX_EXEC=eager disables the capture.
GRAPH_EXEC = os.environ.get('ADVMIX_EXEC', 'graph') != 'eager'
# With more than one rank the step is replayed from seven HIP graphs (graph.AdvMixGraphRunner; ADVMIX_DP_GRAPH=0 falls back to
# the eager pieces).  Round 3 shipped the eager fallback after the two-ranks-on-one-GPU test began to fail; round 4 found the
# cause outside the step: replays served from the runtime's captured AQL packets on the NULL stream compute garbage when a
# second process shares the GPU (tools/dp_graph_repro.py) - the runners now replay on their own stream.
DP_GRAPH = os.environ.get('ADVMIX_DP_GRAPH', '1') == '1'


def _graph_ok(grad_sync):
    return GRAPH_EXEC and (DP_GRAPH or grad_sync is None or not grad_sync.active or grad_sync.world == 1)
_RUNNERS = {}


def _runner_for(kind, key_objs, sig, build):
    """One cached runner per set of live objects; ``sig`` = shapes and loss weights it was captured with.
    Returns None when a runner exists for other shapes (no second capture: that batch runs eagerly)."""
    key = (kind,) + tuple(id(o) for o in key_objs)
    hit = _RUNNERS.get(key)
    if hit is not None and all(a is b for a, b in zip(hit[2], key_objs)):
        return hit[1] if hit[0] == sig else None
    runner = build()
    _RUNNERS[key] = (sig, runner, tuple(key_objs))         # (the objects are kept alive with their runner)
    return runner


def release_graphs():
    """Drop every captured step (and the models / optimizers it keeps alive)."""
    _RUNNERS.clear()


class _Prefetch:
    """Iterate a loader one batch ahead.  While the step of batch i runs, the tensors of batch i+1 are copied to the
    device on a side HIP stream (asynchronously when the loader pins its memory, PIN_MEMORY in the experiment YAMLs);
    the consumer waits for the copy's event before it touches them.  ``pick(batch)`` names the tensors that go to the
    device; everything else (metas) passes through untouched."""

    def __init__(self, loader, pick):
        self.it, self.pick = iter(loader), pick
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.nxt = self._fetch()

    def _fetch(self):
        try:
            batch = next(self.it)
        except StopIteration:
            return None
        host = self.pick(batch)
        if self.stream is None or all(t.is_cuda for t in host):
            return batch, [t.float() if t.is_floating_point() else t for t in host], None
        with torch.cuda.stream(self.stream):
            dev = [t.cuda(non_blocking=True).float() for t in host]
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return batch, dev, ev

    def __iter__(self):
        return self

    def __next__(self):
        cur = self.nxt
        if cur is None:
            raise StopIteration
        self.nxt = self._fetch()                           # batch i+1 goes in flight before batch i is consumed
        batch, dev, ev = cur
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            for t in dev:
                t.record_stream(torch.cuda.current_stream())
        return batch, dev


class _Deferred:
    """``loss.item()`` and ``accuracy()`` of iteration i (function.py:62-66 / :167-169), read while iteration i + 1 is
    already running on the GPU (evaluate.PendingAccuracy); ``flush()`` before anything prints the meters, so a log line
    shows exactly what the reference's shows."""

    def __init__(self, losses, acc):
        self.losses, self.acc, self.pending = losses, acc, None

    def push(self, loss, output, target, bs):
        if torch.is_tensor(output) and output.is_cuda and torch.is_tensor(target) and target.is_cuda:
            from .evaluate import PendingAccuracy
            cur = (PendingAccuracy(output, target, loss), bs)      # enqueued behind this iteration's step
            self.flush()                                           # the previous iteration's numbers: long since there
            self.pending = cur
        else:
            self.flush()
            self.losses.update(loss.item(), bs)
            _, avg_acc, cnt, _ = accuracy(output, target)
            self.acc.update(avg_acc, cnt)

    def flush(self):
        if self.pending is not None:
            (p, bs), self.pending = self.pending, None
            _, avg_acc, cnt, _, lv = p.get()
            self.losses.update(lv, bs)
            self.acc.update(avg_acc, cnt)


def _log(config, epoch, i, n, batch_time, data_time, losses, acc, bs, writer_dict):
    msg = 'Epoch: [{0}][{1}/{2}]\t' \
          'Time {batch_time.val:.3f}s ({batch_time.avg:.3f}s)\t' \
          'Speed {speed:.1f} samples/s\t' \
          'Data {data_time.val:.3f}s ({data_time.avg:.3f}s)\t' \
          'Loss {loss.val:.5f} ({loss.avg:.5f})\t' \
          'Accuracy {acc.val:.3f} ({acc.avg:.3f})'.format(
              epoch, i, n, batch_time=batch_time, speed=bs / batch_time.val,
              data_time=data_time, loss=losses, acc=acc)
    logger.info(msg)
    if writer_dict:
        writer = writer_dict['writer']
        global_steps = writer_dict['train_global_steps']
        writer.add_scalar('train_loss', losses.val, global_steps)
        writer.add_scalar('train_acc', acc.val, global_steps)
        writer_dict['train_global_steps'] = global_steps + 1


_LOOP_STREAMS = {}


class _loop_stream:
    """The training loops run on a CREATED stream, never on the NULL stream: graph replays, gradient exchanges, the
    replica checks and the metrics read-back all inherit it as their current stream.  Round 4: a
    collective issued while the NULL stream was current made later graph replays compute garbage on a GPU shared by two ranks
    (gloo), and the NULL stream is also the one stream every blocking stream of the process synchronises with implicitly -
    there is nothing to gain from it.  No-op on the CPU or when the caller already runs on a stream of its own."""

    def __enter__(self):
        self.ctx = None
        if torch.cuda.is_available():
            dev = torch.cuda.current_device()
            cur = torch.cuda.current_stream(dev)
            if cur.cuda_stream == 0:
                st = _LOOP_STREAMS.get(dev)
                if st is None:
                    st = _LOOP_STREAMS[dev] = torch.cuda.Stream(device=dev)
                st.wait_stream(cur)
                self.cur, self.st = cur, st
                self.ctx = torch.cuda.stream(st)
                self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.cur.wait_stream(self.st)
        return False


_AUTO_SYNC = {}


def _auto_sync(models, optimizers, grad_sync):
    """The reference's call sites (tools/train.py:291-296) pass no ``grad_sync``: when the process is one rank of an
    initialised process group (one process per GPU instead of nn.DataParallel), the loops create the GradSync themselves
    - once per set of models - and make every replica start from rank 0's state, which is what DataParallel's
    per-forward weight broadcast did implicitly."""
    import torch.distributed as dist
    if grad_sync is not None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return grad_sync
    key = tuple(id(m) for m in models)
    hit = _AUTO_SYNC.get(key)
    if hit is None or any(a is not b for a, b in zip(hit[1], models)):
        from ..dp import GradSync
        gs = GradSync()
        gs.broadcast_state(list(models), list(optimizers))
        hit = _AUTO_SYNC[key] = (gs, tuple(models))
    return hit[0]


def _replica_check(grad_sync, optimizers, epoch, i):
    """Every PRINT_FREQ iterations with more than one rank: the replicas' parameters and optimizer state must still be the
    same bits and finite on every rank (one 24-byte all-gather).  nn.DataParallel re-broadcast GPU 0's weights before
    every forward (lib/core/function.py:138,146,160 through tools/train.py:69,106,109), so the reference could not drift;
    separate processes can, silently, and a log line must not be printed over a job that has."""
    if grad_sync is not None and grad_sync.active and grad_sync.world > 1:
        grad_sync.assert_replicas(optimizers, 'epoch %d iteration %d' % (epoch, i))


def _net(m):
    """The network inside a DataParallel-shaped wrapper (dp.Replica), or ``m`` itself."""
    from ..dp import unwrap
    return unwrap(m)


def _capturable(optimizers, tensors):
    """The graph runners need the flat optimizers of get_optimizer (flat_state / flat_grads / sync_hyper) and batches
    that are already on the GPU; a plain torch.optim optimizer handed to the drop-in signature takes the eager step."""
    return all(hasattr(o, 'flat_state') and hasattr(o, 'sync_hyper') for o in optimizers) \
        and all(torch.is_tensor(t) and t.is_cuda for t in tensors)


def _rank_loader(train_loader, epoch):
    """This rank's view of a training loader for ``epoch`` (dp.ShardedDataLoader: the epoch's permutation - a resumed run
    continues where it stopped - and a sharded twin of an unshuffled loader); any other loader is returned as it is."""
    if hasattr(train_loader, 'for_training'):
        train_loader = train_loader.for_training()
    if hasattr(train_loader, 'set_epoch'):
        train_loader.set_epoch(epoch)
    elif hasattr(getattr(train_loader, 'sampler', None), 'set_epoch'):
        train_loader.sampler.set_epoch(epoch)
    return train_loader


def train(config, args, train_loader, model, criterion, optimizer, epoch,
          output_dir, tb_log_dir, writer_dict, grad_sync=None):
    batch_time, data_time, losses, acc = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    if isinstance(model, list):
        model = model[0]
    model = _net(model)
    model.train()
    grad_sync = _auto_sync([model], [optimizer], grad_sync)
    meters = _Deferred(losses, acc)
    end = time.time()
    train_loader = _rank_loader(train_loader, epoch)
    n = len(train_loader) if hasattr(train_loader, '__len__') else -1
    pick = lambda b: [b[0], b[1][0] if isinstance(b[1], (list, tuple)) else b[1], b[2]]     # noqa: E731  (:48-51)
    with _loop_stream():
        _train_loop(config, train_loader, pick, model, criterion, optimizer, epoch, grad_sync, meters, batch_time, data_time,
                    losses, acc, end, n, writer_dict)


def _train_loop(config, train_loader, pick, model, criterion, optimizer, epoch, grad_sync, meters, batch_time, data_time,
                losses, acc, end, n, writer_dict):
    for i, ((input, _t, _w, meta), (x, target, target_weight)) in enumerate(_Prefetch(train_loader, pick)):
        data_time.update(time.time() - end)
        x = x.contiguous()
        runner = None
        if _graph_ok(grad_sync) and _capturable([optimizer], [x, target, target_weight]):
            from ..graph import PlainGraphRunner
            sig = (tuple(x.shape), tuple(target.shape), tuple(target_weight.shape), id(grad_sync))
            runner = _runner_for('plain', (model, criterion, optimizer), sig, lambda: PlainGraphRunner(
                model, criterion, optimizer, x, target, target_weight, grad_sync))
        if runner is not None:
            runner.load_batch(x, target, target_weight)
            loss, outputs = runner.step()
        else:
            loss, outputs = plain_step(model, criterion, optimizer, x, target, target_weight, grad_sync)
        meters.push(loss, outputs, target, input.size(0))                                # :62-66, one iteration late
        batch_time.update(time.time() - end)
        end = time.time()
        if i % config.PRINT_FREQ == 0:
            meters.flush()
            _replica_check(grad_sync, [optimizer], epoch, i)
            _log(config, epoch, i, n, batch_time, data_time, losses, acc, input.size(0), writer_dict)
    meters.flush()


def train_advmix(config, args, train_loader, models, criterion, optimizers, epoch,
                 output_dir, tb_log_dir, writer_dict, grad_sync=None):
    batch_time, data_time, losses, acc = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    model = _net(models[0]).train()
    model_G = _net(models[1]).train()
    model_teacher = _net(models[2]).eval()
    optimizer, optimizer_G = optimizers[0], optimizers[1]
    grad_sync = _auto_sync([model, model_G, model_teacher], [optimizer, optimizer_G], grad_sync)
    meters = _Deferred(losses, acc)
    end = time.time()
    train_loader = _rank_loader(train_loader, epoch)
    n = len(train_loader) if hasattr(train_loader, '__len__') else -1
    pick = lambda b: [b[0][0], b[0][1], b[0][2], b[1][0], b[2][0]]                        # noqa: E731  (:129-133)
    with _loop_stream():
        _advmix_loop(config, args, train_loader, pick, model, model_G, model_teacher, criterion, optimizer, optimizer_G, epoch,
                     grad_sync, meters, batch_time, data_time, losses, acc, end, n, writer_dict)


def _advmix_loop(config, args, train_loader, pick, model, model_G, model_teacher, criterion, optimizer, optimizer_G, epoch,
                 grad_sync, meters, batch_time, data_time, losses, acc, end, n, writer_dict):
    for i, (_batch, dev) in enumerate(_Prefetch(train_loader, pick)):
        data_time.update(time.time() - end)
        inputs = [v.contiguous() for v in dev[:3]]
        target, target_weight = dev[3], dev[4]
        runner = None
        if _graph_ok(grad_sync) and _capturable([optimizer, optimizer_G], inputs + [target, target_weight]):
            from ..graph import AdvMixGraphRunner
            sig = (tuple(inputs[0].shape), tuple(target.shape), tuple(target_weight.shape), float(args.alpha),
                   float(args.adv_loss_weight), id(grad_sync))
            runner = _runner_for('advmix', (model, model_G, model_teacher, criterion, optimizer, optimizer_G), sig,
                                 lambda: AdvMixGraphRunner(args, model, model_G, model_teacher, criterion, optimizer,
                                                           optimizer_G, inputs, target, target_weight, grad_sync))
        if runner is not None:                             # replay the captured step on this batch
            runner.load_batch(inputs, target, target_weight)
            loss_D, output = runner.step()
        else:                                              # another batch shape: eager
            loss_D, output = advmix_step(args, model, model_G, model_teacher, criterion, optimizer,
                                         optimizer_G, inputs, target, target_weight, grad_sync)
        meters.push(loss_D, output, target, inputs[0].size(0))                           # :167-169, one iteration late
        batch_time.update(time.time() - end)
        end = time.time()
        if i % config.PRINT_FREQ == 0:
            meters.flush()
            _replica_check(grad_sync, [optimizer, optimizer_G], epoch, i)
            _log(config, epoch, i, n, batch_time, data_time, losses, acc, inputs[0].size(0), writer_dict)
    meters.flush()


def validate_batch(config, model, criterion, input, target, target_weight, flip_pairs):
    """function.py:223-276 for one batch, entirely on the device: eval forward, optional flip test
    (flipped forward, flip-back + joint swap + SHIFT_HEATMAP + average as ONE kernel), loss.
    Returns (output [B,J,H,W] CUDA tensor, loss 0-dim tensor)."""
    with torch.no_grad():
        input = _cuda(input).float().contiguous()
        output = model(input)                                                            # :230
        if isinstance(output, list):
            output = output[-1]
        if config.TEST.FLIP_TEST:
            output_flipped = model(ops.flip_w(input))                                    # :241-242
            if isinstance(output_flipped, list):
                output_flipped = output_flipped[-1]
            output = ops.flip_merge(output, output_flipped, flip_pairs,
                                    shift=bool(config.TEST.SHIFT_HEATMAP))               # :249-261
        loss = criterion(output, target, target_weight)                                  # :269
    return output, loss


def validate(config, args, val_loader, val_dataset, model, criterion, output_dir,
             tb_log_dir, writer_dict=None, cpu=False):
    """function.py:200-358.  Same bookkeeping (all_preds [N,J,3], all_boxes [N,6], image paths, the
    ``val_dataset.evaluate`` call, markdown table, tensorboard scalars); the heat-maps never leave the
    GPU - per batch only the loss scalar, [B,J] argmax indices (accuracy) and [B,J,3] predictions do."""
    if cpu:
        raise RuntimeError('advmix_amd: validate() has no CPU path (the reference\'s cpu=True debug mode)')
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    # One process per GPU (round 5): nn.DataParallel scattered every validation batch over all GPUs
    # (tools/train.py:106,300); here rank r evaluates batches r, r + world, ... of the SAME loader order, the ranks' rows
    # meet on rank 0 (gather_object: [B,J,3] predictions, boxes, paths - never heat-maps), rank 0 alone runs
    # ``val_dataset.evaluate`` and writes the results, and the others wait for it on the HOST (a key in the process group's
    # store, hours of timeout) - not inside the next epoch's first all-reduce, where NCCL's watchdog would abort the job
    # when COCOeval on one rank takes longer than its default ten minutes.
    loader, mine = _val_shard(val_loader, rank, world)
    batch_time, losses, acc = AverageMeter(), AverageMeter(), AverageMeter()
    model = _net(model)
    model.eval()
    num_samples = len(val_dataset)
    all_preds = np.zeros((num_samples, config.MODEL.NUM_JOINTS, 3), dtype=np.float32)
    all_boxes = np.zeros((num_samples, 6))
    image_path, filenames, imgnums = [], [], []
    rows = []                                               # [(first row, image paths)] of the batches this rank evaluated
    idx = 0
    end = time.time()
    time_gpu = 0.
    n_batches = 0
    for i, (input, target, target_weight, meta) in enumerate(loader):
        num_images = input.size(0)
        if mine is None:                                    # the loader could not be re-sharded: every rank walks all of it
            if i % world != rank:                           # and computes its own batches only
                idx += num_images
                continue
        else:
            idx = mine[i]
        if isinstance(target, (list, tuple)):
            target = target[0]                                                           # :264
        target = _cuda(target)
        target_weight = _cuda(target_weight)
        torch.cuda.synchronize()
        infer_start = time.time()
        output, loss = validate_batch(config, model, criterion, input, target, target_weight,
                                      val_dataset.flip_pairs)
        torch.cuda.synchronize()
        time_gpu += time.time() - infer_start
        losses.update(loss.item(), num_images)
        _, avg_acc, cnt, pred = accuracy(output, target, args=None, cfg=config)          # :274
        acc.update(avg_acc, cnt)
        batch_time.update(time.time() - end)
        end = time.time()

        c = np.asarray(meta['center'], dtype=np.float32) if not torch.is_tensor(meta['center']) \
            else meta['center'].numpy()
        s = np.asarray(meta['scale'], dtype=np.float32) if not torch.is_tensor(meta['scale']) \
            else meta['scale'].numpy()
        score = np.asarray(meta['score']) if not torch.is_tensor(meta['score']) else meta['score'].numpy()
        preds, maxvals = get_final_preds(config, args, output, c, s)                     # :286-287
        all_preds[idx:idx + num_images, :, 0:2] = preds[:, :, 0:2]
        all_preds[idx:idx + num_images, :, 2:3] = maxvals
        all_boxes[idx:idx + num_images, 0:2] = c[:, 0:2]
        all_boxes[idx:idx + num_images, 2:4] = s[:, 0:2]
        all_boxes[idx:idx + num_images, 4] = np.prod(s * 200, 1)
        all_boxes[idx:idx + num_images, 5] = score
        rows.append((idx, list(meta['image'])))
        idx += num_images
        n_batches += 1
        if i % config.PRINT_FREQ == 0:
            logger.info('Test: [{0}/{1}]\t'
                        'Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t'
                        'Loss {loss.val:.4f} ({loss.avg:.4f})\t'
                        'Accuracy {acc.val:.3f} ({acc.avg:.3f})'.format(
                            i, len(loader) if hasattr(loader, '__len__') else -1,
                            batch_time=batch_time, loss=losses, acc=acc))
    logger.info('=> The average inference time is : %s', time_gpu / max(n_batches, 1))

    if world > 1:
        part = {'rows': [(r0, all_preds[r0:r0 + len(pths)], all_boxes[r0:r0 + len(pths)], pths) for r0, pths in rows],
                'loss': (losses.sum, losses.count), 'acc': (acc.sum, acc.count)}
        parts = [None] * world if rank == 0 else None
        dist.gather_object(part, parts, dst=0)
        validate.calls = getattr(validate, 'calls', 0) + 1
        key = 'advmix_validate_done_%d' % validate.calls
        store = dist.distributed_c10d._get_default_store()
        if rank != 0:
            import datetime
            store.wait([key], datetime.timedelta(hours=int(os.environ.get('ADVMIX_VALIDATE_WAIT_HOURS', '12'))))
            perf = float(store.get(key).decode())
            validate.last = {'loss': losses.avg, 'acc': acc.avg}
            return {}, perf
        losses, acc = AverageMeter(), AverageMeter()
        for prt in parts:
            for r0, pr, bx, pths in prt['rows']:
                all_preds[r0:r0 + len(pths)] = pr
                all_boxes[r0:r0 + len(pths)] = bx
            for m, k in ((losses, 'loss'), (acc, 'acc')):
                m.sum += prt[k][0]
                m.count += prt[k][1]
                m.avg = m.sum / m.count if m.count else 0
        rows = sorted((r for prt in parts for r in ((r0, pths) for r0, _p, _b, pths in prt['rows'])), key=lambda r: r[0])
    for _r0, pths in rows:
        image_path.extend(pths)

    try:
        name_values, perf_indicator = val_dataset.evaluate(
            config, all_preds, output_dir, all_boxes, image_path, filenames, imgnums)
        model_name = config.MODEL.NAME
        for name_value in (name_values if isinstance(name_values, list) else [name_values]):
            _print_name_value(name_value, model_name)
        if writer_dict:
            writer = writer_dict['writer']
            global_steps = writer_dict['valid_global_steps']
            writer.add_scalar('valid_loss', losses.avg, global_steps)
            writer.add_scalar('valid_acc', acc.avg, global_steps)
            for name_value in (name_values if isinstance(name_values, list) else [name_values]):
                writer.add_scalars('valid', dict(name_value), global_steps)
            writer_dict['valid_global_steps'] = global_steps + 1
    except BaseException:
        if world > 1:                                       # never leave the other ranks waiting for a rank that has failed
            store.set(key, 'nan')
        raise
    if world > 1:
        store.set(key, repr(float(perf_indicator)))
    validate.last = {'loss': losses.avg, 'acc': acc.avg}        # for callers that want the meters
    return name_values, perf_indicator


def _val_shard(val_loader, rank, world):
    """(loader, first rows): this rank's share of a validation loader - batches rank, rank + world, ... in the loader's own
    order, with the row of ``all_preds`` each starts at - as a DataLoader over the same dataset / collate function /
    workers.  (val_loader, None) when there is one rank, or when the loader cannot be re-sharded (no batch sampler to
    enumerate): validate() then walks the whole loader on every rank and computes every world-th batch."""
    if world <= 1:
        return val_loader, None
    bs, ds = getattr(val_loader, 'batch_sampler', None), getattr(val_loader, 'dataset', None)
    if bs is None or ds is None:
        return val_loader, None
    try:
        batches = [list(b) for b in bs]
    except TypeError:
        return val_loader, None
    first, r0 = [], 0
    for b in batches:
        first.append(r0)
        r0 += len(b)
    pick = range(rank, len(batches), world)
    kw = {k: getattr(val_loader, k) for k in ('num_workers', 'collate_fn', 'pin_memory') if hasattr(val_loader, k)}
    loader = torch.utils.data.DataLoader(ds, batch_sampler=[batches[i] for i in pick], **kw)
    return loader, [first[i] for i in pick]


def _print_name_value(name_value, full_arch_name):
    """function.py:363-380: one markdown table row per metric set."""
    names, values = list(name_value.keys()), list(name_value.values())
    logger.info('| Arch ' + ' '.join('| {}'.format(n) for n in names) + ' |')
    logger.info('|---' * (len(names) + 1) + '|')
    if len(full_arch_name) > 15:
        full_arch_name = full_arch_name[:8] + '...'
    logger.info('| ' + full_arch_name + ' ' + ' '.join('| {:.3f}'.format(v) for v in values) + ' |')


class AverageMeter(object):
    """Computes and stores the average and current value"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count if self.count != 0 else 0
