"""HIP-graph execution of the AdvMix step.

HRNet-W32's step is ~9,000 kernel launches (293 convs + 292 BNs per pass, 3 forward and 2
backward passes); driven eagerly from Python it is launch-bound.  The step has static shapes,
so it is captured ONCE into HIP graphs (hipStreamBeginCapture via torch.cuda.CUDAGraph - every
kernel in libadvmix_hip.so is capture-safe: no allocation, no sync) and replayed per batch.

Captured through ops.GraphSeq: one HIP graph with parallel branches per segment (the launch lanes' streams fork and
join inside the capture).
Single GPU: two segments (phase a: G fwd ... D bwd; phase b: Adam(D) ... G bwd, Adam(G)).
With data parallelism the step is cut into seven graphs, one per piece of a backward pass (core.function.
_backward_pieces): [G fwd, mix, D fwd, T fwd, losses, top third of D's bwd] | [middle third] | [bottom third] |
[Adam(D), D fwd, loss, bwd through D, top third of G's bwd] | [middle] | [bottom] | [Adam(G)].  After each piece the
range of the flat gradient buffer it completed is all-reduced on a side HIP stream (dp.GradSync.reduce_async) while
the next graph runs; the RCCL calls stay outside the graphs.  The autograd tape recorded while capturing one graph is
consumed while capturing the next; all share one memory pool."""
import torch

from . import ops
from .core.function import (advmix_phase_a, advmix_phase_a1, advmix_phase_a2, advmix_phase_b, advmix_step, plain_step,
                            teacher_forward, _PAIR_TEACHER)


def _snapshot(models, optimizers):
    """Clones of everything a training step mutates: each optimizer's flat parameters / Adam moments / step counter
    / gradient buffer, and every module buffer (BatchNorm running_mean, running_var, num_batches_tracked)."""
    pairs = []
    for opt in optimizers:
        if opt is not None:
            pairs += [(t, t.clone()) for t in opt.flat_state() + [opt.flat_grads]]
    for m in models:
        if m is not None:
            pairs += [(b, b.clone()) for b in m.buffers()]
    flags = [(p, p.requires_grad) for m in models if m is not None for p in m.parameters()]
    return pairs, flags


def _restore(snap):
    pairs, flags = snap
    with torch.no_grad():
        for t, c in pairs:
            t.copy_(c)
    for p, f in flags:
        p.requires_grad = f


class AdvMixGraphRunner:
    def __init__(self, args, model, model_G, model_teacher, criterion, optimizer, optimizer_G,
                 inputs, target, target_weight, grad_sync=None, warmup=2):
        self.opt, self.optG, self.sync = optimizer, optimizer_G, grad_sync
        dev = inputs[0].device
        # static input buffers: new batches are copied into these
        self.inputs = [v.clone() for v in inputs]
        self.target, self.tw = target.clone(), target_weight.clone()
        # The eager warm-up (allocator pool, lazy workspaces, RCCL communicators) runs REAL steps on the live
        # models: without a snapshot D and G would take ``warmup`` extra Adam updates on batch 0 and the BatchNorm
        # running statistics 2 x warmup extra momentum updates before the first replayed batch - a different
        # trajectory from the reference loop (function.py:129-171).  Everything a step mutates is restored.
        snap = _snapshot([model, model_G, model_teacher], [optimizer, optimizer_G])
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                     # eager warm-up off the default stream
            for _ in range(warmup):
                advmix_step(args, model, model_G, model_teacher, criterion, optimizer, optimizer_G,
                            self.inputs, self.target, self.tw, grad_sync)
        torch.cuda.current_stream(dev).wait_stream(side)
        _restore(snap)
        torch.cuda.synchronize(dev)
        optimizer.sync_hyper()
        optimizer_G.sync_hyper()
        self.segments = []          # [(segment id, None | (optimizer, lo, hi) to all-reduce after it, finish-before flag)]
        self.seq = ops.GraphSeq(dev)

        def capture(fn):
            return self.seq.capture(fn)

        sync = grad_sync if (grad_sync is not None and grad_sync.active) else None
        self.tseq = None
        if sync is None and _PAIR_TEACHER == 2:
            # ADVMIX_PAIR_TEACHER=2 (round 6): the frozen teacher's forward is a HIP graph of its OWN (own memory pool, own lane
            # set: its kernels run beside the step's), replayed on a side stream at the start of the step - it needs the clean
            # view only - beside the generator's single-lane forward and the student's; phase a is captured in two halves and
            # the loss half waits for it.
            self.tseq = ops.GraphSeq(dev)
            self.tside = ops.aux_stream(dev)
            with ops.lane_set(1):
                teacher_forward(model_teacher, self.inputs[0])        # (eager once: lane set 1's streams and scratch exist)
                torch.cuda.synchronize(dev)
                box_t = {}

                def seg_t():
                    box_t['out'] = teacher_forward(model_teacher, self.inputs[0])
                self.tseg, _ = self.tseq.capture(seg_t)
            self._tout = box_t['out']
            box = {}

            def seg_a1():
                box['state'] = advmix_phase_a1(args, model, model_G, model_teacher, optimizer, self.inputs, None, self._tout)
            g, _ = capture(seg_a1)
            self.segments.append((g, None))

            def seg_a2():
                self.loss_D, self._tmp = advmix_phase_a2(args, model, criterion, box['state'], self.target, self.tw)
                box.clear()
            g, _ = capture(seg_a2)
            self.segments.append((g, None, False, True))   # (4th field: wait for the teacher's graph before this segment)
        elif sync is None:
            def seg_a():
                self.loss_D, self._tmp = advmix_phase_a(args, model, model_G, model_teacher, criterion, optimizer,
                                                        self.inputs, self.target, self.tw)
            g, _ = capture(seg_a)
            self.segments.append((g, None))

        if sync is None:
            def seg_b():
                self.output = advmix_phase_b(args, model, criterion, optimizer, optimizer_G, self._tmp,
                                             self.target, self.tw)
                optimizer_G.step(sync_hyper=False)
            g, _ = capture(seg_b)
            self.segments.append((g, None))
        else:
            # Data parallel: the step is cut where a range of a flat gradient buffer is complete (backward pieces of D,
            # then of G).  After each segment's replay the range is all-reduced on the side stream (RCCL stays outside
            # the graphs) while the NEXT segment - the next piece of the backward pass - already runs; the segment that
            # holds the optimizer step is replayed after ``finish()``.
            cuts_D, cuts_G = sync.cuts_for(model), sync.cuts_for(model_G)
            box = {}

            def seg_a0():
                self.loss_D, self._tmp, box['pieces'], box['pairs_G'] = advmix_phase_a(
                    args, model, model_G, model_teacher, criterion, optimizer, self.inputs, self.target, self.tw,
                    (cuts_D, cuts_G))
                return box['pieces'][0]()
            g, rng = capture(seg_a0)
            self.segments.append((g, (optimizer,) + tuple(rng)))
            for piece in box['pieces'][1:]:
                g, rng = capture(piece)
                self.segments.append((g, (optimizer,) + tuple(rng)))

            def seg_b0():
                self.output, box['pieces'] = advmix_phase_b(args, model, criterion, optimizer, optimizer_G, self._tmp,
                                                            self.target, self.tw, cuts_G, model_G, box['pairs_G'])
                return box['pieces'][0]()
            g, rng = capture(seg_b0)                       # starts with Adam(D): replayed after finish()
            self.segments.append((g, (optimizer_G,) + tuple(rng), True))
            for piece in box['pieces'][1:]:
                g, rng = capture(piece)
                self.segments.append((g, (optimizer_G,) + tuple(rng)))
            g, _ = capture(lambda: optimizer_G.step(sync_hyper=False))
            self.segments.append((g, None, True))
        self._tmp = None
        torch.cuda.synchronize(dev)

    def load_batch(self, inputs, target, target_weight):
        for d, s in zip(self.inputs, inputs):
            d.copy_(s, non_blocking=True)
        self.target.copy_(target, non_blocking=True)
        self.tw.copy_(target_weight, non_blocking=True)

    def step(self):
        """Replay one AdvMix step on the current static batch. Returns (loss_D, output) views
        of graph-owned tensors (valid until the next replay).  The replays and the exchanges between them run on the
        runner's own stream when the caller's current stream is the NULL stream (ops.GraphSeq.replay_stream); the caller's
        stream waits for the step, so the results can be consumed on it as before."""
        self.opt.sync_hyper()
        self.optG.sync_hyper()
        cur = torch.cuda.current_stream(self.seq.device)
        rs, hop = self.seq.replay_stream()
        if hop:
            rs.wait_stream(cur)                            # load_batch's copies, the previous step's consumers
        with torch.cuda.stream(rs):
            if self.tseq is not None:                      # the teacher's graph beside the first half of phase a
                self.tside.wait_stream(rs)
                with torch.cuda.stream(self.tside):
                    self.tseq.graphs[self.tseg].replay()
            for seg in self.segments:
                g, red = seg[0], seg[1]
                if len(seg) > 2 and seg[2] and self.sync is not None:
                    self.sync.finish()                     # this segment's optimizer step consumes reduced gradients
                if len(seg) > 3 and seg[3]:
                    rs.wait_stream(self.tside)             # the loss reads the teacher's heat-maps
                self.seq.replay(g)
                if red is not None and self.sync is not None:
                    self.sync.reduce_async(red[0].flat_grads, red[1], red[2])
        if hop:
            cur.wait_stream(rs)
        return self.loss_D, self.output


class PlainGraphRunner:
    """The plain (non-AdvMix) step of ``train`` (lib/core/function.py:48-59) as HIP graphs: [forward, loss, zero_grad,
    backward] -> all-reduce (outside the graphs) -> [Adam]."""

    def __init__(self, model, criterion, optimizer, input, target, target_weight, grad_sync=None, warmup=2):
        self.opt, self.sync = optimizer, grad_sync
        dev = input.device
        self.input, self.target, self.tw = input.clone(), target.clone(), target_weight.clone()
        snap = _snapshot([model], [optimizer])
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                plain_step(model, criterion, optimizer, self.input, self.target, self.tw, grad_sync)
        torch.cuda.current_stream(dev).wait_stream(side)
        _restore(snap)
        torch.cuda.synchronize(dev)
        optimizer.sync_hyper()
        self.seq = ops.GraphSeq(dev)

        def seg_a():
            outputs = model(self.input)
            loss = criterion(outputs, self.target, self.tw)
            optimizer.zero_grad()
            loss.backward(ops.unit_grad(loss.device))       # (the constant seed: core.function._backward)
            self.loss, self.output = loss.detach(), outputs.detach()
        self.s1, _ = self.seq.capture(seg_a)
        self.s2, _ = self.seq.capture(lambda: optimizer.step(sync_hyper=False))
        torch.cuda.synchronize(dev)

    def load_batch(self, input, target, target_weight):
        self.input.copy_(input, non_blocking=True)
        self.target.copy_(target, non_blocking=True)
        self.tw.copy_(target_weight, non_blocking=True)

    def step(self):
        self.opt.sync_hyper()
        cur = torch.cuda.current_stream(self.seq.device)
        rs, hop = self.seq.replay_stream()
        if hop:
            rs.wait_stream(cur)
        with torch.cuda.stream(rs):                        # never the NULL stream (ops.GraphSeq)
            self.seq.replay(self.s1)
            if self.sync is not None:
                self.sync.sync(self.opt)
            self.seq.replay(self.s2)
        if hop:
            cur.wait_stream(rs)
        return self.loss, self.output
