"""HIP-graph execution of the AdvMix step.

HRNet-W32's step is ~9,000 kernel launches (293 convs + 292 BNs per pass, 3 forward and 2
backward passes); driven eagerly from Python it is launch-bound.  The step has static shapes,
so it is captured ONCE into HIP graphs (hipStreamBeginCapture via torch.cuda.CUDAGraph - every
kernel in libadvmix_hip.so is capture-safe: no allocation, no sync) and replayed per batch.

With data parallelism the step is cut into three graphs at the two points where gradients are
exchanged (see dp.GradSync): [G fwd, mix, D fwd, T fwd, losses, D bwd] -> all-reduce(D grads)
-> [Adam(D), D fwd, loss, bwd through D into G] -> all-reduce(G grads) -> [Adam(G)].
The RCCL calls stay outside the graphs.  The autograd tape recorded while capturing graph 1
is consumed while capturing graph 2; all three share one memory pool."""
import torch

from .core.function import advmix_phase_a, advmix_phase_b, advmix_step, plain_step


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
        self.g1, self.g2, self.g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread_local: only this thread launches work; RCCL's watchdog thread may query events meanwhile
        mode = dict(capture_error_mode='thread_local')
        with torch.cuda.graph(self.g1, **mode):
            self.loss_D, tmp = advmix_phase_a(args, model, model_G, model_teacher, criterion, optimizer,
                                              self.inputs, self.target, self.tw)
        pool = self.g1.pool()
        with torch.cuda.graph(self.g2, pool=pool, **mode):
            self.output = advmix_phase_b(args, model, criterion, optimizer, optimizer_G, tmp,
                                         self.target, self.tw)
        del tmp
        with torch.cuda.graph(self.g3, pool=pool, **mode):
            optimizer_G.step(sync_hyper=False)
        torch.cuda.synchronize(dev)

    def load_batch(self, inputs, target, target_weight):
        for d, s in zip(self.inputs, inputs):
            d.copy_(s, non_blocking=True)
        self.target.copy_(target, non_blocking=True)
        self.tw.copy_(target_weight, non_blocking=True)

    def step(self):
        """Replay one AdvMix step on the current static batch. Returns (loss_D, output) views
        of graph-owned tensors (valid until the next replay)."""
        self.opt.sync_hyper()
        self.optG.sync_hyper()
        self.g1.replay()
        if self.sync is not None:
            self.sync.sync(self.opt)
        self.g2.replay()
        if self.sync is not None:
            self.sync.sync(self.optG)
        self.g3.replay()
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
        self.g1, self.g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        mode = dict(capture_error_mode='thread_local')
        with torch.cuda.graph(self.g1, **mode):
            outputs = model(self.input)
            loss = criterion(outputs, self.target, self.tw)
            optimizer.zero_grad()
            loss.backward()
            self.loss, self.output = loss.detach(), outputs.detach()
        with torch.cuda.graph(self.g2, pool=self.g1.pool(), **mode):
            optimizer.step(sync_hyper=False)
        torch.cuda.synchronize(dev)

    def load_batch(self, input, target, target_weight):
        self.input.copy_(input, non_blocking=True)
        self.target.copy_(target, non_blocking=True)
        self.tw.copy_(target_weight, non_blocking=True)

    def step(self):
        self.opt.sync_hyper()
        self.g1.replay()
        if self.sync is not None:
            self.sync.sync(self.opt)
        self.g2.replay()
        return self.loss, self.output
