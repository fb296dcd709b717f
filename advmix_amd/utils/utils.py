"""Mirror of the hot-path pieces of lib/utils/utils.py: ``get_optimizer`` (:78-94, Adam with
lr only) backed by ONE flat-buffer HIP Adam kernel instead of 878 per-tensor launches, and
``save_checkpoint`` (:97-108)."""
import ctypes
import os

import torch

from .._lib import call


def _ceil4(n):
    return (n + 3) // 4 * 4


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr) semantics (betas (0.9, 0.999), eps 1e-8, no weight decay).

    On first use every parameter is re-pointed into one flat fp32 buffer (16-byte aligned
    slots, conv weights keep their channels_last strides) and ``p.grad`` into a parallel flat
    gradient buffer that the backward kernels accumulate into.  ``step()`` is a single kernel
    over the flat buffers; lr / betas / eps and the step counter live in device memory so a
    captured HIP graph replays correctly.  ``zero_grad()`` zero-fills (never sets to None)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self._flat = None
        if all(p.is_cuda for p in self._params()):
            self._ensure_flat()

    # ---- flat storage ---------------------------------------------------------------------
    def _params(self):
        return [p for g in self.param_groups for p in g['params']]

    @staticmethod
    def _view(flat, off, p):
        n = p.numel()
        if p.dim() == 4:
            O, I, R, S = p.shape
            return flat[off:off + n].view(O, R, S, I).permute(0, 3, 1, 2)
        return flat[off:off + n].view(p.shape)

    def _ensure_flat(self):
        if self._flat is not None:
            return
        ps = self._params()
        dev = ps[0].device
        if dev.type != 'cuda':
            raise RuntimeError('FlatAdam needs CUDA parameters: move the model to the GPU before '
                               'calling get_optimizer (there is no CPU fallback)')
        total = sum(_ceil4(p.numel()) for p in ps)
        fp = torch.zeros(total, device=dev, dtype=torch.float32)
        fg = torch.zeros(total, device=dev, dtype=torch.float32)
        # Layout: parameters in EXECUTION order (PlanNet stamps ``_flat_rank``), whatever order model.parameters()
        # lists them in - the backward pass then completes the flat gradient buffer from its END towards its start, so
        # dp.GradSync can all-reduce finished suffixes while earlier layers still run.  param_groups keep the
        # reference order (torch.optim state_dict indices stay those of model.parameters()).
        layout = sorted(range(len(ps)), key=lambda i: (getattr(ps[i], '_flat_rank', i), i))
        offs, off = [0] * len(ps), 0
        for i in layout:
            offs[i] = off
            off += _ceil4(ps[i].numel())
        self._offsets = []
        with torch.no_grad():
            for p, off in zip(ps, offs):
                if p.dtype != torch.float32:
                    raise TypeError('FlatAdam handles fp32 parameters only')
                if p.dim() == 4 and not p.is_contiguous(memory_format=torch.channels_last):
                    p.data = p.data.contiguous(memory_format=torch.channels_last)
                v = self._view(fp, off, p)
                v.copy_(p.data)
                old_grad = p.grad
                p.data = v
                g = self._view(fg, off, p)
                if old_grad is not None:
                    g.copy_(old_grad)
                p.grad = g
                p._flat_grad_view = g                     # ops._grad_buf re-attaches it if the grad is set to None
                p._flat_off = off                         # element offset of this parameter in the flat buffers
                self._offsets.append(off)
        self._grad_views = [p.grad for p in ps]
        self._flat = (fp, fg, torch.zeros_like(fp), torch.zeros_like(fp))
        self._hyper_host = self._hyper_tuple()
        self._hyper = torch.tensor(self._hyper_host, device=dev, dtype=torch.float32)
        self._step = torch.zeros((), device=dev, dtype=torch.int64)

    def _hyper_tuple(self):
        g0 = self.param_groups[0]
        return (g0['lr'], g0['betas'][0], g0['betas'][1], g0['eps'])

    def flat_state(self):
        """Everything a replica needs to continue identically: parameters, Adam moments, step counter
        (dp.GradSync.broadcast_state sends these from rank 0)."""
        self._ensure_flat()
        return [self._flat[0], self._flat[2], self._flat[3], self._step]

    @property
    def flat_params(self):
        self._ensure_flat()
        return self._flat[0]

    @property
    def flat_grads(self):
        self._ensure_flat()
        return self._flat[1]

    def sync_hyper(self):
        """Push lr/betas/eps to the device if a scheduler changed them (host-side, not captured)."""
        self._ensure_flat()
        cur = self._hyper_tuple()
        if cur != self._hyper_host:
            self._hyper.copy_(torch.tensor(cur, dtype=torch.float32))
            self._hyper_host = cur

    # ---- torch.optim API ------------------------------------------------------------------
    def _attach_grads(self, strict):
        """Every p.grad must alias its slot of the flat gradient buffer (the backward kernels accumulate into
        p.grad, step() reads the flat buffer).  nn.Module.zero_grad() / ``p.grad = None`` detach them: zero_grad
        re-attaches silently, step() (``strict``) refuses to apply gradients that went somewhere else."""
        for p, gv in zip(self._params(), self._grad_views):
            g = p.grad
            if g is gv:
                continue
            if g is not None and g.data_ptr() == gv.data_ptr():
                continue
            if strict and g is not None:
                raise RuntimeError('FlatAdam: a parameter\'s .grad no longer aliases the flat gradient buffer '
                                   '(was it replaced after zero_grad(set_to_none=True)?); its gradient would be lost')
            p.grad = gv

    def zero_grad(self, set_to_none=False):
        self._ensure_flat()
        self._attach_grads(strict=False)
        fg = self._flat[1]
        call('advmix_fill', ctypes.c_void_p(fg.data_ptr()), 0.0, fg.numel(),
             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    @torch.no_grad()
    def step(self, closure=None, sync_hyper=True):
        self._ensure_flat()
        if sync_hyper and not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        self._attach_grads(strict=True)
        fp, fg, m, v = self._flat
        P = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731
        call('advmix_adam', P(fp), P(fg), P(m), P(v), fp.numel(), P(self._hyper), P(self._step),
             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def state_dict(self):
        self._ensure_flat()
        ps = self._params()
        step = float(self._step.item())
        state = {}
        for i, (p, off) in enumerate(zip(ps, self._offsets)):
            state[i] = {'step': torch.tensor(step),
                        'exp_avg': self._view(self._flat[2], off, p).detach().clone(),
                        'exp_avg_sq': self._view(self._flat[3], off, p).detach().clone()}
        groups = [dict((k, v) for k, v in g.items() if k != 'params') for g in self.param_groups]
        for g in groups:
            g['params'] = list(range(len(ps)))
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        self._ensure_flat()
        ps = self._params()
        for k, v in sd['param_groups'][0].items():
            if k != 'params':
                self.param_groups[0][k] = v
        step = 0
        with torch.no_grad():
            for i, (p, off) in enumerate(zip(ps, self._offsets)):
                st = sd['state'].get(i) or sd['state'].get(str(i))
                if st is None:
                    continue
                self._view(self._flat[2], off, p).copy_(st['exp_avg'])
                self._view(self._flat[3], off, p).copy_(st['exp_avg_sq'])
                step = max(step, int(float(st['step'])))
        self._step.fill_(step)
        self.sync_hyper()


class FlatSGD(FlatAdam):
    """torch.optim.SGD(params, lr, momentum, weight_decay, nesterov) semantics (dampening 0) on the same flat buffers as
    FlatAdam: one kernel per step, hyper-parameters in device memory, gradients accumulated in place by the backward
    kernels (lib/utils/utils.py:80-88)."""

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and momentum <= 0:
            raise ValueError('Nesterov momentum requires a momentum')
        torch.optim.Optimizer.__init__(self, list(params), dict(lr=lr, momentum=momentum, dampening=0,
                                                                  weight_decay=weight_decay, nesterov=nesterov))
        self._flat = None
        if all(p.is_cuda for p in self._params()):
            self._ensure_flat()

    def _hyper_tuple(self):
        g0 = self.param_groups[0]
        return (g0['lr'], g0['momentum'], g0['weight_decay'], 1.0 if g0['nesterov'] else 0.0)

    @torch.no_grad()
    def step(self, closure=None, sync_hyper=True):
        self._ensure_flat()
        if sync_hyper and not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        self._attach_grads(strict=True)
        fp, fg, buf, _ = self._flat
        P = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731
        call('advmix_sgd', P(fp), P(fg), P(buf), fp.numel(), P(self._hyper),
             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def flat_state(self):
        self._ensure_flat()
        return [self._flat[0], self._flat[2]]

    def state_dict(self):
        self._ensure_flat()
        ps = self._params()
        state = {}
        if self.param_groups[0]['momentum'] != 0:
            for i, (p, off) in enumerate(zip(ps, self._offsets)):
                state[i] = {'momentum_buffer': self._view(self._flat[2], off, p).detach().clone()}
        groups = [dict((k, v) for k, v in g.items() if k != 'params') for g in self.param_groups]
        for g in groups:
            g['params'] = list(range(len(ps)))
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        self._ensure_flat()
        ps = self._params()
        for k, v in sd['param_groups'][0].items():
            if k != 'params':
                self.param_groups[0][k] = v
        with torch.no_grad():
            for i, (p, off) in enumerate(zip(ps, self._offsets)):
                st = sd['state'].get(i) or sd['state'].get(str(i))
                if st is not None and st.get('momentum_buffer') is not None:
                    self._view(self._flat[2], off, p).copy_(st['momentum_buffer'])
        self.sync_hyper()


def get_optimizer(cfg, model):
    """lib/utils/utils.py:78-94: Adam(lr) or SGD(lr, momentum, weight_decay, nesterov), both as ONE flat-buffer HIP
    kernel per step."""
    if cfg.TRAIN.OPTIMIZER == 'adam':
        return FlatAdam(model.parameters(), lr=cfg.TRAIN.LR)
    if cfg.TRAIN.OPTIMIZER == 'sgd':
        return FlatSGD(model.parameters(), lr=cfg.TRAIN.LR, momentum=cfg.TRAIN.MOMENTUM,
                       weight_decay=cfg.TRAIN.WD, nesterov=cfg.TRAIN.NESTEROV)
    return None


def save_checkpoint(states, is_best, output_dir, filename='checkpoint.pth', suffix='', grad_sync=None):
    """lib/utils/utils.py:97-108.  Data parallel (``grad_sync``): only rank 0 writes.  Its replica IS what the
    reference saves - nn.DataParallel keeps GPU 0's module, whose BatchNorm running statistics come from GPU 0's shard
    alone (tools/train.py:311-337); parameters and Adam state are identical on every rank (dp.GradSync)."""
    if grad_sync is not None and not grad_sync.checkpoint_rank():
        return
    import torch.distributed as dist
    if grad_sync is None and dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
        return                                              # the reference's call sites pass no grad_sync (tools/train.py:311-328)
    if suffix != '':
        torch.save(states, os.path.join(output_dir, filename[:-4] + '_' + suffix + '.pth'))
        if is_best and 'state_dict' in states:
            torch.save(states['best_state_dict'], os.path.join(output_dir, 'model_best_{}.pth'.format(suffix)))
    else:
        torch.save(states, os.path.join(output_dir, filename))
        if is_best and 'state_dict' in states:
            torch.save(states['best_state_dict'], os.path.join(output_dir, 'model_best.pth'))


_LOGGER_CALLS = [0]


def create_logger(args, cfg, cfg_name, phase='train'):
    """lib/utils/utils.py:22-75 for N processes: same arguments, same ``(logger, final_output_dir, tensorboard_log_dir)``
    triple, same directory names and log format.  The reference's version belongs to ONE process - ``if not exists():
    mkdir()`` (:24-27) lets two ranks both pass the test and the second ``mkdir`` raise FileExistsError, and every rank
    stamps its own minute into the names.  Here rank 0 makes the three directories (``exist_ok``) and publishes its time
    stamp through the process group's store; the other ranks wait for that key, so every rank returns the SAME two
    paths; log files carry a ``_rank<r>`` suffix on ranks > 0 (one writer per file).  Without a process group this is
    the reference's behaviour with ``exist_ok``."""
    import logging
    import time
    from pathlib import Path
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if multi else 0

    dataset = cfg.DATASET.DATASET + '_' + cfg.DATASET.HYBRID_JOINTS_TYPE if cfg.DATASET.HYBRID_JOINTS_TYPE \
        else cfg.DATASET.DATASET
    dataset = dataset.replace(':', '_')
    model = cfg.MODEL.NAME
    cfg_name = os.path.basename(cfg_name).split('.')[0]
    suffix = getattr(args, 'save_suffix', '')
    cfg_name = suffix if suffix != '' else cfg_name
    robust = bool(getattr(args, 'test_robust', False))
    root_output_dir = Path('output_robustness') if robust else Path(cfg.OUTPUT_DIR)
    final_output_dir = root_output_dir / dataset / model / cfg_name
    if robust:
        final_output_dir = final_output_dir / 'test_corruption'

    _LOGGER_CALLS[0] += 1
    key = 'advmix_create_logger_%d_%s' % (_LOGGER_CALLS[0], phase)
    if rank == 0:
        time_str = time.strftime('%Y-%m-%d-%H-%M')
    else:
        import datetime
        store = dist.distributed_c10d._get_default_store()
        store.wait([key], datetime.timedelta(minutes=10))
        time_str = store.get(key).decode()
    tensorboard_log_dir = Path(cfg.LOG_DIR) / dataset / model / (cfg_name + '_' + time_str)
    if robust:
        tensorboard_log_dir = Path(cfg.LOG_DIR) / dataset / model / 'test_robustness' / (cfg_name + '_' + time_str) / \
            args.corruption_type / str(args.severity)
    if rank == 0:
        print('=> creating {}'.format(final_output_dir))
        final_output_dir.mkdir(parents=True, exist_ok=True)
        print('=> creating {}'.format(tensorboard_log_dir))
        tensorboard_log_dir.mkdir(parents=True, exist_ok=True)
        if multi:                                           # only now may the other ranks open files in there
            dist.distributed_c10d._get_default_store().set(key, time_str)

    log_file = '{}_{}'.format(cfg_name, phase) if robust else '{}_{}_{}'.format(cfg_name, time_str, phase)
    log_file += ('_rank%d' % rank if rank else '') + '.log'
    logging.basicConfig(filename=str(final_output_dir / log_file), format='%(asctime)-15s %(message)s')
    logger = logging.getLogger()
    logger.setLevel(logging.INFO if rank == 0 else logging.WARNING)     # one voice on the console and in the main log
    logging.getLogger('').addHandler(logging.StreamHandler())
    return logger, str(final_output_dir), str(tensorboard_log_dir)


class NullSummaryWriter:
    """What ranks > 0 get for ``tensorboardX.SummaryWriter(log_dir=...)`` (tools/train.py:87-91): every method is a no-op,
    so the loops' ``writer.add_scalar`` / ``writer_dict['writer'].close()`` lines run unchanged and rank 0's event file is
    the only one."""

    def __init__(self, *a, **k):
        self.log_dir = k.get('log_dir', a[0] if a else None)

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return lambda *a, **k: None


def rank0_summary_writer(cls):
    """``cls`` on rank 0 (or without a process group), NullSummaryWriter on the others - decided when the writer is made."""
    def make(*a, **k):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return NullSummaryWriter(*a, **k)
        return cls(*a, **k)
    return make
