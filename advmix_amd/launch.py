"""One command, N ranks: start one fresh process per GPU and relay rank 0's output.

Replaces the reference's single-command multi-GPU entry - `GPUS: (0,...,7)` in the experiment YAMLs
(experiments/*:7) handed to `torch.nn.DataParallel(model, device_ids=cfg.GPUS)` in tools/train.py:69,106,109 -
for the one-process-per-GPU design (dp.py): the parent never touches the GPU (no HIP call, no
`torch.cuda.is_available()`; counting devices is safe on this image), starts N children with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, waits for them and exits with the first
failing child's code.  Nothing is re-exec'ed: a process that has initialised the GPU must not be replaced.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """Number of GPUs this process could hand to children - WITHOUT initialising the HIP runtime."""
    import torch
    return torch.cuda.device_count()


def spawn_ranks(argv, n, need_gpus=True, timeout_s=None, env_extra=None):
    """Run ``argv`` (a full command line) as ranks 0..n-1 of one node.  Rank 0 inherits stdout (its JSON line is the
    job's output); the other ranks' stdout goes to stderr.  Returns the job's exit code: 0 only if EVERY rank exited 0.
    Raises SystemExit with a non-zero code if fewer than ``n`` GPUs are visible (``need_gpus``) - never runs fewer
    ranks than asked for."""
    if n < 2:
        raise ValueError('spawn_ranks is for n >= 2 ranks')
    if need_gpus:
        have = visible_gpus()
        if have < n:
            sys.stderr.write('launch: --gpus %d asked for but only %d GPU(s) visible - refusing to run fewer ranks\n'
                             % (n, have))
            raise SystemExit(2)
    port = os.environ.get('MASTER_PORT') or str(free_port())
    procs = []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR='127.0.0.1', MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY='0')
            env.update(env_extra or {})
            procs.append(subprocess.Popen(argv, env=env, stdout=None if r == 0 else sys.stderr))
    except OSError as e:
        for p in procs:
            p.kill()
        sys.stderr.write('launch: could not start rank %d: %s\n' % (len(procs), e))
        raise SystemExit(3)
    t0 = time.time()
    code = 0
    live = list(procs)

    def _stop(signum, frame):                               # the parent is being stopped (driver timeout, ^C): take the
        for q in procs:                                     # ranks along - an orphaned rank would keep its GPU
            if q.poll() is None:
                q.terminate()
        raise SystemExit(128 + signum)
    old = {}
    for sg in (signal.SIGTERM, signal.SIGINT):
        try:
            old[sg] = signal.signal(sg, _stop)
        except ValueError:                                  # not the main thread: leave the handlers alone
            pass
    try:
        return _wait(procs, live, code, t0, timeout_s)
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)


def _wait(procs, live, code, t0, timeout_s):
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 1
                sys.stderr.write('launch: rank %d exited with %d - stopping the others\n' % (procs.index(p), rc))
                for q in live:
                    q.terminate()
        if timeout_s is not None and time.time() - t0 > timeout_s and live:
            sys.stderr.write('launch: timeout after %.0f s\n' % timeout_s)
            for q in live:
                q.kill()
            code = code or 124
        time.sleep(0.05)
    return code
