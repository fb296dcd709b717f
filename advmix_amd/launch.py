"""One command, N ranks: start one fresh process per GPU and relay rank 0's output.

Replaces the reference's single-command multi-GPU entry - `GPUS: (0,...,7)` in the experiment YAMLs
(experiments/*:7) handed to `torch.nn.DataParallel(model, device_ids=cfg.GPUS)` in tools/train.py:69,106,109 -
for the one-process-per-GPU design (dp.py): the parent never touches the GPU (no HIP call, no `torch.cuda.*` call at all:
GPUs are counted from the kernel driver's sysfs topology), starts N children with
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


KFD_NODES = '/sys/class/kfd/kfd/topology/nodes'


def _kfd_gpu_nodes(root=KFD_NODES, dev_dir='/dev/dri'):
    """GPU nodes of the amdkfd topology, in node order: those with ``simd_count > 0`` (CPU nodes have none) whose render
    node this process may open - the ROCr runtime skips a GPU it cannot open (a container that was handed a subset)."""
    try:
        nodes = sorted((n for n in os.listdir(root) if n.isdigit()), key=int)
    except OSError:
        return None
    gpus = []
    for n in nodes:
        props = {}
        try:
            with open(os.path.join(root, n, 'properties')) as f:
                for ln in f:
                    k, _, v = ln.partition(' ')
                    props[k] = v.strip()
        except OSError:
            continue
        if int(props.get('simd_count', '0') or 0) <= 0:
            continue
        minor = props.get('drm_render_minor')
        if minor not in (None, '', '0', '-1') and dev_dir:
            dev = os.path.join(dev_dir, 'renderD%s' % minor)
            if os.path.exists(dev_dir) and not os.access(dev, os.R_OK | os.W_OK):
                continue
        gpus.append(n)
    return gpus


def _visible_filter(n_phys, env=None):
    """How many of ``n_phys`` devices survive ROCR_VISIBLE_DEVICES, then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (HIP
    reads the latter as an alias).  Entries are indices (or GPU-xxxx UUIDs, which are taken at face value); like the
    runtimes, the list ends at the first entry that does not name a device."""
    env = os.environ if env is None else env
    n = n_phys
    for keys in (('ROCR_VISIBLE_DEVICES',), ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES')):
        val = next((env[k] for k in keys if k in env), None)
        if val is None:
            continue
        seen, cnt = set(), 0
        for tok in (t.strip() for t in val.split(',')):
            if tok.upper().startswith('GPU-'):
                pass
            elif not tok.lstrip('-').isdigit() or not (0 <= int(tok) < n) or tok in seen:
                break
            seen.add(tok)
            cnt += 1
        n = min(n, cnt)
    return n


def visible_gpus():
    """Number of GPUs this process could hand to children - WITHOUT initialising the HIP runtime here: on this torch
    build ``torch.cuda.device_count()`` is ``hipGetDeviceCount``, which brings HIP / HSA up in the calling process, and
    the children are forked from it.  The amdkfd sysfs topology says the same thing without a runtime; when the driver
    exposes no topology (no sysfs in the sandbox) a throw-away CHILD process asks torch - the parent still never does."""
    nodes = _kfd_gpu_nodes()
    if nodes is not None:
        return _visible_filter(len(nodes))
    try:
        out = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'],
                             capture_output=True, text=True, timeout=600)
        return int(out.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
        return 0


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
                       MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
            env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC (the only kind this driver has); a user's value wins
            # Multi-rank jobs replay their HIP graphs WITHOUT the runtime's captured-AQL-packet path: with two ranks on one
            # GPU, replays served from captured packets computed garbage after NULL-stream traffic between them, and never with
            # this switch (tools/dp_graph_repro.py: 0 of 5); at four launch lanes it costs nothing
            # (573.5 vs 573.6 / 575.3 images/s).  The step keeps the NULL stream idle as well - this is the second lock.
            env.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
            env.update(env_extra or {})
            # ranks > 0: stdout onto the parent's stderr by DESCRIPTOR (sys.stderr may be a capture object with no fileno)
            procs.append(subprocess.Popen(argv, env=env, stdout=None if r == 0 else 2))
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
